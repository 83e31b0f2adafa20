#!/bin/bash
# Sweep of k_flatten_bbox's range sizing (run on the GPU box): FB_TARGET_WAVES on C3 at 100 k and 20 k paths.
cd "$(dirname "$0")/.."
# performance-only macro (results do not change); the product library is rebuilt with the default flags on ANY exit
trap 'make -s -C jello_amd/csrc > /dev/null 2>&1' EXIT
for T in 256 1024 2048 8192; do
  make -s -C jello_amd/csrc EXTRA="-DFB_TARGET_WAVES=${T}u" > /dev/null 2>&1
  for P in 100000 20000; do
    echo -n "target_waves=$T paths=$P  "
    bash tools/kprof.sh bbs --paths $P | grep "k_flatten_bbox"
  done
done
