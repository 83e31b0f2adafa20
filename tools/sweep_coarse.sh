#!/bin/bash
# Sweep of coarse's workgroup shape (run on the GPU box): tile-pair cache size (LDS per workgroup), workgroups per CU the
# split aims for, and the largest split, on C3 and C4.
cd "$(dirname "$0")/.."
# performance-only macros (results do not change); the product library is rebuilt with the default flags on ANY exit
trap 'make -s -C jello_amd/csrc > /dev/null 2>&1' EXIT
CFGS=("1536 2 16" "1536 4 16" "768 6 16" "512 8 16")
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  make -s -C jello_amd/csrc EXTRA="-DCOARSE_TILE_CACHE=${1}u -DCOARSE_WG_PER_CU=${2}u -DCOARSE_MAX_SPLIT=${3}u" > /dev/null 2>&1
  for S in c3 c4; do
    echo -n "cache=$1 wg/cu=$2 maxsplit=$3 $S  "
    bash tools/kprof.sh sw --scene $S | grep "k_coarse" | awk '{printf "%s %s us  ", $1, $6}'; echo
  done
done
