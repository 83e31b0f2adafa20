#!/bin/bash
# Occupancy of the clip/blend instantiations of k_fine_area on C4 (30 k paths, 9000 clip layers, 2048^2).
cd "$(dirname "$0")/.."
# performance-only macros (results do not change); the product library is rebuilt with the default flags on ANY exit
trap 'make -s -C jello_amd/csrc > /dev/null 2>&1' EXIT
for e in ${CLIP_WAVES:-2 3 4}; do
  make -s -C jello_amd/csrc EXTRA="-DFINE_CLIP_WAVES_PER_EU=$e" > /dev/null 2>&1
  echo "FINE_CLIP_WAVES_PER_EU=$e"
  timeout -k 10 300 python3 tools/time_configs.py 2>/dev/null | grep "C4"
done
