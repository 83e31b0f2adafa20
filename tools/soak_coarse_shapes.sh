#!/bin/bash
# Shapes of the coarse kernel that the product defaults never pick on a 256-CU part: one or two workgroups per bin
# (COARSE_MAX_SPLIT), the smallest Tile cache (every batch worked on in many windows).  Parity suite + a fuzz soak per
# build (run on the GPU box; the product library is rebuilt with the default flags on ANY exit).
cd "$(dirname "$0")/.."
trap 'make -s -C jello_amd/csrc > /dev/null 2>&1' EXIT
for X in "-DCOARSE_MAX_SPLIT=1u" "-DCOARSE_MAX_SPLIT=2u" "-DCOARSE_TILE_CACHE=256u" "-DCOARSE_MAX_SPLIT=8u -DCOARSE_TILE_CACHE=300u"; do
  make -s -C jello_amd/csrc EXTRA="$X" > /dev/null 2>&1 || exit 1
  echo "[$X]"
  timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -m gpu -x -q 2>&1 | tail -1
  timeout -k 10 300 python3 tools/parity_soak.py 100 200 2>&1 | tail -1
done
