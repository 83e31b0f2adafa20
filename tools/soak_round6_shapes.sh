#!/bin/bash
# Round-6 code paths that the product defaults (almost) never take, forced by special builds and run through the parity suite + a fuzz
# soak each (on the GPU box; the product library is rebuilt with the default flags on ANY exit):
#   multisampled fine: the walk at the fill for every segment with more than 5 touched pixels (MsState::direct, otherwise only for
#   segments far outside their tile); lists of 64 touched pixels (fills that span many batches, the wholesale clear after a fill of
#   several pieces);  coarse with clip layers: one workgroup per bin (64 tiles per wave), sixteen, an arena share of ONE chunk (a new
#   share per chunk boundary), and the one-lane walk of rounds 1-5 on the one-walk route.
cd "$(dirname "$0")/.."
trap 'make -s -C jello_amd/csrc > /dev/null 2>&1' EXIT
for X in "-DMS_FORCE_DIRECT_ABOVE=5u" "-DMS_CAP_OVERRIDE=64u" "-DCOARSE_MAX_SPLIT=1u" "-DCOARSE_PAR_WG_PER_CU=16u -DCOARSE_POOL_CHUNKS=1u" "-DCOARSE_PAR_WALK=0"; do
  make -s -C jello_amd/csrc EXTRA="$X" > /dev/null 2>&1 || { echo "[$X] build failed"; exit 1; }
  echo "[$X]"
  timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_gpu_clip.py -m gpu -x -q 2>&1 | tail -1
  timeout -k 10 300 python3 tools/parity_soak.py 940000 300 2>&1 | tail -1
done
