#!/bin/bash
# Occupancy sweep for k_flatten_items (run on the GPU box): waves/EU x blocks/CU.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cfg in "2 4" "2 2" "3 3" "4 4"; do
  set -- $cfg
  rm -f jello_amd/csrc/kernels_flatten.o
  make -s -C jello_amd/csrc EXTRA="-DFL_WAVES_PER_EU=$1 -DFL_BLOCKS_PER_CU=$2" > /dev/null 2>&1
  echo "== waves/eu=$1 blocks/cu=$2"
  timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-graph | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['stage_ms'].get('flatten'))"
done
rm -f jello_amd/csrc/kernels_flatten.o
make -s -C jello_amd/csrc > /dev/null 2>&1
