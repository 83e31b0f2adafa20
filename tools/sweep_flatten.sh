#!/bin/bash
# Sweep of k_flatten_items tuning macros (run on the GPU box): refill threshold, waves/EU, workgroups/CU.
cd "$(dirname "$0")/.."
# performance-only macros (results do not change); the product library is rebuilt with the default flags on ANY exit
trap 'make -s -C jello_amd/csrc > /dev/null 2>&1' EXIT
FL_CFGS=("32 4 5" "32 4 4" "32 4 6" "32 4 8" "32 4 10" "32 3 4" "32 3 5" "32 3 6")
for cfg in "${FL_CFGS[@]}"; do
  set -- $cfg
  make -s -C jello_amd/csrc EXTRA="-DFL_REFILL_LANES=${1}u -DFL_WAVES_PER_EU=$2 -DFL_BLOCKS_PER_CU=$3" > /dev/null 2>&1
  echo -n "refill=$1 waves/eu=$2 blocks/cu=$3  "
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 2 --blocks 3 --no-cpu-baseline --no-graph 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['stage_ms'].get('flatten'))"
done
