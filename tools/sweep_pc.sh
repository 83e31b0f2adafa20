#!/bin/bash
# path_count stage time on C3 for several thresholds between the serial per-path ranking and the list route.
cd "$(dirname "$0")/.."
# performance-only macros (results do not change); the product library is rebuilt with the default flags on ANY exit
trap 'make -s -C jello_amd/csrc > /dev/null 2>&1' EXIT
for e in ${PC_THRESH:-64 128 256 1024}; do
  make -s -C jello_amd/csrc EXTRA="-DPC_BIG_PATH=${e}u" > /dev/null 2>&1
  echo -n "PC_BIG_PATH=$e  "
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['stage_ms'].get('path_count'))"
done
