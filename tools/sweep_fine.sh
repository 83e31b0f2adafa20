#!/bin/bash
# Differential timing of k_fine_area parts (results are wrong for FINE_EXP != 0; timing only).
cd "$(dirname "$0")/.."
for e in ${FINE_EXPS:-0 1 4 2 8 16 18 26}; do
  rm -f jello_amd/csrc/kernels_fine.o
  make -s -C jello_amd/csrc EXTRA="-DFINE_EXP=$e" > /dev/null 2>&1
  echo -n "FINE_EXP=$e  "
  timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-graph 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['stage_ms'].get('fine_area'))"
done
rm -f jello_amd/csrc/kernels_fine.o
make -s -C jello_amd/csrc > /dev/null 2>&1
