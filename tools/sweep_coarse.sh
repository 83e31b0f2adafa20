#!/bin/bash
# Differential timing of k_coarse parts (results are wrong for COARSE_EXP != 0; timing only).
cd "$(dirname "$0")/.."
for e in ${COARSE_EXPS:-0 1 2 3 7}; do
  rm -f jello_amd/csrc/kernels_coarse.o
  make -s -C jello_amd/csrc EXTRA="-DCOARSE_EXP=$e" > /dev/null 2>&1
  echo -n "COARSE_EXP=$e  "
  timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-graph 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['stage_ms'].get('coarse'))"
done
rm -f jello_amd/csrc/kernels_coarse.o
make -s -C jello_amd/csrc > /dev/null 2>&1
