#!/bin/bash
# Differential timing of fill_path_ms parts (FINE_MS_EXP builds; timing only, results wrong for EXP != 0):
# 1 = no pixel pass, 2 = no resolve, 4 = no clearing of the sample words.
cd "$(dirname "$0")/.."
for e in ${MS_EXPS:-0 1 2 4 7}; do
  rm -f jello_amd/csrc/kernels_fine.o
  make -s -C jello_amd/csrc EXTRA="-DFINE_MS_EXP=$e" > /dev/null 2>&1
  for a in msaa8 msaa16; do
    echo -n "FINE_MS_EXP=$e $a  "
    timeout -k 10 200 python3 bench.py --aa $a --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['stage_ms'].get('fine_$a'))"
  done
done
rm -f jello_amd/csrc/kernels_fine.o
make -s -C jello_amd/csrc > /dev/null 2>&1
