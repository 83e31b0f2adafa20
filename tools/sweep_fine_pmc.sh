#!/bin/bash
# Instruction counts of k_fine_area parts (FINE_EXP builds; timing-only variants, results wrong for EXP != 0).
cd "$(dirname "$0")/.."
R=$PWD
for e in 0 1 4 2 16 26; do
  rm -f jello_amd/csrc/kernels_fine.o
  make -s -C jello_amd/csrc EXTRA="-DFINE_EXP=$e" > /dev/null 2>&1
  mkdir -p $R/gpurun_out/fpmc/e$e
  (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/fpmc/e$e -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $R/gpurun_out/fpmc/e$e.log 2>&1)
  echo "FINE_EXP=$e"; python3 profiles/pmc.py $R/gpurun_out/fpmc/e$e/*/*counter_collection.csv --k=k_fine_area
done
rm -f jello_amd/csrc/kernels_fine.o
make -s -C jello_amd/csrc > /dev/null 2>&1
