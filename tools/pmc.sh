#!/bin/bash
# Hardware counters of selected kernels for one bench configuration (run on the GPU box): one rocprofv3 --pmc pass per
# counter group (counters of one group fit the hardware together), kernel trace only.
#   tools/pmc.sh <tag> <kernel,kernel,...> [bench.py args...]
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG=$1; KS=$2; shift 2
OUT=$R/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 "$R/bench.py" --steps 2 --warmup 1 --blocks 1 --min-seconds 0 --no-cpu-baseline --no-graph "$@" > "$OUT/g$i.log" 2>&1 || { tail -3 "$OUT/g$i.log"; }
  python3 "$R/profiles/pmc.py" "$OUT"/g$i/*/*counter_collection.csv --k=$KS | tee -a "$OUT/summary.txt"
  rm -rf "$OUT/g$i"
done
