#!/bin/bash
# Memory-side counters of selected kernels (run on the GPU box): vector L1 (TCP) requests to L2, their latency and the
# cycles the L1 stalls on pending requests, L2 (TCC) busy / tag-stall cycles.
#   tools/pmc_mem.sh <tag> <kernel,kernel,...> [bench.py args...]
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG=$1; KS=$2; shift 2
OUT=$R/gpurun_out/pmcmem_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
# (only this group: the TCC_EA0_* / TCC_HIT / TA_* groups made rocprofv3 abort after its 300 s limit on this pool)
for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 "$R/bench.py" --steps 2 --warmup 1 --blocks 1 --min-seconds 0 --no-cpu-baseline --no-graph "$@" > "$OUT/g$i.log" 2>&1 || { tail -3 "$OUT/g$i.log"; }
  python3 "$R/profiles/pmc.py" "$OUT"/g$i/*/*counter_collection.csv --k=$KS | tee -a "$OUT/summary.txt"
  rm -rf "$OUT/g$i"
done
