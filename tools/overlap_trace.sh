#!/bin/bash
# How the kernels of two frames in flight share the device: rocprofv3 --kernel-trace around a short default bench run, then
# tools/overlap_trace.py on the trace (run on the GPU box).   tools/overlap_trace.sh [bench.py args...]  ->  gpurun_out/overlap.txt
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$R/gpurun_out/overlap
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -- python3 "$R/bench.py" --steps 60 --warmup 3 --blocks 1 --min-seconds 0 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -3 "$OUT/bench.err"; exit 1; }
F=$(ls "$OUT"/raw/*/*kernel_trace.csv | head -1)
head -2 "$F" > "$R/gpurun_out/overlap_head.txt"; python3 "$R/tools/overlap_trace.py" "$F" > "$R/gpurun_out/overlap.txt" 2> "$R/gpurun_out/overlap_err.txt"
rm -rf "$OUT/raw"
cat "$R/gpurun_out/overlap.txt"
