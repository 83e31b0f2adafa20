#!/bin/bash
# Per-kernel device times of one bench configuration: rocprofv3 --kernel-trace --stats around bench.py (run on the GPU box).
#   tools/kprof.sh <tag> [bench.py args...]   ->  gpurun_out/kprof_<tag>/kernel_stats.csv + a short table on stdout
# (--in-flight 1: one context, so that a kernel's duration is its own and not that of two frames sharing the device -- bench.py's
#  per-stage times and roofline.avg_ms come from a one-context pass as well; pass --in-flight 2 behind the tag to see the overlapped run)
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG=${1:-run}; shift
OUT=$R/gpurun_out/kprof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --in-flight 1 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
F=$(ls "$OUT"/raw/*/*kernel_stats.csv | head -1)
cp "$F" "$OUT/kernel_stats.csv"
rm -rf "$OUT/raw"
python3 "$R/profiles/kstats.py" "$OUT/kernel_stats.csv" 60
python3 - "$OUT/bench.json" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", j["ms_per_step"], "value", j["value"])
print({k:v for k,v in j["stage_ms"].items() if v>0.02})
PY
