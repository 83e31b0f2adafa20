#!/bin/bash
# same-box counters of one kernel across variant libraries (results of a variant may be wrong by construction):
#   [AB_ARGS="--scene c4"] tools/ab_counters.sh <kernel-regex> "<counter> <counter> ..." variant1 variant2 ...   (product = the product library)
# one rocprofv3 --pmc pass per variant (kernel trace only), averages per launch of the kernels that match
R="$(cd "$(dirname "$0")/.." && pwd)"
PAT=$1; CTRS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = product ]; then unset JELLO_HIP_LIB; else export JELLO_HIP_LIB=$R/jello_amd/libjello_hip_$v.so; fi
  OUT=$R/gpurun_out/abc_$v
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/raw" -- python3 "$R/bench.py" --steps 2 --warmup 1 --blocks 1 --min-seconds 0 --no-cpu-baseline --no-graph --in-flight 1 $AB_ARGS > "$OUT/bench.json" 2> "$OUT/bench.err" || true
  python3 - "$OUT" "$PAT" "$v" <<'PY'
import csv, glob, re, sys, collections
out, pat, v = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/raw/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r["Kernel_Name"]):
            agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print("==", v, k, {n: round(sum(x) / len(x)) for n, x in sorted(c.items())})
PY
  rm -rf "$OUT/raw"
done
