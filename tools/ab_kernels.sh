#!/bin/bash
# same-box A/B of kernel times (AB_TOLERATE=1: a variant whose frames are wrong by construction still reports its kernel times): [AB_ARGS="--scene c4"] tools/ab_kernels.sh <kernel-regex> variant1 variant2 ... (product = the product library), 2 rounds
R="$(cd "$(dirname "$0")/.." && pwd)"
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for v in "$@"; do
  if [ $v = product ]; then unset JELLO_HIP_LIB; else export JELLO_HIP_LIB=$R/jello_amd/libjello_hip_$v.so; fi
  OUT=$R/gpurun_out/ab_$v
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -- python3 "$R/bench.py" --steps 20 --warmup 3 --blocks 2 --min-seconds 0 --no-cpu-baseline --in-flight 1 $AB_ARGS > "$OUT/bench.json" 2> "$OUT/bench.err" || { [ -n "$AB_TOLERATE" ] || { tail -3 "$OUT/bench.err"; exit 1; }; }
  F=$(ls "$OUT"/raw/*/*kernel_stats.csv | head -1)
  echo "== $v (round $round)"; python3 "$R/profiles/kstats.py" "$F" 60 | grep -E "$PAT"
  rm -rf "$OUT/raw"
done
done
