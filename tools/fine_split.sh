#!/bin/bash
# Per-stage split of k_fine_area by DIFFERENTIAL BUILDS (run on the GPU box): each variant library leaves one part of the
# kernel out (FINE_SKIP in kernels_fine.hip; its results are wrong, only its counters and times are read), built as
# jello_amd/libjello_hip_skipN.so beside the product library, which is never touched.
#   tools/fine_split.sh <commit-id> [bench.py args...]   ->  gpurun_out/fine_split.json
R="$(cd "$(dirname "$0")/.." && pwd)"
COMMIT=${1:-unknown}; shift
OUT=$R/gpurun_out/fine_split
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for n in 0 1 2 3 4 5 6; do
  if [ $n = 0 ]; then LIB=$R/jello_amd/libjello_hip.so; else
    make -s -j8 -C $R/jello_amd/csrc VARIANT=skip$n EXTRA="-DJH_VARIANT_BUILD -DFINE_SKIP=$n" > $OUT/build$n.log 2>&1 || { tail -5 $OUT/build$n.log; exit 1; }
    LIB=$R/jello_amd/libjello_hip_skip$n.so
  fi
  export JELLO_HIP_LIB=$LIB
  python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/bench$n.json 2> $OUT/bench$n.err || tail -3 $OUT/bench$n.err
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/p$n" -- python3 "$R/bench.py" --steps 2 --warmup 1 --blocks 1 --min-seconds 0 --no-cpu-baseline --no-graph "$@" > "$OUT/p$n.log" 2>&1 || tail -3 "$OUT/p$n.log"
done
unset JELLO_HIP_LIB
python3 - "$OUT" "$COMMIT" "$R" <<'PY'
import csv, glob, json, sys, collections
out, commit, root = sys.argv[1:4]
names = {0: "product library", 1: "without stage 3 (crossing-pixel formula)", 2: "without stage 4 (row walk, y_edge terms)",
         3: "without stages 2 + 3 (pair evaluation)", 4: "without batches (stages 1-4)", 5: "without the solid-colour composite",
         6: "FLOOR: real PTCL / windows / pair and crossing-pixel counts, only the arithmetic of the output (y-part per pair, formula per crossing pixel, two packed adds per segment, finalisation, composite, store)"}
rows = []
for n in range(7):
    agg = collections.defaultdict(list)
    for f in glob.glob("%s/p%d/*/*counter_collection.csv" % (out, n)):
        for r in csv.DictReader(open(f)):
            if "k_fine_area" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = {k: sum(v) / len(v) for k, v in agg.items()}
    try:
        b = json.load(open("%s/bench%d.json" % (out, n)))
        ms = b["roofline"]["avg_ms"]
    except Exception:
        ms = None
    rows.append({"variant": n, "what": names[n], "fine_ms": ms, "counters": {k: round(v) for k, v in sorted(avg.items())}})
tiles = 65536.0
base = rows[0]["counters"]
for r in rows[1:]:
    c = r["counters"]
    r["delta_per_tile"] = {k: round((base.get(k, 0) - c.get(k, 0)) / tiles, 1) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
    if r["fine_ms"] is not None and rows[0]["fine_ms"] is not None:
        r["delta_ms"] = round(rows[0]["fine_ms"] - r["fine_ms"], 4)
j = {"kernel": "k_fine_area", "commit": commit, "method": "differential builds (FINE_SKIP=n variant libraries, results wrong by construction), rocprofv3 --pmc per launch, bench.py stage time", "rows": rows}
json.dump(j, open(root + "/gpurun_out/fine_split.json", "w"), indent=1)
for r in rows:
    print(r["variant"], r["what"], r["fine_ms"], r.get("delta_per_tile"), r.get("delta_ms"))
PY
