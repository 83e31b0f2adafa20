#!/bin/bash
# usage: run_variants.sh name1 name2 ... ; "product" = product library
R=/root/repo
mkdir -p $R/gpurun_out/var
for v in "$@"; do
  if [ $v = product ]; then unset JELLO_HIP_LIB; else export JELLO_HIP_LIB=$R/jello_amd/libjello_hip_$v.so; fi
  timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/var/$v.json 2> $R/gpurun_out/var/$v.err || { tail -3 $R/gpurun_out/var/$v.err; exit 1; }
  python3 - $R/gpurun_out/var/$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "frame", d["ms_per_step"], "one at a time", (d.get("one_frame_at_a_time") or {}).get("ms_per_step"), "fine", d["roofline"]["avg_ms"], "stages", {k:round(v,4) for k,v in d.get("stage_ms",{}).items()} if "stage_ms" in d else "")
PY
done
