#!/bin/bash
# Counter summary of the fine kernel for the default bench workload (run on the GPU box):
#   tools/pmc_fine.sh <commit-id> [bench.py args...]  ->  gpurun_out/fine_counters[_<scene>].json (copy it to profiles/ to have
#   bench.py report `roofline.traffic`: the file records the SHA-256 of the kernel sources it was measured on)
# Separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel trace only, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; gfx950 correction: FETCH_SIZE x 2 for wide coalesced reads.
R="$(cd "$(dirname "$0")/.." && pwd)"
COMMIT=${1:-unknown}; shift
OUT=$R/gpurun_out/pmc_fine
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 "$R/bench.py" --steps 2 --warmup 1 --blocks 1 --min-seconds 0 --no-cpu-baseline --no-graph "$@" > "$OUT/g$i.log" 2>&1 || tail -3 "$OUT/g$i.log"
done
# static instruction mix of the kernel, priced with the measured issue costs (tools/isa_price.py)
"$R/tools/fine_isa.sh" > "$OUT/isa.txt" 2>&1 || tail -3 "$OUT/isa.txt"
python3 - "$OUT" "$COMMIT" "$R" "$@" <<'PY'
import csv, glob, hashlib, json, os, sys, collections
out, commit, root = sys.argv[1], sys.argv[2], sys.argv[3]
h = hashlib.sha256()
for f in ("kernels_fine.hip", "kcommon.h", "dmath.h"):
    h.update(open(os.path.join(root, "jello_amd", "csrc", f), "rb").read())
args = sys.argv[4:]
def opt(name, default):
    return args[args.index(name) + 1] if name in args else default
scene = opt("--scene", "c3")
agg = collections.defaultdict(list)
for f in glob.glob(out + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_fine_area" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in agg.items()}
fetch_kb, write_kb = avg.get("FETCH_SIZE"), avg.get("WRITE_SIZE")
j = {"kernel": "k_fine_area", "scene": scene, "paths": int(opt("--paths", 100000 if scene == "c3" else 30000)),
     "size": int(opt("--size", 4096 if scene == "c3" else 2048)), "aa": opt("--aa", "area"), "commit": commit,
     "kernel_source_sha256": h.hexdigest(),
     "source": "rocprofv3 --kernel-trace --pmc <group> (one pass per group: FETCH_SIZE | WRITE_SIZE | SQ_* x2), averages over the launches of bench.py --steps 2 --warmup 1 --no-graph",
     "counters_avg_per_launch": {k: round(v, 1) for k, v in sorted(avg.items())},
     "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
     "hbm_bytes_per_launch": None if fetch_kb is None or write_kb is None else int(fetch_kb * 1024 * 2 + write_kb * 1024),
     "correction": "bytes = KB * 1024; gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md: wide coalesced reads are tallied at half) -- an upper bound here, the kernel mixes 4/8/16-byte-per-lane loads; WRITE_SIZE exact",
     "valu_insts_per_launch": avg.get("SQ_INSTS_VALU"), "salu_insts_per_launch": avg.get("SQ_INSTS_SALU"),
     "lds_insts_per_launch": avg.get("SQ_INSTS_LDS"), "simds": 1024, "clock_ghz": 2.4,
     "tiles": (int(opt("--size", 4096 if scene == "c3" else 2048)) // 16) ** 2}
try:
    import subprocess
    inst = "ILi%dELb%dELb%dE" % ({"area": 0, "msaa8": 8, "msaa16": 16}[opt("--aa", "area")], 0 if scene in ("c3", "c1", "c2") else 1,
                                 0 if scene in ("c3", "c1", "c2") else 1)
    pr = json.loads(subprocess.check_output([sys.executable, root + "/tools/isa_price.py", "/tmp/asm/fine.s", "k_fine_area" + inst]).decode())
    j["isa_static_mix"] = pr
    j["valu_cycles_per_inst_static_mix"] = pr["valu_cycles_per_inst_static_mix"]
    j["issue_rates_source"] = "profiles/r03_ubench_issue_rates.txt (tools/ubench/valu3.hip on an MI355X): 2.2 / 4.2 / 8.1 cycles of a SIMD per wave64 instruction by class; SALU 4.08"
except Exception as e:  # noqa: BLE001
    j["isa_static_mix_error"] = str(e)
name = "fine_counters.json" if scene == "c3" else "fine_counters_%s.json" % scene
json.dump(j, open(root + "/gpurun_out/" + name, "w"), indent=1)
print(json.dumps(j, indent=1))
PY
