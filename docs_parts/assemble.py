"""Assembles DESIGN.md from docs_parts/design_*.md, filling the {{...}} placeholders from the collected bench lines
(profiles/r04_bench_*.json).  Run from the repo root after tools/collect_profiles.sh results were copied to profiles/."""
import json, os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def J(name):
    p = os.path.join(R, "profiles", name)
    return json.loads(open(p).read().strip().splitlines()[-1]) if os.path.exists(p) else None
c3, c4, c4n, c1, c2 = J("r04_bench_c3.json"), J("r04_bench_c4.json"), J("r04_bench_c4n.json"), J("r04_bench_c1.json"), J("r04_bench_c2.json")
def f(x, n=3): return ("%." + str(n) + "f") % x
v = {}
if c3:
    st = c3["stage_ms"]
    v.update(FRAME_MS=f(c3["ms_per_step"]), MPIX="%d" % round(c3["value"]), PATHS=f(c3["paths_per_s"] / 1e6, 1), BLOCKS=str(c3["blocks"]),
             FINE_MS=f(c3["roofline"]["avg_ms"]), FRAC=f(100 * c3["roofline"]["frac"], 2),
             FRAC_COPY=f(100 * c3["roofline"]["achieved"] / c3["roofline"]["peak_measured_copy"], 1) if c3["roofline"].get("peak_measured_copy") else "n/a",
             FLATTEN_MS=f(st["flatten"]), PC_MS=f(st["path_count"]), COARSE_MS=f(st["coarse"]), PT_MS=f(st["path_tiling"]))
    one = c3.get("one_frame_at_a_time") or {"ms_per_step": c3["ms_per_step"], "value": c3["value"]}
    v.update(FRAME1_MS=f(one["ms_per_step"]), MPIX1="%d" % round(one["value"]))
    cb = c3.get("cpu_baseline") or {}
    v.update(CPU_MPIX=f(cb.get("value", 0), 1), CPU1_MPIX=f(cb.get("value_1thread", 0), 1))
for key, j in (("C4", c4), ("C4N", c4n), ("C1", c1), ("C2", c2)):
    if j:
        one = j.get("one_frame_at_a_time") or {"ms_per_step": j["ms_per_step"]}
        v[key + "_MS"] = f(one["ms_per_step"], 3 if key in ("C1", "C2") else 2)  # one frame at a time
        v[key + "_IF2"] = f(j["ms_per_step"], 3 if key in ("C1", "C2") else 2)  # the line's value: two frames in flight
        if key == "C4":
            v["C4_FINE"] = f(j["stage_ms"]["fine_area"], 2); v["C4_COARSE"] = f(j["stage_ms"]["coarse"], 2)
for k in ("N_CPU", "N_GPU"):
    if k in os.environ: v[k] = os.environ[k]
text = ""
for part in ("design_0_3.md", "design_4.md", "design_5.md", "design_6_8.md"):
    text += open(os.path.join(R, "docs_parts", part)).read().rstrip("\n") + "\n\n"
missing = set()
def sub(m):
    k = m.group(1)
    if k in v: return v[k]
    missing.add(k); return m.group(0)
text = re.sub(r"\{\{(\w+)\}\}", sub, text)
open(os.path.join(R, "DESIGN.md"), "w").write(text.rstrip("\n") + "\n")
print("DESIGN.md written, %d bytes; unfilled: %s" % (len(text), sorted(missing)))
