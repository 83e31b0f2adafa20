"""Sums the logs of tools/soak_flatten_fast.py runs (gpurun_out/r4_ffsoak*.log, gpurun_out/ffsoak_*.log) into
profiles/r04_flatten_fast_soak.txt: per log the seeds it covered (its TOTAL line, or its last progress line if the run was cut off
by the time limit) and the totals.  usage: python3 tools/soak_flatten_fast_summary.py > profiles/r04_flatten_fast_soak.txt"""
import glob, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
logs = sorted(glob.glob(os.path.join(R, "gpurun_out", "r4_ffsoak*.log")) + glob.glob(os.path.join(R, "gpurun_out", "ffsoak_*.log")))
tot = dict(seeds=0, nodes=0, und=0, con=0, vio=0, runs=0)
rows = []
scene_rows = {}
for p in logs:
    txt = open(p).read()
    first = None
    seeds = nodes = und = con = vio = 0
    done = False
    for line in txt.splitlines():
        m = re.match(r"(C3|C4|C4 nested)\s+nodes\s+(\d+)\s+undecided\s+(\d+) \(([\d.]+) %\)\s+contradictions (\d+)\s+bound violations (\d+)", line)
        if m:
            scene_rows[m.group(1)] = line.strip()
        m = re.match(r"\.\.\. (\d+) seeds, nodes (\d+) undecided (\d+) contradictions (\d+) violations (\d+)", line)
        if m:
            seeds, nodes, und, con, vio = (int(m.group(i)) for i in range(1, 6))
        m = re.match(r"TOTAL over C3, C4, C4 nested and fuzz seeds (\d+)\.\.(\d+) .*: nodes (\d+)\s+undecided (\d+) .*contradictions (\d+)\s+bound violations (\d+)", line)
        if m:
            first = int(m.group(1))
            seeds = int(m.group(2)) - first + 1
            nodes, und, con, vio = (int(m.group(i)) for i in range(3, 7))
            done = True
    if seeds == 0:
        continue
    rows.append("%-28s %6d seeds%s  nodes %10d  undecided %7d (%.3f %%)  contradictions %d  bound violations %d" % (
        os.path.basename(p), seeds, "" if done else " (cut off by the time limit)", nodes, und, 100.0 * und / max(nodes, 1), con, vio))
    tot["seeds"] += seeds; tot["nodes"] += nodes; tot["und"] += und; tot["con"] += con; tot["vio"] += vio; tot["runs"] += 1
print("tools/soak_flatten_fast.py on MI355X, check build (make VARIANT=ffcheck EXTRA=-DFL_FAST_CHECK): k_flatten_items evaluates the pinned")
print("sequence next to ff_decide on every node it tests; every run renders C3, C4 and the nested C4 once, then its fuzz seeds, each as a")
print("plain and an extreme scene (2 scenes per seed).  The first 3000 seeds ran the estimate with IEEE division / square root, all later")
print("ones the 1-ulp instructions of the product build.\n")
for k in ("C3", "C4", "C4 nested"):
    if k in scene_rows: print(scene_rows[k])
print()
for r in rows: print(r)
print("\nTOTAL  %d runs, %d seeds = %d fuzz scenes (+ C3, C4, C4 nested per run): nodes %d  undecided %d (%.3f %%)  contradictions %d  bound violations %d" % (
    tot["runs"], tot["seeds"], 2 * tot["seeds"], tot["nodes"], tot["und"], 100.0 * tot["und"] / max(tot["nodes"], 1), tot["con"], tot["vio"]))
sys.exit(1 if (tot["con"] or tot["vio"]) else 0)
