"""Reads a rocprofv3 kernel trace of `bench.py` (two frames in flight) and reports, for the longest stretch of back-to-back
k_fine_area launches (= the timed block), how the two streams share the device: wall time per frame, the fraction of the wall time
with 0 / 1 / 2+ kernels in flight, per kernel its mean duration here, and how much of it ran beside a kernel of the OTHER stream."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def col(r, *names):
    for n in names:
        if n in r: return r[n]
    raise KeyError(names)
ev = []
for r in rows:
    s, e = int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp"))
    q = col(r, "Queue_Id", "Queue_ID") + "/" + col(r, "Stream_Id")
    name = col(r, "Kernel_Name").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    ev.append((s, e, q, name))
ev.sort()
# bench.py --steps K --blocks 1 launches the fine kernel, from the end of the run backwards: K times in the eager per-stage pass, 3 (K + 1)
# times one frame at a time (warm-up 3 + three blocks ... with --warmup 3: 3 + 3 K), K times in the timed block with two frames in
# flight -- the block this report is about.
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
fine = [x for x in ev if "k_fine" in x[3]]
tail = K + 3 + 3 * K
blk = fine[-(tail + K):-tail]
t0, t1 = blk[0][0], blk[-1][1]
win = [x for x in ev if x[0] >= t0 and x[1] <= t1]
queues = sorted(set(x[2] for x in win))
print("window: %d fine launches, %.3f ms wall, %.4f ms per frame, queues %s" % (len(blk), (t1 - t0) / 1e6, (t1 - t0) / 1e6 / len(blk), queues))
pts = []
for s, e, q, n in win:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth = 0; last = t0; hist = collections.Counter()
for t, d in pts:
    hist[min(depth, 3)] += t - last
    last = t; depth += d
tot = sum(hist.values())
print("kernels in flight: " + ", ".join("%d: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
# per kernel: duration and the share of it spent beside a kernel of another queue
byq = collections.defaultdict(list)
for s, e, q, n in win: byq[q].append((s, e))
def overlap_with_others(s, e, q):
    o = 0
    for q2, iv in byq.items():
        if q2 == q: continue
        for s2, e2 in iv:
            if e2 <= s: continue
            if s2 >= e: break
            o += min(e, e2) - max(s, s2)
    return o
agg = collections.defaultdict(lambda: [0, 0, 0])
for s, e, q, n in win:
    a = agg[n]; a[0] += 1; a[1] += e - s; a[2] += overlap_with_others(s, e, q)
print("%-34s %6s %10s %8s" % ("kernel", "calls", "mean us", "beside"))
for n, (c, d, o) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-34s %6d %10.1f %7.0f %%" % (n[:34], c, d / c / 1e3, 100.0 * o / max(d, 1)))
