"""Two frames in flight with the fine stage on a stream of its own launch priority (run on the GPU box).

With two engine contexts taking the frames in turn (tools/frames_in_flight.py, bench.py's default) the kernel trace shows k_fine_area
keeping its speed -- its 65 536 workgroups take every slot that frees up -- while the other frame's chain of small, latency-bound kernels
crawls beside it (profiles/r04_overlap_in_flight.txt).  Graph kernel nodes take no priority attribute on this ROCm, so this splits a frame
in two: a captured graph of everything BUT fine on the context's main stream, and the fine dispatch launched eagerly on a second stream,
ordered by events (main -> fine -> main).  Modes: which priority the two streams get.
usage: python3 tools/fine_priority.py [--scene c3|c4|c4n] [--steps 200] [--blocks 5]"""
import argparse, ctypes, hashlib, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import jello_amd
from jello_amd import BumpSizes, scenes
from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS, RUN_SKIP_FINE, RUN_ONLY_FINE

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="c3")
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--blocks", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
s, p = {"c3": scenes.scene_c3, "c4": scenes.scene_c4, "c4n": scenes.scene_c4_nested}[args.scene]()
W, H = p.width, p.height
host = jello_amd.Host()


def make_stream(eng, level):
    h = ctypes.c_void_p()
    eng.hip.jh_stream_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    assert eng.hip.jh_stream_create(eng.ctx, level, ctypes.byref(h)) == 0
    return torch.cuda.ExternalStream(h.value, dev)


class Ctx:
    def __init__(self, rec, main_level, fine_level, split):
        self.e = jello_amd.Engine(0)
        self.main = make_stream(self.e, main_level)
        self.fine = make_stream(self.e, fine_level) if split else None
        self.e.set_stream(self.main.cuda_stream)
        self.out = torch.empty((H, W, 4), dtype=torch.float16, device=dev)
        self.rec = rec
        self.e.run(rec, RUN_UPLOADS | RUN_DISPATCHES, self.out.data_ptr())
        self.e.sync()
        self.split = split
        if split:  # the graph holds everything but fine
            self.e._check(self.e.hip.jh_graph_begin(self.e.ctx), "graph_begin")
            try:
                self.e.run(rec, RUN_DISPATCHES | RUN_SKIP_FINE, self.out.data_ptr())
            finally:
                g = ctypes.c_void_p()
                rc = self.e.hip.jh_graph_end(self.e.ctx, ctypes.byref(g))
            self.e._check(rc, "graph_end")
            self.g = g
            self.ev1, self.ev2 = torch.cuda.Event(), torch.cuda.Event()
        else:
            self.g = self.e.capture(rec, self.out.data_ptr())

    def frame(self):
        self.e.replay(self.g)
        if self.split:
            self.ev1.record(self.main)
            self.fine.wait_event(self.ev1)
            self.e.set_stream(self.fine.cuda_stream)
            self.e.run(self.rec, RUN_DISPATCHES | RUN_ONLY_FINE, self.out.data_ptr())
            self.ev2.record(self.fine)
            self.main.wait_event(self.ev2)
            self.e.set_stream(self.main.cuda_stream)

    def digest(self):
        torch.cuda.synchronize(dev)
        return hashlib.sha256(self.out.cpu().numpy().tobytes()).hexdigest()

    def close(self):
        self.e.graph_destroy(self.g)
        self.e.release(self.rec)
        self.e.close()


# sizes as bench.py takes them
e0 = jello_amd.Engine(0)
p.bump = s.bump_sizes(W, H)
rec0, bump, attempts = e0.render(s, p, robust=True)
assert bump["failed"] == 0
cfg0 = rec0.config
m = lambda x: int(x * 1.1) + 4096
p.bump = BumpSizes(lines=m(bump["lines"]), seg_counts=m(bump["seg_counts"]), segments=m(bump["segments"]), tiles=m(bump["tile"]),
                   ptcl=m(bump["ptcl"] + cfg0["width_in_tiles"] * cfg0["height_in_tiles"] * 64),
                   bin_data=m(bump["binning"] + cfg0["bin_data_start"]), blend_spill=max(4096, m(bump["blend"])))
del rec0
e0.close()
rec = host.record(s, p)

res = {}
ref = None
for name, main_level, fine_level, split in (("one graph per frame (bench.py)", 0, 0, False),
                                            ("fine eager on a second stream, same priority", 0, 0, True),
                                            ("fine on a LOW priority stream", 0, -1, True),
                                            ("chain on a HIGH priority stream, fine default", 1, 0, True),
                                            ("chain HIGH, fine LOW", 1, -1, True)):
    ctxs = [Ctx(rec, main_level, fine_level, split) for _ in range(2)]
    for c in ctxs:
        d = c.digest()
        if ref is None: ref = d
        assert d == ref
    row = {}
    for n in (1, 2):
        for i in range(6):
            ctxs[i % n].frame()
        times = []
        for b in range(args.blocks):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(args.steps):
                ctxs[i % n].frame()
            torch.cuda.synchronize(dev)
            times.append((time.perf_counter() - t0) / args.steps)
        times.sort()
        row["in_flight_%d_ms" % n] = round(times[len(times) // 2] * 1e3, 4)
    for c in ctxs:
        assert c.digest() == ref, "the frame changed"
        c.close()
    res[name] = row
    print(name, row, flush=True)
print(json.dumps({"scene": args.scene, "steps": args.steps, "verified": "every context's image equals the first frame, before and after", "modes": res}))
