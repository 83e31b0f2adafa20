"""Frame times for scenes with mid-size and large paths (hundreds to thousands of tile crossings per path): the cases
the per-path serial rank route of path_count handles badly.  Prints ms/frame and the slowest stages."""
import os, sys, time, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import jello_amd
from jello_amd import scenes, BumpSizes, Aa
from jello_amd.scene import Scene, Path, Brush, Fill, Stroke, RenderParams
from jello_amd.scenes import SplitMix64

def scene_shapes(n, rmin, rmax, size, seed=77):
    r = SplitMix64(seed)
    s = Scene()
    for i in range(n):
        cx, cy, rad = r.uniform(0, size), r.uniform(0, size), r.uniform(rmin, rmax)
        col = (r.uniform(), r.uniform(), r.uniform(), 0.5)
        if i % 2 == 0:
            s.fill(Fill.NonZero, None, Brush.solid(col), None, Path.circle(cx, cy, rad))
        else:
            s.stroke(Stroke(width=3.0), None, Brush.solid(col), None, Path.circle(cx, cy, rad))
    return s, RenderParams(size, size, base_color=(0, 0, 0, 1))

eng = jello_amd.Engine(0)
big = lambda: BumpSizes(lines=1 << 23, seg_counts=1 << 24, segments=1 << 24, tiles=1 << 24, ptcl=1 << 27, bin_data=1 << 22, blend_spill=1 << 20)
cases = [("C1", scenes.scene_c1),
         ("2000 circles r 20..150, 1920x1088", lambda: scene_shapes(2000, 20, 150, 1920)),
         ("300 circles r 200..1000, 4096^2", lambda: scene_shapes(300, 200, 1000, 4096)),
         ("20 circles r 1000..2000, 4096^2", lambda: scene_shapes(20, 1000, 2000, 4096)),
         ("C3 20k paths, 4096^2", lambda: scenes.scene_c3(20000, 4096))]
for name, mk in cases:
    s, p = mk()
    if name.startswith("2000"):
        p.height = 1088
    p.bump = big()
    rec, bump, attempts = eng.render(s, p, robust=True, retain=True)
    assert bump["failed"] == 0, bump
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.run(rec, jello_amd.engine.RUN_DISPATCHES)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    eng.profile(True)
    eng.run(rec, jello_amd.engine.RUN_DISPATCHES)
    torch.cuda.synchronize()
    prof = eng.profile_collect(1 << 12)
    eng.profile(False)
    top = sorted(((ms, n) for n, ms in prof), reverse=True)[:4]
    print("%-36s %.3f ms/frame  segs %d  %s" % (name, dt * 1e3, bump["seg_counts"], ", ".join("%s %.3f" % (n, ms) for ms, n in top)), flush=True)
    eng.release(rec)
