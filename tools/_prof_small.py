import os, sys, time
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tools")
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
which = sys.argv[1] if len(sys.argv) > 1 else "huge"
s, p = scene_shapes(20, 1000, 2000, 4096) if which == "huge" else scene_shapes(300, 200, 1000, 4096)
p.bump = BumpSizes(lines=1 << 23, seg_counts=1 << 24, segments=1 << 24, tiles=1 << 24, ptcl=1 << 27, bin_data=1 << 22, blend_spill=1 << 20)
rec, bump, attempts = eng.render(s, p, robust=True, retain=True)
torch.cuda.synchronize()
for _ in range(20):
    eng.run(rec, jello_amd.engine.RUN_DISPATCHES)
torch.cuda.synchronize()
