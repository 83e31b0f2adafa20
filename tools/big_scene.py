"""Sanity run at 4x the headline size: 400 k stroked+filled cubics at 8192^2 (regrow loop sizes the buffers)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import jello_amd
from jello_amd import scenes, BumpSizes

n, size = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (400_000, 8192)
scene, params = scenes.scene_c3(n, size)
params.bump = BumpSizes(lines=1 << 24, seg_counts=1 << 25, segments=1 << 25, tiles=1 << 23, ptcl=1 << 27, bin_data=1 << 22)
eng = jello_amd.Engine(0)
rec, bump, attempts = eng.render(scene, params, robust=True, retain=True)
print("bump", bump, "attempts", attempts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eng.run(rec, jello_amd.engine.RUN_DISPATCHES)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("ms/frame %.3f  Mpixels/s %.1f  paths/s %.3g" % (dt * 1e3, size * size / dt / 1e6, n / dt))
