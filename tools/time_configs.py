"""Frame times of the other BASELINE.json configs (C1, C2 substitute, C4) on one GPU: regrow once, then replay."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import jello_amd
from jello_amd import scenes, BumpSizes, Aa

eng = jello_amd.Engine(0)
big = lambda: BumpSizes(lines=1 << 23, seg_counts=1 << 24, segments=1 << 24, tiles=1 << 23, ptcl=1 << 27, bin_data=1 << 22, blend_spill=1 << 26)
for name, mk in [("C1 rect + stroked cubic, 512^2", scenes.scene_c1),
                 ("C2 substitute: 300 blobs, 1024^2", lambda: scenes.scene_c2(300, 1024)),
                 ("C4 30k paths, clips + radial gradients + blends, 2048^2", lambda: scenes.scene_c4(30000, 2048))]:
    for aa in (Aa.Area, Aa.Msaa16):
        s, p = mk()
        p.bump = big()
        p.aa = aa
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
        print("%-58s %-7s %.3f ms/frame  %s" % (name, aa.name, dt * 1e3, ", ".join("%s %.3f" % (n, ms) for ms, n in top)), flush=True)
        eng.release(rec)
