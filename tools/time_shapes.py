"""Frame times for scenes with mid-size and large paths, long lines, very many tiny paths, one huge path: the cases where
a work distribution made for many small paths collapses.  Prints ms/frame and the slowest stages.
usage: python3 tools/time_shapes.py [substring of a case name ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import jello_amd
from shape_scenes import select, big_buffers

eng = jello_amd.Engine(0)
for name, mk in select(sys.argv[1:]):
    s, p = mk()
    p.bump = big_buffers()
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
