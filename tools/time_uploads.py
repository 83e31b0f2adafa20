"""Frame time of C3 with and without re-uploading the scene every frame (run on the GPU box): the PCIe-inclusive rate
DESIGN.md section 6 quotes."""
import sys, time
sys.path.insert(0, '.')
import jello_amd
from jello_amd import scenes
from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS
s, p = scenes.scene_c3(100000, 4096)
p.bump = s.bump_sizes(4096, 4096)
eng = jello_amd.Engine()
rec = jello_amd.Host().record(s, p)
eng.run(rec, RUN_UPLOADS | RUN_DISPATCHES); eng.sync()
nbytes = sum(len(c["data"]) for c in rec.commands() if c["kind"] in (0, 1, 2))
def t(flags, n=20):
    eng.sync(); t0 = time.perf_counter()
    for _ in range(n): eng.run(rec, flags)
    eng.sync(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(2):
    a = t(RUN_DISPATCHES); b = t(RUN_UPLOADS | RUN_DISPATCHES)
print("upload bytes %d  dispatch-only (eager) %.3f ms  uploads+dispatch %.3f ms  delta %.3f ms" % (nbytes, a, b, b - a))
