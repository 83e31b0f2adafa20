"""Where a tile-wave of k_fine_area spends its cycles (variant library built with -DFINE_TIMING; JELLO_HIP_LIB selects it):
total, inside build_batch, inside the stage-4 walk, batches and fills per tile.   python tools/fine_timing.py c3|c4|c4n"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jello_amd
from jello_amd import scenes
from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
s, p = {"c3": lambda: scenes.scene_c3(100_000, 4096), "c4": lambda: scenes.scene_c4(30_000, 2048), "c4n": lambda: scenes.scene_c4_nested(30_000, 2048)}[which]()
eng = jello_amd.Engine(0)
p.bump = s.bump_sizes(p.width, p.height)
rec, bump, attempts = eng.render(s, p, retain=True)
assert bump["failed"] == 0
out = (ctypes.c_uint64 * 6)()
eng.hip.jh_debug_fine_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
eng.hip.jh_debug_fine_timing(eng.ctx, out, 1)
eng.run(rec, RUN_DISPATCHES)
eng.sync()
eng.hip.jh_debug_fine_timing(eng.ctx, out, 0)
tot, batch, walk, nb, nf, nt = [int(v) for v in out]
print(which, "tiles", nt, "cycles per tile-wave: total %.0f, build_batch %.0f (%.1f %%), walk %.0f (%.1f %%); batches per tile %.2f, fills per tile %.2f, walk cycles per fill %.0f, batch cycles per batch %.0f"
      % (tot / nt, batch / nt, 100.0 * batch / tot, walk / nt, 100.0 * walk / tot, nb / nt, nf / nt, walk / max(nf, 1), batch / max(nb, 1)))
