"""One-off parity runs at sizes beyond the test suite (oracle takes tens of seconds): every buffer and the image."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import jello_amd
from jello_amd import BumpSizes, scenes, Aa
from parity import compare

eng = jello_amd.Engine(0)
big = BumpSizes(lines=1 << 22, seg_counts=1 << 23, segments=1 << 23, tiles=1 << 22, ptcl=1 << 26, bin_data=1 << 21, blend_spill=1 << 24)
for name, mk, aa in [("c4 8000 @1024", lambda: scenes.scene_c4(8000, 1024), Aa.Area),
                     ("c2 1500 @2048", lambda: scenes.scene_c2(1500, 2048), Aa.Area),
                     ("c4 4000 @1024 msaa16", lambda: scenes.scene_c4(4000, 1024), Aa.Msaa16),
                     ("c3 50000 @2048 msaa8", lambda: scenes.scene_c3(50000, 2048), Aa.Msaa8)]:
    s, p = mk()
    p.bump = big
    p.aa = aa
    t0 = time.time()
    r = compare(eng, s, p)
    print("%-26s ok  lines %d segments %d ptcl %d  (%.1f s)" % (name, r["bump"]["lines"], r["bump"]["segments"], r["bump"]["ptcl"], time.time() - t0), flush=True)
