"""One-off parity sweep beyond the test suite: more fuzz seeds in all three AA modes, and the shape scenes of
tools/shape_scenes.py at sizes the oracle finishes in seconds.  Every buffer, the PTCL and the image, bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import jello_amd
from jello_amd import BumpSizes, scenes, Aa
from parity import compare
import shape_scenes as sh

eng = jello_amd.Engine(0)
big = lambda: BumpSizes(lines=1 << 21, seg_counts=1 << 22, segments=1 << 22, tiles=1 << 23, ptcl=1 << 25, bin_data=1 << 21, blend_spill=1 << 22)
cases = []
for seed in range(16, 64):
    cases.append(("fuzz %d" % seed, (lambda sd: lambda: scenes.scene_fuzz(sd))(seed), [Aa.Area, Aa.Msaa8, Aa.Msaa16][seed % 3]))
for seed in range(32):
    cases.append(("extreme fuzz %d" % seed, (lambda sd: lambda: scenes.scene_fuzz(sd, extreme=True))(seed), [Aa.Area, Aa.Msaa8, Aa.Msaa16][seed % 3]))
cases += [("circles r 20..150 @1024", lambda: sh.scene_shapes(400, 20, 150, 1024), Aa.Area),
          ("circles r 200..1000 @2048", lambda: sh.scene_shapes(120, 200, 1000, 2048), Aa.Area),
          ("circles r 1000..2000 @2048", lambda: sh.scene_shapes(12, 1000, 2000, 2048), Aa.Msaa8),
          ("long strokes @2048", lambda: sh.scene_long_lines(300, 2048), Aa.Area),
          ("tiny rects @1024", lambda: sh.scene_tiny_rects(50000, 1024), Aa.Area),
          ("polygon 60k @1024", lambda: sh.scene_polygon(60000, 1024), Aa.Area),
          ("gradients linear @1024", lambda: sh.scene_gradients(3000, 1024, "linear"), Aa.Area),
          ("gradients radial @1024", lambda: sh.scene_gradients(3000, 1024, "radial"), Aa.Msaa16),
          ("gradients sweep @1024", lambda: sh.scene_gradients(3000, 1024, "sweep"), Aa.Area),
          ("images @1024", lambda: sh.scene_gradients(3000, 1024, "image"), Aa.Area),
          ("glyphs @1280x720", lambda: sh.scene_glyphs(4000, 1280, 720), Aa.Area),
          ("glyphs msaa16 @1280x720", lambda: sh.scene_glyphs(4000, 1280, 720), Aa.Msaa16),
          ("wide 16384x1024", lambda: sh._wide(), Aa.Area)]
sel = sys.argv[1:]
bad = 0
for name, mk, aa in cases:
    if sel and not any(k in name for k in sel):
        continue
    s, p = mk()
    p.bump = big()
    p.aa = aa
    t0 = time.time()
    try:
        r = compare(eng, s, p)
        print("%-28s %-7s ok  lines %d segments %d tiles %d max/tile %d (%.1f s)" % (name, aa.name, r["bump"]["lines"], r["bump"]["segments"],
              r["bump"]["tile"], r.get("max_tile_segments", 0), time.time() - t0), flush=True)
    except AssertionError as e:
        bad += 1
        print("%-28s %-7s MISMATCH: %s" % (name, aa.name, str(e)[:300]), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
