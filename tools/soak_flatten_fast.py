"""Soak of the transcendental-free subdivision test (jello_amd/csrc/flatten_fast.h) on the GPU.

Needs the CHECK build of the library (`make -C jello_amd/csrc VARIANT=ffcheck EXTRA=-DFL_FAST_CHECK`), which evaluates the
pinned sequence next to the fast decision on every node k_flatten_items tests and counts
    nodes, undecided nodes (the fall-back rate), contradictions (fast decision != pinned decision), bound violations
    (|v~ - v| > delta).
Scenes: C3 (100 k paths, 4096^2), C4 and its nested variant (30 k paths), then `count` fuzz seeds, plain and extreme.
usage: JELLO_HIP_LIB=jello_amd/libjello_hip_ffcheck.so python3 tools/soak_flatten_fast.py [first_seed [count]]
Exit code 1 on any contradiction or violation."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if "ffcheck" not in os.environ.get("JELLO_HIP_LIB", ""):
    sys.exit("set JELLO_HIP_LIB to the ffcheck build (see the docstring)")
import jello_amd
from jello_amd import BumpSizes, scenes

first = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
eng = jello_amd.Engine(0)
hip = eng.hip
hip.jh_debug_flatten_fast_stats.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]


def stats(reset=True):
    out = (ctypes.c_uint32 * 8)()
    assert hip.jh_debug_flatten_fast_stats(out, 1 if reset else 0) == 0
    return list(out)[:4]


stats()
tot = [0, 0, 0, 0]


def add(name, st):
    for i in range(4):
        tot[i] += st[i]
    if name:
        print("%-10s nodes %9d  undecided %7d (%.3f %%)  contradictions %d  bound violations %d" % (
            name, st[0], st[1], 100.0 * st[1] / max(st[0], 1), st[2], st[3]), flush=True)


for name, fn in (("C3", lambda: scenes.scene_c3()), ("C4", lambda: scenes.scene_c4()), ("C4 nested", lambda: scenes.scene_c4_nested())):
    s, p = fn()
    p.bump = s.bump_sizes(p.width, p.height)
    rec, bump, attempts = eng.render(s, p, robust=True)
    assert bump["failed"] == 0
    add(name, stats())
t0 = time.time()
for seed in range(first, first + count):
    for extreme in (False, True):
        s, p = scenes.scene_fuzz(seed, extreme=extreme, size=[256, 300, 512][seed % 3])
        p.bump = BumpSizes(lines=1 << 19, seg_counts=1 << 20, segments=1 << 20, tiles=1 << 21, ptcl=1 << 23, bin_data=1 << 19, blend_spill=1 << 21)
        eng.render(s, p, robust=False)
    if (seed - first) % 500 == 499:
        add("", stats())
        print("... %d seeds, nodes %d undecided %d contradictions %d violations %d, %.0f s" % (seed - first + 1, tot[0], tot[1], tot[2], tot[3], time.time() - t0), flush=True)
add("", stats())
print("TOTAL over C3, C4, C4 nested and fuzz seeds %d..%d (plain + extreme): nodes %d  undecided %d (%.3f %%)  contradictions %d  bound violations %d" % (
    first, first + count - 1, tot[0], tot[1], 100.0 * tot[1] / max(tot[0], 1), tot[2], tot[3]))
sys.exit(1 if (tot[2] or tot[3]) else 0)
