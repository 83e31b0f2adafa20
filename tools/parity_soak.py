"""Soak: many fuzz seeds (normal and extreme) through the HIP pipeline and the oracle, every buffer and the image.
usage: python3 tools/parity_soak.py [first_seed [count]]      TIGHT_LINES=1: the smallest line buffer the frame fits"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import jello_amd
from jello_amd import BumpSizes, scenes, Aa
from parity import compare

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
eng = jello_amd.Engine(0)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    for extreme in (False, True):
        s, p = scenes.scene_fuzz(seed, extreme=extreme, size=[256, 300, 512][seed % 3])
        p.bump = BumpSizes(lines=1 << 19, seg_counts=1 << 20, segments=1 << 20, tiles=1 << 21, ptcl=1 << 23, bin_data=1 << 19, blend_spill=1 << 21)
        p.aa = [Aa.Area, Aa.Msaa8, Aa.Msaa16][(seed // 3) % 3]
        if os.environ.get("TIGHT_LINES"):
            _, bump, _ = eng.render(s, p, robust=True)  # (only the robust path reads the bump allocators back)
            p.bump.lines = int(bump["lines"]) + 1  # (the smallest line buffer the frame does not overflow)
        try:
            compare(eng, s, p)
        except AssertionError as e:
            bad += 1
            print("seed %d extreme=%s %s MISMATCH: %s" % (seed, extreme, p.aa.name, str(e)[:300]), flush=True)
    if (seed - first) % 50 == 49:
        print("... %d seeds, %d mismatches, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
print("seeds %d..%d (x2 modes): mismatches %d" % (first, first + count - 1, bad))
sys.exit(1 if bad else 0)
