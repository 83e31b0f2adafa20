"""CPU: the oracle's parallel forms of the three stages that allocate in canonical order (flatten, path_count, coarse:
count -> scan -> write over chunks; oracle.cpp, g_oracle_parallel_alloc) exist only so that bench.py's cpu_baseline can
use every core.  They must produce exactly the serial oracle's buffers and image."""
import numpy as np
import pytest

import jello_amd
from jello_amd import BumpSizes, scenes
from oracle import oracle_engine
from oracle.oracle_engine import OracleEngine

NAMES = [("bumpBuf", np.uint32), ("linesBuf", np.uint32), ("pathBboxBuf", np.uint32), ("segCountsBuf", np.uint32), ("tileBuf", np.uint32),
         ("segmentsBuf", np.uint32), ("ptclBuf", np.uint32)]


@pytest.mark.parametrize("which", ["c3", "c4", "c4n"])
def test_parallel_allocation_equals_serial(built, which):
    s, p = {"c3": lambda: scenes.scene_c3(3000, 512), "c4": lambda: scenes.scene_c4(1500, 512),
            "c4n": lambda: scenes.scene_c4_nested(1500, 512)}[which]()
    p.bump = BumpSizes(lines=1 << 20, seg_counts=1 << 20, segments=1 << 20, tiles=1 << 20, ptcl=1 << 23, bin_data=1 << 19, blend_spill=1 << 16)
    rec = jello_amd.Host().record(s, p)
    L = oracle_engine.lib()
    try:
        L.oracle_set_threads(1)
        L.oracle_set_parallel_alloc(0)
        a = OracleEngine(poison=0)  # (unwritten tails compare equal: both runs start from zeroed buffers)
        a.run(rec)
        L.oracle_set_threads(4)
        L.oracle_set_parallel_alloc(1)
        b = OracleEngine(poison=0)
        b.run(rec)
    finally:
        L.oracle_set_threads(1)
        L.oracle_set_parallel_alloc(0)
    assert a.get(rec, "bumpBuf", np.uint32)[0] == 0
    for name, dt in NAMES:
        assert np.array_equal(a.get(rec, name, dt), b.get(rec, name, dt)), name
    assert np.array_equal(a.target(rec), b.target(rec))
