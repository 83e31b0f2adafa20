"""-m gpu: the fine stage ALONE on segments no earlier stage would write -- far outside their tile, NaN, of zero length, on
pixel corners and on the tile's left edge -- in all three coverage modes, against the oracle on the same bytes.  The pipeline up to
path_tiling runs normally on both sides; the segment buffer is then overwritten identically and only fine is dispatched again
(RUN_ONLY_FINE / OracleEngine.run(only=...)).  This is what reaches the routes of the multisampled kernel that a frame from
path_tiling never takes: a segment with more touched pixels than the batch list holds and one whose first sample mask depends on
the fill rule are walked at the fill (kernels_fine.hip, MsState::direct), next to segments that go through the list."""
import ctypes

import numpy as np
import pytest

import jello_amd
from jello_amd import scenes
from jello_amd.engine import RUN_DISPATCHES, RUN_ONLY_FINE, RUN_UPLOADS
from oracle.oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu


def _damage(seg, seed):
    """seg: (n, 6) float32 view of the Segment buffer (p0x p0y p1x p1y y_edge pad).  Every 5th segment is replaced."""
    rng = np.random.default_rng(seed)
    # (no infinities and nothing beyond a few thousand pixels: the reference walks EVERY touched pixel of a segment -- 2^32 of them for
    # an infinite span -- and so do the oracle and the kernel's walk at the fill)
    nan = np.float32(np.nan)
    kinds = [
        lambda s: (s[0], s[1], s[2] + 300.0, s[3]),                   # hundreds of touched pixels: does not fit the list
        lambda s: (s[0] - 1000.0, s[1] - 700.0, s[2] + 900.0, s[3] + 800.0),
        lambda s: (0.0, np.floor(s[1]), s[2], s[3]),                   # starts on the left edge AND on a pixel row
        lambda s: (0.0, np.floor(s[1]), s[2] - 500.0, s[3] + 40.0),
        lambda s: (nan, s[1], s[2], s[3]),
        lambda s: (s[0], nan, s[2], nan),
        lambda s: (s[0], s[1], s[0], s[1]),                           # zero length
        lambda s: (np.floor(s[0]), np.floor(s[1]), np.floor(s[0]) + 3.0, np.floor(s[1])),   # along a pixel row
        lambda s: (np.floor(s[0]), s[1], np.floor(s[0]), s[3]),       # along a pixel column
        lambda s: (16.0, s[1], s[2], 16.0),                           # the tile's right / bottom edge
        lambda s: (s[0], s[1], 0.0, np.floor(s[3])),                  # ends on the left edge on a row
        lambda s: (-0.0, s[1], s[2], -0.0),
        lambda s: (s[0] * 90.0, s[1], s[2], s[3] * -70.0),
        lambda s: (0.0, 0.0, 0.0, 0.0),                               # what a read behind the buffer gives
    ]
    for i in range(0, seg.shape[0], 5):
        f = kinds[int(rng.integers(len(kinds)))]
        with np.errstate(all="ignore"):
            seg[i, 0:4] = np.array(f(seg[i, 0:4].copy()), dtype=np.float32)
        if rng.integers(4) == 0:
            seg[i, 4] = np.float32([0.0, 3.5, 16.0, 1e9, -2.0, np.nan][int(rng.integers(6))])   # y_edge (the area mode reads it)


@pytest.mark.parametrize("aa", [jello_amd.Aa.Area, jello_amd.Aa.Msaa8, jello_amd.Aa.Msaa16])
@pytest.mark.parametrize("which", ["c2", "c3"])
def test_fine_alone_on_damaged_segments(engine, aa, which):
    if which == "c2":
        s, p = scenes.scene_c2(120, 512)   # even-odd and non-zero fills, all joins and caps
    else:
        s, p = scenes.scene_c3(1500, 512)
    p.aa = aa
    host = jello_amd.Host()
    rec = host.record(s, p)
    o = OracleEngine()
    o.run(rec)
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    engine.sync()
    seg_id, seg_size = rec.buffer("segmentsBuf")
    seg = o.bufs[seg_id].view(np.float32)
    n = seg.size // 6
    used = int(o.get(rec, "bumpBuf", np.uint32)[5])  # bump.segments
    assert 0 < used <= n
    _damage(seg[:n * 6].reshape(n, 6)[:used], 1234 + int(aa.value if hasattr(aa, "value") else aa))
    raw = o.bufs[seg_id]
    engine.hip.jh_upload.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    assert engine.hip.jh_upload(engine.ctx, seg_id, raw.ctypes.data, raw.nbytes) == 0
    fine_stage = {jello_amd.Aa.Area: "fine_area", jello_amd.Aa.Msaa8: "fine_msaa8", jello_amd.Aa.Msaa16: "fine_msaa16"}[aa]
    o.run(rec, only=fine_stage)
    engine.run(rec, RUN_DISPATCHES | RUN_ONLY_FINE)
    engine.sync()
    t = rec.target
    got = engine.download_image(t["id"], t["width"], t["height"]).view(np.uint16).reshape(t["height"], t["width"], 4)
    want = o.target(rec).view(np.uint16).reshape(t["height"], t["width"], 4)

    def canon(x):  # (0/0 is -NaN on x86 and +NaN on gfx950: DESIGN 5)
        f = x.view(np.float16)
        x = x.copy()
        x[np.isnan(f)] = 0x7e00
        return x
    bad = np.argwhere((canon(got) != canon(want)).any(axis=2))
    assert bad.shape[0] == 0, "%d pixels differ, first at (y, x) = %s: gpu %s oracle %s" % (
        bad.shape[0], tuple(bad[0]), got[tuple(bad[0])], want[tuple(bad[0])])
    engine.release(rec)
