"""-m gpu: misuse of the C ABI returns negative jh_status codes (never aborts, never faults): unknown ids, wrong binding
counts, bad stages, out-of-range transfers, a band with row1 < row0 -- the reference panics in these cases
(wgpu.go:77,213,282,544,558,594,955); SURVEY 8b asks for error codes."""
import ctypes

import numpy as np
import pytest

import jello_amd
from jello_amd import scenes

pytestmark = pytest.mark.gpu


class Binding(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("count", ctypes.c_uint32), ("id", ctypes.c_uint64), ("ids", ctypes.POINTER(ctypes.c_uint64))]


def test_misuse_returns_error_codes(engine):
    hip, ctx = engine.hip, engine.ctx
    hip.jh_dispatch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(Binding), ctypes.c_int]
    hip.jh_free.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    hip.jh_image_download.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    unknown = 0xdead0000beef
    buf = np.zeros(64, np.uint8)
    assert hip.jh_download(ctx, unknown, buf.ctypes.data, 0, 64) < 0             # unknown buffer id
    assert hip.jh_clear(ctx, unknown, 0, -1) < 0
    assert hip.jh_image_download(ctx, unknown, buf.ctypes.data, 64) < 0
    assert hip.jh_set_band(ctx, 5, 2) < 0                                        # row1 < row0
    assert hip.jh_upload(ctx, 0x1234, buf.ctypes.data, 64) == 0
    assert hip.jh_download(ctx, 0x1234, buf.ctypes.data, 32, 64) < 0             # range past the end
    one = (Binding * 1)(Binding(1, 0, 0x1234, None))
    assert hip.jh_dispatch(ctx, 99, 1, 1, 1, one, 1) < 0                         # no such stage
    assert hip.jh_dispatch(ctx, -1, 1, 1, 1, one, 1) < 0
    for stage in range(22):                                                      # every stage needs more than one binding
        assert hip.jh_dispatch(ctx, stage, 1, 1, 1, one, 1) < 0, stage
    bad = (Binding * 2)(Binding(1, 0, unknown, None), Binding(1, 0, 0x1234, None))
    assert hip.jh_dispatch(ctx, 0, 1, 1, 1, bad, 2) < 0                          # unknown id in the binding list
    assert hip.jh_free(ctx, 0x1234) == 0
    assert len(hip.jh_last_error(ctx)) > 0
    # the context is still usable
    s, p = scenes.scene_c1()
    rec, bump, attempts = engine.render(s, p, retain=False)
    assert bump["failed"] == 0


def test_binning_and_coarse_refuse_more_than_256_bins(engine):
    """A ConfigUniform whose target has more than 256 bins (binning.wgsl:52,131 index them in a 256-entry table): the dispatch
    returns an error instead of producing a frame that misses the bins past the 256th."""
    hip, ctx = engine.hip, engine.ctx
    hip.jh_dispatch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(Binding), ctypes.c_int]
    hip.jh_upload.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    hip.jh_free.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    cfg = np.zeros(25, np.uint32)  # a ConfigUniform is recognised by its size (100 bytes): the launchers read its host shadow
    cfg[0], cfg[1] = 257, 256   # width / height in tiles: 17 x 16 = 272 bins
    ids = [0x5b100 + i for i in range(9)]
    buf = np.zeros(4096, np.uint8)
    try:
        assert hip.jh_upload(ctx, ids[0], cfg.ctypes.data, cfg.nbytes) == 0
        for i in ids[1:]:
            assert hip.jh_upload(ctx, i, buf.ctypes.data, buf.nbytes) == 0
        b8 = (Binding * 8)(*[Binding(1, 0, i, None) for i in ids[:8]])
        b9 = (Binding * 9)(*[Binding(1, 0, i, None) for i in ids[:9]])
        assert hip.jh_dispatch(ctx, 11, 1, 1, 1, b8, 8) < 0    # JH_BINNING
        assert b"bins" in hip.jh_last_error(ctx)
        assert hip.jh_dispatch(ctx, 16, 17, 16, 1, b9, 9) < 0  # JH_COARSE
        cfg[0] = 256
        assert hip.jh_upload(ctx, ids[0], cfg.ctypes.data, cfg.nbytes) == 0
        assert hip.jh_dispatch(ctx, 11, 1, 1, 1, b8, 8) == 0   # 256 bins: accepted (n_drawobj = 0: nothing to bin)
        engine.sync()
    finally:
        for i in ids:
            hip.jh_free(ctx, i)
