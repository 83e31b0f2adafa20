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
