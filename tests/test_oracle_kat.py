"""CPU: the oracle and the host encoder against the hand-derived known answers of SURVEY Appendix D
(tests/golden/c1_kat.json).  The reference ships no fixtures of its own, so these KATs --
derived by reading the reference encoder and WGSL -- are what pins the oracle."""
import ctypes
import json
import os

import numpy as np
import pytest

import jello_amd
from jello_amd import scenes
from oracle import oracle_engine
from oracle.oracle_engine import OracleEngine

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "c1_kat.json")))


@pytest.fixture(scope="module")
def c1(built):
    s, p = scenes.scene_c1()
    rec = jello_amd.Host().record(s, p)
    o = OracleEngine()
    o.run(rec)
    return s, p, rec, o


def test_encoder_streams(c1):
    s = c1[0]
    assert s.stream("path_tags").hex(" ") == KAT["path_tags_hex"]
    assert list(np.frombuffer(s.stream("path_data"), np.float32)) == KAT["path_data_f32"]
    assert list(np.frombuffer(s.stream("draw_tags"), np.uint32)) == KAT["draw_tags"]
    assert list(np.frombuffer(s.stream("draw_data"), np.float32)) == KAT["draw_data_f32"]
    assert list(np.frombuffer(s.stream("transforms"), np.float32)) == KAT["transforms_f32"]
    assert [hex(x) for x in np.frombuffer(s.stream("styles"), np.uint32)] == [hex(int(x, 16)) for x in KAT["styles_u32"]]
    for val, bits in KAT["f32_bits"].items():
        assert np.float32(float(val)).view(np.uint32) == int(bits, 16)


def test_layout_config_and_dispatch_geometry(c1):
    rec = c1[2]
    cfg = rec.config
    for k, v in KAT["layout"].items():
        assert cfg[k] == v, k
    for k, v in KAT["config"].items():
        assert cfg[k] == v
    scene_cmd = rec.commands()[1]
    assert scene_cmd["buf_name"] == "scene" and len(scene_cmd["data"]) == KAT["scene_bytes"]
    words = np.frombuffer(scene_cmd["data"], np.uint32)
    assert [hex(w) for w in words[:3]] == [hex(int(x, 16)) for x in KAT["path_tag_words"]]
    wg = rec.workgroup_counts()
    for k, v in KAT["workgroup_counts"].items():
        if isinstance(v, list):
            assert list(wg[k][:2]) == v, k
        elif isinstance(v, bool):
            assert wg[k] == v
        else:
            assert wg[k][0] == v, k


def test_reduce_tag_and_draw_tag_kats(built):
    L = oracle_engine.lib()
    for w, want in KAT["reduce_tag"].items():
        out = (ctypes.c_uint32 * 5)()
        L.oracle_reduce_tag(ctypes.c_uint32(int(w, 16)), out)
        assert list(out) == want, w
    for t, want in KAT["map_draw_tag"].items():
        out = (ctypes.c_uint32 * 4)()
        L.oracle_map_draw_tag(ctypes.c_uint32(int(t, 16)), out)
        assert list(out) == want, t


def test_tag_monoids(c1):
    _, _, rec, o = c1
    tm = o.get(rec, "tagmonoidBuf", np.uint32).reshape(-1, 5)
    assert tm[:3].tolist() == KAT["tag_monoids_exclusive"]


def test_rect_flattens_to_four_exact_lines(c1):
    _, _, rec, o = c1
    lines = o.get(rec, "linesBuf", np.uint32).reshape(-1, 6)
    assert lines[:4, 0].tolist() == [0, 0, 0, 0]
    assert lines[:4, 2:].view(np.float32).tolist() == KAT["rect_lines"]
    assert lines[4, 0] == 1  # the next line already belongs to the stroke
    pb = o.get(rec, "pathBboxBuf", np.int32).reshape(-1, 6)
    assert pb[0, :4].tolist() == KAT["rect_path_bbox"] and pb[0, 4] == 0 and pb[0, 5] == 0


def test_tile_alloc_of_rect(c1):
    _, _, rec, o = c1
    paths = o.get(rec, "pathBuf", np.uint32).reshape(-1, 8)
    assert paths[0, :4].tolist() == KAT["rect_tiles"]["bbox"]
    assert paths[0, 4] == KAT["rect_tiles"]["offset"]
    assert paths[1, 4] == KAT["rect_tiles"]["count"]  # the stroke's tiles start right after the rect's 130


def test_rect_fill_exact_area(built):
    """Filled rect alone: interior un-premultiplied (1,0,0,1); column x=9 empty, x=10 full; sum(alpha) = 190*140."""
    s = jello_amd.Scene()
    s.fill(jello_amd.Fill.NonZero, None, jello_amd.Brush.solid((1, 0, 0, 1)), None, jello_amd.Path.rect(10, 10, 200, 150))
    rec = jello_amd.Host().record(s, jello_amd.RenderParams(512, 512))
    o = OracleEngine()
    o.run(rec)
    img = o.target(rec).view(np.float16).astype(np.float64)
    assert img[..., 3].sum() == KAT["rect_fill_alpha_sum"]
    assert tuple(img[80, 100]) == (1.0, 0.0, 0.0, 1.0)
    assert img[80, 9, 3] == 0.0 and img[80, 10, 3] == 1.0 and img[9, 100, 3] == 0.0 and img[10, 100, 3] == 1.0
    assert img[149, 199, 3] == 1.0 and img[150, 199, 3] == 0.0 and img[149, 200, 3] == 0.0


def test_float16_and_span_kats(built):
    L = oracle_engine.lib()
    L.oracle_f32_to_f16.restype = ctypes.c_uint16
    L.oracle_f32_to_f16.argtypes = [ctypes.c_float]
    L.oracle_f16_to_f32.restype = ctypes.c_float
    L.oracle_f16_to_f32.argtypes = [ctypes.c_uint16]
    L.oracle_span.restype = ctypes.c_uint32
    L.oracle_span.argtypes = [ctypes.c_float, ctypes.c_float]
    for v, bits in KAT["float16_bits"].items():
        assert L.oracle_f32_to_f16(float(v)) == int(bits, 16)
        assert L.oracle_f16_to_f32(int(bits, 16)) == float(v)
    for a, b, n in KAT["span"]:
        assert L.oracle_span(a, b) == n
    # every binary16 value round-trips, and conversion agrees with numpy's RTNE on random data
    allh = np.arange(0x10000, dtype=np.uint16)
    finite = allh[(allh & 0x7c00) != 0x7c00]
    for h in finite[::97]:
        assert L.oracle_f32_to_f16(L.oracle_f16_to_f32(int(h))) == h
    x = (np.random.default_rng(3).standard_normal(20000) * 10 ** np.random.default_rng(4).uniform(-9, 5, 20000)).astype(np.float32)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    got = np.array([L.oracle_f32_to_f16(float(v)) for v in x], dtype=np.uint16)
    assert np.array_equal(got, want)
    # the stroke style word of C1: stroke | miter | f16(4.0) (encoding/path.go:86-120)
    assert int(KAT["styles_u32"][2], 16) == 0x80000000 | 0x10000000 | 0x4400


def test_image_brush_srgb_decode_kat():
    """A constant opaque sRGB image under an identity brush transform: every covered pixel is the decoded
    linear colour, c <= 0.04045 ? c/12.92 : ((c+0.055)/1.055)^2.4 of the 8-bit code (rgba8unorm-srgb)."""
    import jello_amd
    from jello_amd import Brush, Fill, Path, RenderParams, Scene
    from oracle import oracle_engine
    px = np.zeros((8, 8, 4), np.uint8)
    px[:, :] = (188, 64, 10, 255)
    s = Scene()
    s.fill(Fill.NonZero, None, Brush.image(px, key=7), (1, 0, 0, 1, 16, 16), Path.rect(16, 16, 24, 24))
    p = RenderParams(32, 32)
    rec = jello_amd.Host().record(s, p)
    eng = oracle_engine.OracleEngine()
    eng.run(rec)
    img = eng.target(rec).view(np.float16).astype(np.float64).reshape(32, 32, 4)
    def lin(c):
        c = c / 255.0
        return c / 12.92 if c <= 0.04045 else ((c + 0.055) / 1.055) ** 2.4
    want = np.array([lin(188), lin(64), lin(10), 1.0])
    got = img[19, 19]  # interior pixel: all four bilinear taps inside the image
    assert np.all(np.abs(got - want) <= 2e-3 * np.maximum(want, 1e-3)), (got, want)
    assert tuple(img[5, 5]) == (0.0, 0.0, 0.0, 0.0)


@pytest.mark.parametrize("samples", [8, 16])
@pytest.mark.parametrize("rule", ["nonzero", "evenodd"])
def test_msaa_rect_coverage_kat(samples, rule):
    """Half-plane masks of the D3D11 sample patterns (mask.go:43-105): a rectangle whose left edge sits at x+0.5
    and whose bottom edge sits at y+0.25 covers exactly 1/2 resp. 1/4 of the samples of the boundary pixels."""
    import jello_amd
    from jello_amd import Aa, Brush, Fill, Path, RenderParams, Scene
    from oracle import oracle_engine
    s = Scene()
    s.fill(Fill.NonZero if rule == "nonzero" else Fill.EvenOdd, None, Brush.solid((0, 1, 0, 1)), None, Path.rect(16.5, 8, 40, 24.25))
    p = RenderParams(64, 32, aa=Aa.Msaa8 if samples == 8 else Aa.Msaa16)
    rec = jello_amd.Host().record(s, p)
    eng = oracle_engine.OracleEngine()
    eng.run(rec)
    img = eng.target(rec).view(np.float16).astype(np.float32).reshape(32, 64, 4)
    assert tuple(img[12, 20]) == (0.0, 1.0, 0.0, 1.0)   # interior
    assert img[12, 16, 3] == 0.5                          # left edge pixel: half covered
    assert img[24, 20, 3] == 0.25                         # bottom edge pixel: quarter covered
    assert img[24, 16, 3] == 0.125                        # corner: 1/2 * 1/4
    assert img[4, 4, 3] == 0.0 and img[12, 41, 3] == 0.0  # outside
