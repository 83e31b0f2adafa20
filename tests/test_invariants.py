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


def _cubic(p, t):
    p = np.asarray(p, np.float64)
    t = np.asarray(t, np.float64)[:, None]
    mt = 1.0 - t
    return mt ** 3 * p[0] + 3 * mt ** 2 * t * p[1] + 3 * mt * t ** 2 * p[2] + t ** 3 * p[3]


def _dist_to_curve(pts, ctrl):
    """Distance of each point to a densely sampled cubic (float64, independent of the pipeline)."""
    ts = np.linspace(0.0, 1.0, 4001)
    c = _cubic(ctrl, ts)
    d = np.sqrt(((pts[:, None, :] - c[None, :, :]) ** 2).sum(axis=2))
    return d.min(axis=1)


@pytest.mark.parametrize("seed", range(6))
def test_flatten_lines_follow_the_curve_within_tolerance(built, seed):
    """flatten.wgsl's error bound: the polyline of a filled cubic stays within the 0.25 px tolerance of the true curve
    (checked against an independent float64 evaluation of the Bezier), is connected, starts and ends in the end points;
    the two sides of its stroke run at half the line width from it (closer only where the offset curve folds)."""
    from jello_amd import Cap, Join, Stroke
    rng = np.random.default_rng(100 + seed)
    ctrl = rng.uniform(40.0, 460.0, (4, 2))
    width = float(rng.uniform(2.0, 12.0))
    s = Scene()
    p = Path().move_to(*ctrl[0]).cubic_to(*ctrl[1], *ctrl[2], *ctrl[3])
    s.fill(Fill.NonZero, None, Brush.solid((1, 0, 0, 1)), None, p)
    s.stroke(Stroke(width, Join.Bevel, 4.0, Cap.Butt, Cap.Butt), None, Brush.solid((0, 0, 1, 1)), None, p)
    rec, o = run(s, RenderParams(512, 512))
    n = int(o.get(rec, "bumpBuf", np.uint32)[7])
    raw = o.get(rec, "linesBuf", np.uint32)[:n * 6].reshape(n, 6)
    path_ix = raw[:, 0]
    xy = raw[:, 2:6].copy().view(np.float32).reshape(n, 4)
    fill = xy[path_ix == 0]
    # the fill: the cubic's polyline followed by the closing line
    poly, closing = fill[:-1], fill[-1]
    assert np.array_equal(poly[0, 0:2], ctrl[0].astype(np.float32)) and np.array_equal(poly[-1, 2:4], ctrl[3].astype(np.float32))
    assert np.array_equal(poly[1:, 0:2], poly[:-1, 2:4])            # connected: every line starts where the previous ends
    assert np.array_equal(closing[0:2], poly[-1, 2:4]) and np.array_equal(closing[2:4], poly[0, 0:2])
    verts = np.vstack([poly[:, 0:2], poly[-1:, 2:4]]).astype(np.float64)
    # two error sources, each bounded by tol = 0.25 px: cubic -> Euler spiral pieces, spiral -> chords
    assert _dist_to_curve(verts, ctrl).max() <= 0.25 + 1e-3         # vertices lie on the spiral approximation
    mids = 0.5 * (poly[:, 0:2] + poly[:, 2:4]).astype(np.float64)
    assert _dist_to_curve(mids, ctrl).max() <= 0.5 + 1e-3           # chord mid-points add the flattening error
    # the stroke: offset curves at +-width/2 (butt caps and the closing pieces are lines across the curve's ends)
    stroke = xy[path_ix == 1].astype(np.float64)
    d0 = _dist_to_curve(stroke[:, 0:2], ctrl)
    d1 = _dist_to_curve(stroke[:, 2:4], ctrl)
    d = np.concatenate([d0, d1])
    assert d.max() <= width / 2 + 0.5                               # never farther than half the width (+ tolerance)
    # ... and at half the width wherever the offset curve does not fold back (radius of curvature > width / 2)
    assert np.mean(np.abs(d - width / 2) < 0.5) > 0.6


def _supersampled_coverage(poly, size, rule, ss=8):
    """Independent coverage: winding number of ss x ss sample points per pixel against the polygon (float64)."""
    n = size * ss
    c = (np.arange(n) + 0.5) / ss
    px, py = np.meshgrid(c, c)
    wind = np.zeros((n, n), np.int32)
    pts = np.asarray(poly, np.float64)
    for a, b in zip(pts, np.roll(pts, -1, axis=0)):
        (x0, y0), (x1, y1) = a, b
        if y0 == y1:
            continue
        up = (y0 <= py) & (y1 > py)
        dn = (y1 <= py) & (y0 > py)
        t = (py - y0) / (y1 - y0)
        xi = x0 + t * (x1 - x0)
        wind += (up & (xi > px)).astype(np.int32) - (dn & (xi > px)).astype(np.int32)
    inside = (wind != 0) if rule == "nonzero" else (wind % 2 != 0)
    return inside.reshape(size, ss, size, ss).mean(axis=(1, 3))


@pytest.mark.parametrize("rule", ["nonzero", "evenodd"])
@pytest.mark.parametrize("aa", ["area", "msaa8", "msaa16"])
def test_coverage_matches_supersampling(built, rule, aa):
    """End to end (flatten ... fine): the alpha of a self-intersecting polygon filled with an opaque colour equals the
    geometric pixel coverage, computed independently by 8x8 supersampling of the winding number."""
    from jello_amd import Aa
    rng = np.random.default_rng(7)
    size = 96
    poly = rng.uniform(6.0, size - 6.0, (9, 2))      # 9 random vertices: self-intersecting star-like polygon
    p = Path().move_to(*poly[0])
    for v in poly[1:]:
        p.line_to(*v)
    p.close()
    s = Scene()
    s.fill(Fill.NonZero if rule == "nonzero" else Fill.EvenOdd, None, Brush.solid((1, 1, 1, 1)), None, p)
    params = RenderParams(size, size, aa={"area": Aa.Area, "msaa8": Aa.Msaa8, "msaa16": Aa.Msaa16}[aa])
    rec, o = run(s, params)
    alpha = f16img(o, rec).reshape(size, size, 4)[:, :, 3].astype(np.float64)
    want = _supersampled_coverage(poly.astype(np.float32).astype(np.float64), size, rule)
    err = np.abs(alpha - want)
    if aa == "area":
        # analytic area coverage is exact except where edges cross inside a pixel (signed areas of overlapping lobes are
        # summed and clamped): a handful of pixels; measured p99 0.009, mean 0.0003
        assert np.percentile(err, 99) < 0.02 and err.mean() < 0.001
    else:
        assert err.max() <= (0.25 if aa == "msaa8" else 0.18)   # 8 / 16 samples against 64; measured 0.20 / 0.14
        assert err.mean() < 0.004


@pytest.mark.parametrize("kind", ["deep", "siblings", "comb", "mixed"])
def test_clip_torture_streams_are_well_formed(built, kind):
    """scenes.scene_clip_torture on the oracle (the checker of test_gpu_parity.test_clip_torture): every tile's command stream parses to
    its END without leaving its chunks (the host's own PTCL walker), BEGIN_CLIP and END_CLIP balance in every tile, no stage failed,
    the image is finite."""
    s, p = scenes.scene_clip_torture(kind)
    p.bump = BumpSizes(ptcl=1 << 24, blend_spill=1 << 23)
    rec, o = run(s, p)
    bump = o.get(rec, "bumpBuf", np.uint32)
    assert int(bump[0]) == 0
    ptcl = o.get(rec, "ptclBuf", np.uint32)
    cfg = rec.config
    L = jello_amd.load_host()
    st = (ctypes.c_uint64 * 8)()
    assert L.jl_ptcl_stats(ptcl.ctypes.data, ptcl.size, cfg["width_in_tiles"], cfg["height_in_tiles"], st) == 0
    # walk every tile's stream: tags 10 / 11 balance, nesting never negative
    sizes = {1: 4, 3: 1, 5: 5, 6: 3, 7: 3, 8: 3, 9: 2, 10: 1, 11: 3}
    for t in range(cfg["width_in_tiles"] * cfg["height_in_tiles"]):
        pc, depth = t * 64 + 1, 0
        for _ in range(1 << 20):
            tag = int(ptcl[pc])
            if tag == 0:
                break
            if tag == 12:
                pc = int(ptcl[pc + 1])
                continue
            depth += 1 if tag == 10 else (-1 if tag == 11 else 0)
            assert depth >= 0 and tag in sizes, (t, pc, tag)
            pc += sizes[tag]
        assert depth == 0, (t, depth)
    img = f16img(o, rec)
    assert np.isfinite(img).all()
