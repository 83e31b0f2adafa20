"""CPU: gradient geometry of the oracle (draw_leaf.wgsl transforms + fine.wgsl:978-1067) against independent float64
statements of the definitions -- linear: projection on the gradient vector; radial: the HTML-canvas / PDF two-point
conical gradient (largest t with r(t) >= 0 and |p - c(t)| = r(t)); sweep: angle from the +x axis towards +y -- incl.
brush transforms and the pad / repeat / reflect extend modes.  The ramp runs black -> white, so the gradient parameter
is read back from the pixel as srgb_encode(value) (our ramps interpolate in gamma-encoded sRGB, renderer.cpp)."""
import math

import numpy as np
import pytest

from jello_amd import Brush, ColorStop, Extend, Fill, Host, Path, RenderParams, Scene
from oracle.oracle_engine import OracleEngine

SIZE = 160
STOPS = [ColorStop(0.0, (0, 0, 0, 1)), ColorStop(1.0, (1, 1, 1, 1))]


def srgb_encode(l):
    l = np.asarray(l, dtype=np.float64)
    return np.where(l <= 0.0031308, 12.92 * l, 1.055 * np.power(np.maximum(l, 1e-12), 1.0 / 2.4) - 0.055)


def render_t(brush, brush_transform=None, allow_transparent=False):
    s = Scene()
    s.fill(Fill.NonZero, None, brush, brush_transform, Path.rect(0, 0, SIZE, SIZE))
    rec = Host().record(s, RenderParams(SIZE, SIZE))
    o = OracleEngine()
    o.run(rec)
    img = o.target(rec).view(np.float16).astype(np.float64)
    if not allow_transparent:
        assert np.allclose(img[..., 3], 1.0)
        return srgb_encode(img[..., 0])
    return srgb_encode(img[..., 0]), img[..., 3]


def extend(t, mode):
    if mode == Extend.Pad:
        return np.clip(t, 0.0, 1.0)
    if mode == Extend.Repeat:
        return t - np.floor(t)
    return np.abs(t - 2.0 * np.round(0.5 * t))  # reflect


def grid(brush_transform=None):
    """Pixel positions in brush space, for the two conventions a sampler may use (corner and centre of the pixel)."""
    ys, xs = np.mgrid[0:SIZE, 0:SIZE].astype(np.float64)
    out = []
    for off in (0.0, 0.5):
        x, y = xs + off, ys + off
        if brush_transform is not None:
            a, b, c, d, e, f = brush_transform  # x' = a x + c y + e ; y' = b x + d y + f maps brush space to pixels
            det = a * d - b * c
            x, y = ((x - e) * d - (y - f) * c) / det, (-(x - e) * b + (y - f) * a) / det
        out.append((x, y))
    return out


def check(got, expect_fn, mode, tol, brush_transform=None, mask_fn=None):
    """The image must agree with the definition evaluated at the pixel corner or at the pixel centre (the WGSL's choice
    is not this test's business), within the ramp's 1/511 quantisation plus half a pixel of gradient."""
    best = None
    for x, y in grid(brush_transform):
        with np.errstate(invalid="ignore", divide="ignore"):
            t = expect_fn(x, y)
            want = extend(t, mode)
        ok = np.isfinite(want)
        if mask_fn is not None:
            ok &= mask_fn(x, y, t)
        if mode != Extend.Pad:  # stay away from the wrap / fold points, where half a pixel flips the value
            fr = t - np.floor(t)
            ok &= (fr > 0.06) & (fr < 0.94)
        err = np.abs(got - want)[ok]
        frac_bad = float(np.mean(err > tol))
        best = frac_bad if best is None else min(best, frac_bad)
    assert best < 0.002, "%.2f %% of the pixels off by more than %g" % (best * 100, tol)


@pytest.mark.parametrize("mode", [Extend.Pad, Extend.Repeat, Extend.Reflect])
def test_linear_gradient(built, mode):
    p0, p1 = (30.0, 20.0), (110.0, 70.0)
    d = (p1[0] - p0[0], p1[1] - p0[1])
    fn = lambda x, y: ((x - p0[0]) * d[0] + (y - p0[1]) * d[1]) / (d[0] ** 2 + d[1] ** 2)
    check(render_t(Brush.linear(p0, p1, STOPS, mode)), fn, mode, 0.012)


def test_linear_gradient_with_brush_transform(built):
    # a similarity (rotation + uniform scale): draw_leaf.wgsl transforms the two end points and projects in pixel space,
    # which is the brush-space definition only for such transforms (an anisotropic one shears the isolines there)
    xf = (1.3 * math.cos(0.4), 1.3 * math.sin(0.4), -1.3 * math.sin(0.4), 1.3 * math.cos(0.4), 20.0, 10.0)
    p0, p1 = (0.0, 0.0), (60.0, 30.0)
    d = (p1[0] - p0[0], p1[1] - p0[1])
    fn = lambda x, y: ((x - p0[0]) * d[0] + (y - p0[1]) * d[1]) / (d[0] ** 2 + d[1] ** 2)
    check(render_t(Brush.linear(p0, p1, STOPS, Extend.Pad), xf), fn, Extend.Pad, 0.015, xf)


def two_point(c0, r0, c1, r1):
    def fn(x, y):
        pdx, pdy = x - c0[0], y - c0[1]
        cdx, cdy = c1[0] - c0[0], c1[1] - c0[1]
        dr = r1 - r0
        a = cdx * cdx + cdy * cdy - dr * dr
        b = pdx * cdx + pdy * cdy + r0 * dr
        c = pdx * pdx + pdy * pdy - r0 * r0
        disc = b * b - a * c
        sq = np.sqrt(np.maximum(disc, 0.0))
        t_hi, t_lo = (b + sq) / a, (b - sq) / a
        if a < 0:
            t_hi, t_lo = t_lo, t_hi
        t = np.where(r0 + t_hi * dr >= 0, t_hi, t_lo)
        t = np.where((disc < 0) | (r0 + t * dr < 0), np.nan, t)
        return t
    return fn


@pytest.mark.parametrize("mode", [Extend.Pad, Extend.Reflect])
def test_radial_gradient_concentric(built, mode):
    c, r0, r1 = (80.0, 75.0), 10.0, 70.0
    fn = lambda x, y: (np.hypot(x - c[0], y - c[1]) - r0) / (r1 - r0)
    check(render_t(Brush.radial(c, r0, c, r1, STOPS, mode)), fn, mode, 0.015)


@pytest.mark.parametrize("geom", [((60.0, 70.0), 8.0, (90.0, 85.0), 75.0),     # start circle inside the end circle
                                  ((70.0, 80.0), 0.0, (95.0, 80.0), 60.0),    # focal point inside
                                  ((40.0, 60.0), 30.0, (110.0, 90.0), 45.0)])  # neither circle contains the other (a cone)
def test_radial_gradient_two_point_conical(built, geom):
    c0, r0, c1, r1 = geom
    fn = two_point(c0, r0, c1, r1)
    got, alpha = render_t(Brush.radial(c0, r0, c1, r1, STOPS, Extend.Pad), allow_transparent=True)
    check(got, fn, Extend.Pad, 0.02, mask_fn=lambda x, y, t: np.isfinite(t) & (alpha > 0.999))
    # pixels the cone does not cover are transparent (canvas definition), all others opaque -- up to the boundary pixels
    x, y = grid()[1]
    with np.errstate(invalid="ignore", divide="ignore"):
        covered = np.isfinite(fn(x, y))
    disagree = np.mean((alpha > 0.5) != covered)
    assert disagree < 0.02, disagree
    if geom[1] == 30.0:
        assert (~covered).mean() > 0.1  # this geometry really has an uncovered region


def test_radial_gradient_with_anisotropic_brush_transform(built):
    # radial gradients invert the transform (draw_leaf.wgsl), so the brush-space definition holds for any affine map
    xf = (1.5 * math.cos(0.4), 1.5 * math.sin(0.4), -0.8 * math.sin(0.4), 0.8 * math.cos(0.4), 70.0, 40.0)
    c0, r0, c1, r1 = (5.0, 10.0), 4.0, (15.0, 20.0), 55.0
    got = render_t(Brush.radial(c0, r0, c1, r1, STOPS, Extend.Pad), xf)
    check(got, two_point(c0, r0, c1, r1), Extend.Pad, 0.02, xf, mask_fn=lambda x, y, t: np.isfinite(t))


@pytest.mark.parametrize("mode", [Extend.Pad, Extend.Repeat])
def test_sweep_gradient(built, mode):
    c, a0, a1 = (80.0, 80.0), 0.5, 4.0  # radians
    def fn(x, y):
        th = np.arctan2(y - c[1], x - c[0])
        th = np.where(th < 0, th + 2 * math.pi, th)
        return (th - a0) / (a1 - a0)
    # away from the centre (the angle changes fast there) and from the +x axis, where the angle wraps
    far = lambda x, y, t: (np.hypot(x - c[0], y - c[1]) > 25) & ~((np.abs(y - c[1]) < 2.0) & (x > c[0]))
    check(render_t(Brush.sweep(c, a0, a1, STOPS, mode)), fn, mode, 0.02, mask_fn=far)
