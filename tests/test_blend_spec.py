"""CPU: the oracle's layer compositing against an independent float64 statement of the W3C "Compositing and Blending
Level 1" formulas (written from the specification, not from blend.wgsl): all 16 blend modes with source-over, and the
Porter-Duff operators with normal blending, on partially transparent backdrops and sources.  This pins
blend.wgsl:142-310 as restated in oracle.cpp (and, through the GPU parity tests, the HIP kernel) to a published
standard; the tolerance is the RGBA16F output quantisation."""
import numpy as np
import pytest

from jello_amd import Brush, Compose, Fill, Host, Mix, Path, RenderParams, Scene
from oracle.oracle_engine import OracleEngine


def lum(c):
    return 0.3 * c[0] + 0.59 * c[1] + 0.11 * c[2]


def clip_color(c):
    l, n, x = lum(c), min(c), max(c)
    c = list(c)
    if n < 0.0:
        c = [l + (v - l) * l / (l - n) for v in c]
    if x > 1.0:
        c = [l + (v - l) * (1.0 - l) / (x - l) for v in c]
    return c


def set_lum(c, l):
    d = l - lum(c)
    return clip_color([v + d for v in c])


def sat(c):
    return max(c) - min(c)


def set_sat(c, s):
    idx = sorted(range(3), key=lambda i: c[i])  # min, mid, max
    out = [0.0, 0.0, 0.0]
    cmin, cmid, cmax = c[idx[0]], c[idx[1]], c[idx[2]]
    if cmax > cmin:
        out[idx[1]] = (cmid - cmin) * s / (cmax - cmin)
        out[idx[2]] = s
    return out


def blend_w3c(mode, cb, cs):
    """B(Cb, Cs) of https://www.w3.org/TR/compositing-1/#blending (separable modes per channel)."""
    def sep(f):
        return [f(b, s) for b, s in zip(cb, cs)]

    def hard_light(b, s):
        return b * 2 * s if s <= 0.5 else b + (2 * s - 1) - b * (2 * s - 1)

    def soft_light(b, s):
        if s <= 0.5:
            return b - (1 - 2 * s) * b * (1 - b)
        d = ((16 * b - 12) * b + 4) * b if b <= 0.25 else np.sqrt(b)
        return b + (2 * s - 1) * (d - b)

    def dodge(b, s):
        return 0.0 if b == 0 else (1.0 if s == 1 else min(1.0, b / (1 - s)))

    def burn(b, s):
        return 1.0 if b == 1 else (0.0 if s == 0 else 1.0 - min(1.0, (1 - b) / s))

    m = Mix(mode)
    if m == Mix.Normal: return list(cs)
    if m == Mix.Multiply: return sep(lambda b, s: b * s)
    if m == Mix.Screen: return sep(lambda b, s: b + s - b * s)
    if m == Mix.Overlay: return sep(lambda b, s: hard_light(s, b))
    if m == Mix.Darken: return sep(min)
    if m == Mix.Lighten: return sep(max)
    if m == Mix.ColorDodge: return sep(dodge)
    if m == Mix.ColorBurn: return sep(burn)
    if m == Mix.HardLight: return sep(hard_light)
    if m == Mix.SoftLight: return sep(soft_light)
    if m == Mix.Difference: return sep(lambda b, s: abs(b - s))
    if m == Mix.Exclusion: return sep(lambda b, s: b + s - 2 * b * s)
    if m == Mix.Hue: return set_lum(set_sat(cs, sat(cb)), lum(cb))
    if m == Mix.Saturation: return set_lum(set_sat(cb, sat(cs)), lum(cb))
    if m == Mix.Color: return set_lum(cs, lum(cb))
    return set_lum(cb, lum(cs))  # Luminosity


def porter_duff(op, ab, as_):
    """(Fa, Fb) of https://www.w3.org/TR/compositing-1/#porterduffcompositingoperators."""
    return {Compose.SrcOver: (1, 1 - as_), Compose.Copy: (1, 0), Compose.Dest: (0, 1), Compose.Clear: (0, 0),
            Compose.DestOver: (1 - ab, 1), Compose.SrcIn: (ab, 0), Compose.DestIn: (0, as_), Compose.SrcOut: (1 - ab, 0),
            Compose.DestOut: (0, 1 - as_), Compose.SrcAtop: (ab, 1 - as_), Compose.DestAtop: (1 - ab, as_),
            Compose.Xor: (1 - ab, 1 - as_), Compose.Plus: (1, 1)}[op]


def expected(mix, compose, backdrop, source, layer_alpha):
    cb, ab = list(backdrop[:3]), backdrop[3]
    cs, as_ = list(source[:3]), source[3] * layer_alpha
    cs_mixed = [(1 - ab) * s + ab * m for s, m in zip(cs, blend_w3c(mix, cb, cs))]
    fa, fb = porter_duff(compose, ab, as_)
    co = [as_ * fa * s + ab * fb * b for s, b in zip(cs_mixed, cb)]
    ao = min(1.0, as_ * fa + ab * fb)
    return [c / max(ao, 1e-6) for c in co] + [ao]  # fine.wgsl stores un-premultiplied colour


def render_pixel(mix, compose, backdrop, source, layer_alpha):
    s = Scene()
    s.fill(Fill.NonZero, None, Brush.solid(backdrop), None, Path.rect(0, 0, 32, 32))
    s.push_layer(mix, compose, layer_alpha, None, Path.rect(0, 0, 32, 32))
    s.fill(Fill.NonZero, None, Brush.solid(source), None, Path.rect(0, 0, 32, 32))
    s.pop_layer()
    rec = Host().record(s, RenderParams(32, 32))  # transparent base colour
    o = OracleEngine()
    o.run(rec)
    return o.target(rec).view(np.float16).astype(np.float64)[16, 16]


PAIRS = [((0.8, 0.3, 0.1, 1.0), (0.2, 0.6, 0.9, 1.0)),
         ((0.25, 0.5, 0.75, 0.6), (0.9, 0.1, 0.4, 0.7)),
         ((0.1, 0.1, 0.1, 0.9), (0.95, 0.9, 0.2, 0.5)),
         ((0.6, 0.2, 0.7, 1.0), (0.3, 0.3, 0.3, 1.0))]


@pytest.mark.parametrize("mix", [m for m in Mix if m != Mix.Clip])
def test_blend_modes_follow_the_w3c_formulas(built, mix):
    for backdrop, source in PAIRS:
        for layer_alpha in (1.0, 0.5):
            got = render_pixel(mix, Compose.SrcOver, backdrop, source, layer_alpha)
            want = expected(mix, Compose.SrcOver, backdrop, source, layer_alpha)
            assert np.allclose(got, want, atol=2.5e-3), (mix.name, backdrop, source, layer_alpha, got, want)


@pytest.mark.parametrize("compose", [c for c in Compose if c != Compose.PlusLighter])
def test_porter_duff_operators(built, compose):
    for backdrop, source in PAIRS[1:3]:
        got = render_pixel(Mix.Normal, compose, backdrop, source, 0.8)
        want = expected(Mix.Normal, compose, backdrop, source, 0.8)
        if want[3] < 1e-3:  # fully transparent result: colour is undefined
            assert abs(got[3]) < 1e-3
        else:
            assert np.allclose(got, want, atol=2.5e-3), (compose.name, backdrop, source, got, want)
