"""CPU: the stroker of the oracle (flatten.wgsl draw_cap :519-543, draw_join :545-614, offset curves) against closed-form
areas: the alpha of a stroked polyline summed over the image is the area of the stroke outline -- exact rational numbers
for butt / square caps and miter / bevel joins on axis-aligned geometry, and within the flattening tolerance where arcs
are involved (round caps / joins, a stroked circle)."""
import math

import numpy as np
import pytest

from jello_amd import Brush, Cap, Fill, Host, Join, Path, RenderParams, Scene, Stroke
from oracle.oracle_engine import OracleEngine


def stroke_area(path, stroke, size=128, transform=None):
    s = Scene()
    s.stroke(stroke, transform, Brush.solid((1, 1, 1, 1)), None, path)
    rec = Host().record(s, RenderParams(size, size))
    o = OracleEngine()
    o.run(rec)
    a = o.target(rec).view(np.float16).astype(np.float64)[..., 3]
    assert a.min() >= 0.0 and a.max() <= 1.0 + 1e-3
    return float(a.sum())


W, X0, X1, Y0, Y1 = 12.0, 20.0, 90.0, 30.0, 100.0   # (integer geometry: the box edges fall on pixel boundaries)


@pytest.mark.parametrize("cap,extra", [(Cap.Butt, 0.0), (Cap.Square, W * W), (Cap.Round, math.pi * W * W / 4.0)])
def test_caps_of_a_straight_line(built, cap, extra):
    p = Path().move_to(X0, Y0).line_to(X1, Y0)
    got = stroke_area(p, Stroke(W, Join.Miter, 4.0, cap, cap))
    want = W * (X1 - X0) + extra
    tol = 1e-2 if cap != Cap.Round else 0.25 * math.pi * W  # arcs: within the 0.25 px flattening tolerance x arc length
    assert abs(got - want) <= tol, (got, want)
    if cap == Cap.Round:
        assert got <= want + 1e-2  # chords lie inside the circle


@pytest.mark.parametrize("join,corner", [(Join.Miter, (W / 2) ** 2), (Join.Bevel, (W / 2) ** 2 / 2.0),
                                         (Join.Round, math.pi * (W / 2) ** 2 / 4.0)])
def test_joins_of_a_right_angle(built, join, corner):
    """Two legs meeting at 90 degrees, butt caps: leg rectangles overlap in a (w/2)^2 square on the inside, and the join
    fills the outside corner with a square (miter), its half (bevel) or a quarter disc (round)."""
    p = Path().move_to(X0, Y0).line_to(X1, Y0).line_to(X1, Y1)
    got = stroke_area(p, Stroke(W, join, 4.0, Cap.Butt, Cap.Butt))
    want = W * ((X1 - X0) + (Y1 - Y0)) - (W / 2) ** 2 + corner
    tol = 1e-2 if join != Join.Round else 0.25 * math.pi * W / 4 + 1e-2
    assert abs(got - want) <= tol, (join, got, want)


def test_miter_limit_falls_back_to_bevel(built):
    """A 30-degree corner needs a miter ratio of 1/sin(15 deg) = 3.86: limit 4 keeps the miter, limit 2 bevels it."""
    ang = math.radians(30.0)
    L = 60.0
    apex = (100.0, 64.0)
    p = Path().move_to(apex[0] - L, apex[1]).line_to(*apex).line_to(apex[0] - L * math.cos(ang), apex[1] - L * math.sin(ang))
    w = 6.0
    mitered = stroke_area(p, Stroke(w, Join.Miter, 4.0, Cap.Butt, Cap.Butt))
    beveled = stroke_area(p, Stroke(w, Join.Miter, 2.0, Cap.Butt, Cap.Butt))
    bevel = stroke_area(p, Stroke(w, Join.Bevel, 4.0, Cap.Butt, Cap.Butt))
    assert abs(beveled - bevel) < 1e-2
    # the miter tip beyond the bevel edge AB is the triangle ABM (A, B: outer offset points at distance h = w/2 from the apex,
    # M: intersection of the two outer offset lines, at h / cos(theta/2) on the bisector; theta = turning angle = 150 deg):
    # area = 1/2 * (2 h sin(theta/2)) * (h / cos(theta/2) - h cos(theta/2)) = h^2 sin^3(theta/2) / cos(theta/2)
    half = math.radians(150.0) / 2
    tip = (w / 2) ** 2 * math.sin(half) ** 3 / math.cos(half)
    assert abs((mitered - bevel) - tip) < 0.05, (mitered - bevel, tip)


def test_stroked_circle_is_an_annulus(built):
    r, w = 40.0, 7.0
    got = stroke_area(Path.circle(64, 64, r), Stroke(w, Join.Round, 4.0, Cap.Butt, Cap.Butt))
    want = math.pi * ((r + w / 2) ** 2 - (r - w / 2) ** 2)
    # both outlines are chords of their circles (outer: inside, inner: outside); tolerance 0.25 px each
    assert abs(got - want) < 0.25 * 2 * math.pi * (2 * r), (got, want)
    assert abs(got - want) / want < 0.01


def test_stroke_under_a_transform_scales_the_width(built):
    p = Path().move_to(10, 10).line_to(40, 10)
    a = stroke_area(p, Stroke(4.0, Join.Miter, 4.0, Cap.Butt, Cap.Butt))
    b = stroke_area(p, Stroke(4.0, Join.Miter, 4.0, Cap.Butt, Cap.Butt), transform=(2, 0, 0, 2, 0, 0))
    assert abs(a - 4.0 * 30.0) < 1e-2 and abs(b - 8.0 * 60.0) < 1e-2
