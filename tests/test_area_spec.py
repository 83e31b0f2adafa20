"""CPU: filled area of the oracle (flatten + path_count + path_tiling + fine, area AA) against closed forms: the sum of
alpha over the image is the area of the shape -- shoelace for polygons (exact up to f32 rounding of the trapezoid sums),
pi r^2 and a numerically integrated Green's formula for curves (within the 0.25 px flattening tolerance)."""
import math

import numpy as np
import pytest

from jello_amd import Brush, Fill, Host, Path, RenderParams, Scene
from oracle.oracle_engine import OracleEngine


def fill_area(path, rule=Fill.NonZero, size=128, transform=None, aa=None):
    s = Scene()
    s.fill(rule, transform, Brush.solid((1, 1, 1, 1)), None, path)
    p = RenderParams(size, size)
    if aa is not None:
        p.aa = aa
    rec = Host().record(s, p)
    o = OracleEngine()
    o.run(rec)
    return float(o.target(rec).view(np.float16).astype(np.float64)[..., 3].sum())


def shoelace(pts):
    return 0.5 * abs(sum(pts[i][0] * pts[(i + 1) % len(pts)][1] - pts[(i + 1) % len(pts)][0] * pts[i][1] for i in range(len(pts))))


def poly(pts):
    p = Path().move_to(*pts[0])
    for q in pts[1:]:
        p.line_to(*q)
    return p.close()


@pytest.mark.parametrize("pts", [[(10.25, 12.5), (100.75, 30.125), (40.5, 110.875)],
                                 [(20.3, 20.7), (90.1, 15.2), (115.6, 70.9), (70.4, 118.3), (12.9, 80.6)],
                                 [(64.0, 5.5), (75.1, 48.2), (120.3, 50.9), (83.2, 74.4), (99.7, 119.8), (64.0, 90.0), (28.3, 119.8),
                                  (44.8, 74.4), (7.7, 50.9), (52.9, 48.2)]])  # (a star: simple polygon, winding 1 everywhere inside)
def test_polygon_area_is_the_shoelace_area(built, pts):
    got, want = fill_area(poly(pts)), shoelace(pts)
    assert abs(got - want) < 2e-3 * want ** 0.5 + 0.02, (got, want)  # f16 alpha rounding of the boundary pixels


def test_circle_and_ellipse_area(built):
    r = 45.0
    got = fill_area(Path.circle(64, 64, r))
    # chords of a convex outline lie inside it, at most the 0.25 px flattening tolerance away
    assert math.pi * r * r - 0.25 * 2 * math.pi * r <= got <= math.pi * r * r + 0.05
    got = fill_area(Path.circle(0, 0, 20.0), transform=(2.0, 0, 0, 1.5, 64, 64))  # ellipse 40 x 30 (tolerance is in pixels)
    perimeter = math.pi * (3 * (40 + 30) - math.sqrt((3 * 40 + 30) * (40 + 3 * 30)))  # Ramanujan
    assert math.pi * 40 * 30 - 0.25 * perimeter <= got <= math.pi * 40 * 30 + 0.05


def test_cubic_segment_area_by_greens_formula(built):
    p0, p1, p2, p3 = (15.0, 100.0), (20.0, -20.0), (110.0, 10.0), (105.0, 105.0)
    def pt(t):
        u = 1 - t
        return (u ** 3 * p0[0] + 3 * u * u * t * p1[0] + 3 * u * t * t * p2[0] + t ** 3 * p3[0],
                u ** 3 * p0[1] + 3 * u * u * t * p1[1] + 3 * u * t * t * p2[1] + t ** 3 * p3[1])
    ts = np.linspace(0.0, 1.0, 200001)
    xs, ys = pt(ts)
    xs, ys = np.append(xs, p0[0]), np.append(ys, p0[1])          # closing line back to the start
    want = 0.5 * abs(np.sum(xs[:-1] * ys[1:] - xs[1:] * ys[:-1]))  # shoelace of a 200 k-gon = Green's integral
    p = Path().move_to(*p0).cubic_to(*p1, *p2, *p3).close()
    got = fill_area(p)
    arc = float(np.sum(np.hypot(np.diff(xs[:-1]), np.diff(ys[:-1]))))
    assert want - 0.25 * arc <= got <= want + 0.05, (got, want)  # (this cubic is convex: chords inside)


def test_even_odd_hole(built):
    outer = [(10, 10), (118, 10), (118, 118), (10, 118)]
    inner = [(40, 40), (40, 90), (90, 90), (90, 40)]  # (either orientation: even-odd ignores it)
    p = poly(outer)
    p.move_to(*inner[0])
    for q in inner[1:]:
        p.line_to(*q)
    p.close()
    assert abs(fill_area(p, Fill.EvenOdd) - (108 * 108 - 50 * 50)) < 1e-2
    same = poly(outer)  # inner contour with the SAME orientation as the outer one: non-zero fills the hole
    for i, q in enumerate([inner[0], inner[3], inner[2], inner[1]]):
        same.move_to(*q) if i == 0 else same.line_to(*q)
    same.close()
    assert abs(fill_area(same, Fill.NonZero) - 108 * 108) < 1e-2
