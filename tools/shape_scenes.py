"""Synthetic scenes outside the BASELINE configs, used to look for scaling cliffs: large shapes, long lines, very many tiny
paths, one path with a very long tag stream (tools/time_shapes.py times them, tools/prof_scene.py profiles one)."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from jello_amd import scenes, BumpSizes
from jello_amd.scene import Scene, Path, Brush, Fill, Stroke, RenderParams
from jello_amd.scenes import SplitMix64


def scene_shapes(n, rmin, rmax, size, seed=77):
    r = SplitMix64(seed)
    s = Scene()
    for i in range(n):
        cx, cy, rad = r.uniform(0, size), r.uniform(0, size), r.uniform(rmin, rmax)
        col = (r.uniform(), r.uniform(), r.uniform(), 0.5)
        if i % 2 == 0:
            s.fill(Fill.NonZero, None, Brush.solid(col), None, Path.circle(cx, cy, rad))
        else:
            s.stroke(Stroke(width=3.0), None, Brush.solid(col), None, Path.circle(cx, cy, rad))
    return s, RenderParams(size, size, base_color=(0, 0, 0, 1))

def scene_long_lines(n, size, seed=78):
    """n thin strokes from edge to edge (a chart / grid): every line crosses hundreds of tiles."""
    r = SplitMix64(seed)
    s = Scene()
    for i in range(n):
        p = Path().move_to(0.0, r.uniform(0, size)).line_to(float(size), r.uniform(0, size)) if i % 2 == 0 else \
            Path().move_to(r.uniform(0, size), 0.0).line_to(r.uniform(0, size), float(size))
        s.stroke(Stroke(width=1.5), None, Brush.solid((r.uniform(), r.uniform(), r.uniform(), 0.8)), None, p)
    return s, RenderParams(size, size, base_color=(1, 1, 1, 1))

def scene_tiny_rects(n, size, seed=79):
    """n rectangles of 2..6 px (particles / a scatter plot): draw-object-bound stages."""
    r = SplitMix64(seed)
    s = Scene()
    for i in range(n):
        x, y, w, h = r.uniform(0, size - 8), r.uniform(0, size - 8), r.uniform(2, 6), r.uniform(2, 6)
        s.fill(Fill.NonZero, None, Brush.solid((r.uniform(), r.uniform(), r.uniform(), 0.9)), None, Path.rect(x, y, x + w, y + h))
    return s, RenderParams(size, size, base_color=(0, 0, 0, 1))

def scene_polygon(n_pts, size, seed=80):
    """One polygon with n_pts vertices (a coastline): a single path with a very long tag stream."""
    r = SplitMix64(seed)
    s = Scene()
    p = Path()
    c = size * 0.5
    for i in range(n_pts):
        a = 2.0 * math.pi * i / n_pts
        rad = size * (0.25 + 0.2 * r.uniform())
        x, y = c + rad * math.cos(a), c + rad * math.sin(a)
        p = p.move_to(x, y) if i == 0 else p.line_to(x, y)
    p = p.close()
    s.fill(Fill.EvenOdd, None, Brush.solid((0.2, 0.5, 0.3, 1.0)), None, p)
    return s, RenderParams(size, size, base_color=(0, 0, 0, 1))


def scene_gradients(n, size, kind, seed=81):
    """C3-like filled cubics whose brushes are all gradients of one kind ("linear", "radial", "sweep") or a 64x64 image."""
    import numpy as np
    from jello_amd.scene import ColorStop, Extend
    r = SplitMix64(seed)
    s = Scene()
    img = (np.arange(64 * 64 * 4, dtype=np.uint32) * 2654435761 >> 24).astype(np.uint8).reshape(64, 64, 4)
    for i in range(n):
        ax, ay = r.uniform(0, size), r.uniform(0, size)
        p = Path().move_to(ax, ay)
        p.cubic_to(ax + r.uniform(-32, 32), ay + r.uniform(-32, 32), ax + r.uniform(-32, 32), ay + r.uniform(-32, 32),
                   ax + r.uniform(-32, 32), ay + r.uniform(-32, 32))
        stops = [ColorStop(0.0, (r.uniform(), r.uniform(), r.uniform(), 1.0)), ColorStop(1.0, (r.uniform(), r.uniform(), r.uniform(), 0.5))]
        bt = None
        if kind == "linear":
            b = Brush.linear((ax - 20, ay), (ax + 20, ay + 10), stops, Extend.Pad)
        elif kind == "radial":
            b = Brush.radial((ax, ay), 2.0, (ax + 5, ay + 3), 40.0, stops, Extend.Pad)
        elif kind == "sweep":
            b = Brush.sweep((ax, ay), 0.0, 1.0, stops, Extend.Pad)
        else:
            b = Brush.image(img, key=7)
            bt = (1, 0, 0, 1, ax - 32, ay - 32)
        s.fill(Fill.NonZero, None, b, bt, p)
    return s, RenderParams(size, size, base_color=(0, 0, 0, 1))


def scene_glyphs(n, w, h, seed=82):
    """n glyph-like outlines: closed paths of 6..14 quadratic segments inside 8..28 px boxes (two contours for every third
    one, filled even-odd), laid out on text lines."""
    r = SplitMix64(seed)
    s = Scene()
    x, y = 8.0, 30.0
    for i in range(n):
        size = r.uniform(8, 28)
        p = Path()
        for contour in range(2 if i % 3 == 0 else 1):
            k = 6 + int(r.uniform(0, 9))
            rad = size * (0.5 if contour == 0 else 0.22)
            cx, cy = x + size * 0.5, y - size * 0.5
            pts = [(cx + rad * math.cos(2 * math.pi * j / k) * r.uniform(0.6, 1.0), cy + rad * math.sin(2 * math.pi * j / k) * r.uniform(0.6, 1.0))
                   for j in range(k)]
            p.move_to(*pts[0])
            for j in range(k):
                a, b = pts[j], pts[(j + 1) % k]
                p.quad_to((a[0] + b[0]) * 0.5 + r.uniform(-2, 2), (a[1] + b[1]) * 0.5 + r.uniform(-2, 2), b[0], b[1])
            p.close()
        s.fill(Fill.EvenOdd, None, Brush.solid((0.05, 0.05, 0.1, 1.0)), None, p)
        x += size * 0.9
        if x > w - 40:
            x = 8.0
            y += 34.0
            if y > h - 8:
                y = 30.0
    return s, RenderParams(w, h, base_color=(1, 1, 1, 1))


def scene_roads(n, size, seed=84):
    """n stroked random-walk polylines of 40..200 vertices, steps of 5..25 px (a road network): mid-size paths with hundreds
    of tile crossings each, all of them on the list route of path_count."""
    r = SplitMix64(seed)
    s = Scene()
    for i in range(n):
        x, y = r.uniform(0, size), r.uniform(0, size)
        a = r.uniform(0, 2 * math.pi)
        p = Path().move_to(x, y)
        for _ in range(40 + int(r.uniform(0, 160))):
            a += r.uniform(-0.5, 0.5)
            st = r.uniform(5, 25)
            x, y = x + st * math.cos(a), y + st * math.sin(a)
            p.line_to(x, y)
        s.stroke(Stroke(width=r.uniform(1.0, 4.0)), None, Brush.solid((r.uniform(), r.uniform(), r.uniform(), 1.0)), None, p)
    return s, RenderParams(size, size, base_color=(0.95, 0.95, 0.9, 1))


def big_buffers():
    return BumpSizes(lines=1 << 23, seg_counts=1 << 24, segments=1 << 24, tiles=1 << 24, ptcl=1 << 27, bin_data=1 << 22, blend_spill=1 << 20)


def _hd(mk):
    s, p = mk()
    p.height = 1088
    return s, p


CASES = [("C1", scenes.scene_c1),
         ("2000 circles r 20..150, 1920x1088", lambda: _hd(lambda: scene_shapes(2000, 20, 150, 1920))),
         ("300 circles r 200..1000, 4096^2", lambda: scene_shapes(300, 200, 1000, 4096)),
         ("20 circles r 1000..2000, 4096^2", lambda: scene_shapes(20, 1000, 2000, 4096)),
         ("C3 20k paths, 4096^2", lambda: scenes.scene_c3(20000, 4096)),
         ("1000 edge-to-edge strokes, 4096^2", lambda: scene_long_lines(1000, 4096)),
         ("200k rects of 2..6 px, 1920x1088", lambda: _hd(lambda: scene_tiny_rects(200000, 1920))),
         ("one polygon, 200k vertices, 2048^2", lambda: scene_polygon(200000, 2048)),
         ("40k solid fills (C3 without strokes), 4096^2", lambda: scenes.scene_c3(40000, 4096)),
         ("40k linear-gradient fills, 4096^2", lambda: scene_gradients(40000, 4096, "linear")),
         ("40k radial-gradient fills, 4096^2", lambda: scene_gradients(40000, 4096, "radial")),
         ("40k sweep-gradient fills, 4096^2", lambda: scene_gradients(40000, 4096, "sweep")),
         ("40k image fills, 4096^2", lambda: scene_gradients(40000, 4096, "image")),
         ("50k glyph outlines, 3840x2160", lambda: scene_glyphs(50000, 3840, 2160)),
         ("C3-like 30k paths, 16384x1024", lambda: _wide()),
         ("10k road polylines (40..200 vertices), 4096^2", lambda: scene_roads(10000, 4096))]


def _wide():
    r = SplitMix64(83)
    s = Scene()
    for i in range(30000):
        ax, ay = r.uniform(0, 16384), r.uniform(0, 1024)
        p = Path().move_to(ax, ay)
        p.cubic_to(ax + r.uniform(-32, 32), ay + r.uniform(-32, 32), ax + r.uniform(-32, 32), ay + r.uniform(-32, 32),
                   ax + r.uniform(-32, 32), ay + r.uniform(-32, 32))
        s.fill(Fill.NonZero, None, Brush.solid((r.uniform(), r.uniform(), r.uniform(), 0.8)), None, p)
    return s, RenderParams(16384, 1024, base_color=(0, 0, 0, 1))


def select(keys):
    return [c for c in CASES if not keys or any(k in c[0] for k in keys)]
