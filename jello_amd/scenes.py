"""Seeded synthetic scenes for the BASELINE.json configs.

C1  rect fill + stroked cubic, 512x512 (plumbing)
C2  tiger substitute: ~300 seeded blobs, fills + strokes, 1024x1024 (the Ghostscript tiger's path
    data is not in the reference tree; this is flagged as a substitute)
C3  n random stroked+filled cubic Beziers (headline: n = 100k at 4096x4096)
C4  nested clips + radial gradients + blends (30k paths at 2048x2048)
"""
import ctypes
import math

import numpy as np

from .scene import (Brush, Cap, Color, ColorStop, Compose, Extend, Fill, Join, Mix, Path, RenderParams, Scene, Stroke)

SEED = 0x6A656C6C6F  # "jello"
_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed=SEED):
        self.s = seed & _M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * ((self.next() >> 11) * (1.0 / (1 << 53)))


def splitmix64_array(n, seed=SEED):
    """Vectorised SplitMix64: the same stream as SplitMix64(seed).next() called n times -> uniform [0,1)."""
    i = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def scene_c1():
    s = Scene()
    s.fill(Fill.NonZero, None, Brush.solid((1, 0, 0, 1)), None, Path.rect(10, 10, 200, 150))
    p = Path().move_to(50, 300).cubic_to(150, 100, 350, 500, 450, 300)
    s.stroke(Stroke(8, Join.Miter, 4, Cap.Butt, Cap.Butt), None, Brush.solid((0, 0, 1, 1)), None, p)
    return s, RenderParams(512, 512)


def scene_c3(n_paths=100_000, size=4096, seed=SEED, spread=32.0, stroked=True):
    """n random cubic Beziers, each filled (non-zero) and then stroked (round join, butt caps)."""
    # 17 uniforms per path, drawn in this order: anchor xy, 3 x (dx, dy), fill rgba, width, stroke rgba
    u = splitmix64_array(n_paths * 17, seed).reshape(n_paths, 17)
    anchor = u[:, 0:2] * size
    pts = np.empty((n_paths, 8), dtype=np.float64)
    pts[:, 0:2] = anchor
    for k in range(3):
        pts[:, 2 + 2 * k:4 + 2 * k] = anchor + (u[:, 2 + 2 * k:4 + 2 * k] * 2.0 - 1.0) * spread
    fill = np.ascontiguousarray(u[:, 8:12])
    widths = np.ascontiguousarray(0.5 + u[:, 12] * 3.5) if stroked else None
    stroke = np.ascontiguousarray(u[:, 13:17])
    s = Scene()
    dp = ctypes.POINTER(ctypes.c_double)
    pts = np.ascontiguousarray(pts)
    rc = s._L.jl_scene_fill_stroke_cubics(s._h, n_paths, pts.ctypes.data_as(dp), fill.ctypes.data_as(dp), stroke.ctypes.data_as(dp),
                                          widths.ctypes.data_as(dp) if stroked else None, int(Join.Round), int(Cap.Butt), int(Cap.Butt))
    if rc != 0:
        raise RuntimeError(s._L.jl_last_error().decode())
    return s, RenderParams(size, size)


def scene_c2(n_blobs=300, size=1024, seed=SEED + 2):
    """Tiger substitute: blobs of 3-6 cubic segments, filled (some even-odd) and some stroked with miter/round joins and caps."""
    r = SplitMix64(seed)
    s = Scene()
    for i in range(n_blobs):
        cx, cy = r.uniform(0, size), r.uniform(0, size)
        rad = r.uniform(8, 120)
        nseg = 3 + int(r.uniform(0, 4))
        p = Path()
        ang0 = r.uniform(0, 2 * math.pi)
        pts = []
        for k in range(nseg):
            a = ang0 + 2 * math.pi * k / nseg
            rr = rad * r.uniform(0.5, 1.2)
            pts.append((cx + rr * math.cos(a), cy + rr * math.sin(a)))
        p.move_to(*pts[0])
        for k in range(nseg):
            a, b = pts[k], pts[(k + 1) % nseg]
            c1 = (a[0] + r.uniform(-rad, rad) * 0.5, a[1] + r.uniform(-rad, rad) * 0.5)
            c2 = (b[0] + r.uniform(-rad, rad) * 0.5, b[1] + r.uniform(-rad, rad) * 0.5)
            p.cubic_to(c1[0], c1[1], c2[0], c2[1], b[0], b[1])
        p.close()
        col = (r.uniform(), r.uniform(), r.uniform(), r.uniform(0.3, 1.0))
        rule = Fill.EvenOdd if i % 5 == 0 else Fill.NonZero
        s.fill(rule, None, Brush.solid(col), None, p)
        if i % 3 == 0:
            joins = [Join.Bevel, Join.Miter, Join.Round]
            caps = [Cap.Butt, Cap.Square, Cap.Round]
            st = Stroke(r.uniform(0.5, 6), joins[i % 3 if i % 9 else (i // 9) % 3], 4.0, caps[(i // 3) % 3], caps[(i // 6) % 3])
            q = Path()
            q.move_to(*pts[0])
            for k in range(1, nseg):
                q.line_to(*pts[k]) if k % 2 else q.quad_to(cx, cy, *pts[k])
            s.stroke(st, None, Brush.solid((r.uniform(), r.uniform(), r.uniform(), 1.0)), None, q)
    return s, RenderParams(size, size, base_color=(1, 1, 1, 1))


def scene_c4(n_paths=30_000, size=2048, seed=SEED + 4, group=10, depth=3):
    """Groups of `group` paths under PushLayer(blend (Mix i%16, SrcOver), alpha .8, clip = random circle) nested `depth` deep;
    every 3rd brush a 3-stop radial gradient."""
    r = SplitMix64(seed)
    s = Scene()
    i = 0
    gi = 0
    while i < n_paths:
        layers = 0
        for d in range(depth):
            cx, cy = r.uniform(0, size), r.uniform(0, size)
            mix = Mix(gi % 16) if d == 0 else Mix.Clip
            s.push_layer(mix, Compose.SrcOver, 0.8 if d == 0 else 1.0, None, Path.circle(cx, cy, 128.0 + 64.0 * (depth - d)))
            layers += 1
            gi += 1
        for _ in range(group):
            if i >= n_paths:
                break
            ax, ay = r.uniform(0, size), r.uniform(0, size)
            p = Path().move_to(ax, ay)
            p.cubic_to(ax + r.uniform(-96, 96), ay + r.uniform(-96, 96), ax + r.uniform(-96, 96), ay + r.uniform(-96, 96),
                       ax + r.uniform(-96, 96), ay + r.uniform(-96, 96))
            if i % 3 == 2:
                stops = [ColorStop(0.0, (r.uniform(), r.uniform(), r.uniform(), 1.0)), ColorStop(0.5, (r.uniform(), r.uniform(), r.uniform(), 0.8)),
                         ColorStop(1.0, (r.uniform(), r.uniform(), r.uniform(), 0.6))]
                b = Brush.radial((ax, ay), 4.0, (ax + 10, ay + 5), 80.0, stops, Extend.Pad)
            else:
                b = Brush.solid((r.uniform(), r.uniform(), r.uniform(), r.uniform(0.2, 1.0)))
            s.fill(Fill.NonZero, None, b, None, p)
            i += 1
        for _ in range(layers):
            s.pop_layer()
    return s, RenderParams(size, size, base_color=(0, 0, 0, 1))


def scene_c4_nested(n_paths=30_000, size=2048, seed=SEED + 14, group=10, depth=3):
    """C4 with clips that nest the way an SVG's do: each group's `depth` clip circles are (nearly) concentric, every one a
    little smaller than its parent and nudged off centre, and the group's paths start inside the innermost.  scene_c4 draws
    its three circles at independent random positions: their intersection is empty almost everywhere, so of its 30 k paths
    92 fills reach the PTCL and the frame is 212 empty blend layers per tile -- a stress of the clip stack, not of
    compositing.  Here the paints are visible: gradients and blends are composited at scale."""
    r = SplitMix64(seed)
    s = Scene()
    i = 0
    gi = 0
    while i < n_paths:
        cx, cy = r.uniform(0, size), r.uniform(0, size)
        layers = 0
        for d in range(depth):
            mix = Mix(gi % 16) if d == 0 else Mix.Clip
            rad = 128.0 + 64.0 * (depth - d)
            s.push_layer(mix, Compose.SrcOver, 0.8 if d == 0 else 1.0, None,
                         Path.circle(cx + r.uniform(-24, 24) * d, cy + r.uniform(-24, 24) * d, rad))
            layers += 1
            gi += 1
        for _ in range(group):
            if i >= n_paths:
                break
            ax, ay = cx + r.uniform(-160, 160), cy + r.uniform(-160, 160)
            p = Path().move_to(ax, ay)
            p.cubic_to(ax + r.uniform(-96, 96), ay + r.uniform(-96, 96), ax + r.uniform(-96, 96), ay + r.uniform(-96, 96),
                       ax + r.uniform(-96, 96), ay + r.uniform(-96, 96))
            if i % 3 == 2:
                stops = [ColorStop(0.0, (r.uniform(), r.uniform(), r.uniform(), 1.0)), ColorStop(0.5, (r.uniform(), r.uniform(), r.uniform(), 0.8)),
                         ColorStop(1.0, (r.uniform(), r.uniform(), r.uniform(), 0.6))]
                b = Brush.radial((ax, ay), 4.0, (ax + 10, ay + 5), 80.0, stops, Extend.Pad)
            else:
                b = Brush.solid((r.uniform(), r.uniform(), r.uniform(), r.uniform(0.2, 1.0)))
            s.fill(Fill.NonZero, None, b, None, p)
            i += 1
        for _ in range(layers):
            s.pop_layer()
    return s, RenderParams(size, size, base_color=(0, 0, 0, 1))


def scene_images(size=256, seed=SEED + 7):
    """Two RGBA8 (sRGB-encoded) images as brushes: one axis-aligned at 1:1, one rotated/scaled through the
    brush transform so that the bilinear taps and the extent test of fine.wgsl:1068-1087 are exercised, plus
    a solid shape on top.  The reference uploads scene images as Rgba8Srgb (render.go:137)."""
    u = splitmix64_array(64 * 48 * 4 + 32 * 32 * 4, seed)
    img_a = (u[:64 * 48 * 4] * 256.0).astype(np.uint8).reshape(48, 64, 4)
    img_b = (u[64 * 48 * 4:] * 256.0).astype(np.uint8).reshape(32, 32, 4)
    img_b[:, :, 3] = 255  # opaque
    s = Scene()
    s.fill(Fill.NonZero, None, Brush.image(img_a, key=1), (1, 0, 0, 1, 20, 30), Path.rect(20, 30, 20 + 64, 30 + 48))
    c, sn = math.cos(0.5), math.sin(0.5)
    xf = (3.0 * c, 3.0 * sn, -3.0 * sn, 3.0 * c, 120.0, 90.0)
    s.fill(Fill.NonZero, None, Brush.image(img_b, key=2), xf, Path.circle(150, 150, 70))
    s.fill(Fill.EvenOdd, None, Brush.solid((0.1, 0.7, 0.2, 0.5)), None, Path.rect(60, 60, 200, 120))
    return s, RenderParams(size, size, base_color=(0.2, 0.2, 0.2, 1.0))


def scene_many_images(n=13, seed=SEED + 13):
    """n small opaque sRGB images, each filling its own 32x32 square (5 per row): more than the 8 image descriptors
    that fit in fine's kernel arguments, so the device descriptor table is used."""
    s = Scene()
    u = splitmix64_array(n * 8 * 8 * 4, seed)
    for k in range(n):
        px = (u[k * 256:(k + 1) * 256] * 256.0).astype(np.uint8).reshape(8, 8, 4)
        px[:, :, 3] = 255
        x, y = 36 * (k % 5), 36 * (k // 5)
        s.fill(Fill.NonZero, None, Brush.image(px, key=100 + k), (4, 0, 0, 4, x, y), Path.rect(x, y, x + 32, y + 32))
    rows = (n + 4) // 5
    return s, RenderParams(192, max(16, 36 * rows), base_color=(0, 0, 0, 0))


def scene_large_shapes(size=1536, n=60, seed=SEED + 11):
    """Shapes of hundreds of tiles: a background rectangle over the whole target, filled and stroked circles of radius
    100..700 (partly outside the target), under one clip layer.  Exercises what small random curves never reach: a
    draw object in every bin, tile rows far wider than a wave handles in one step, paths with hundreds to thousands
    of tile crossings, rows clipped by the target edge."""
    r = SplitMix64(seed)
    s = Scene()
    s.fill(Fill.NonZero, None, Brush.solid((0.9, 0.9, 0.85, 1.0)), None, Path.rect(0, 0, size, size))
    for i in range(n):
        if i == n // 2:
            s.push_layer(Mix.Multiply, Compose.SrcOver, 0.9, None, Path.circle(size * 0.5, size * 0.5, size * 0.45))
        cx, cy, rad = r.uniform(-100, size + 100), r.uniform(-100, size + 100), r.uniform(100, 700)
        col = (r.uniform(), r.uniform(), r.uniform(), r.uniform(0.3, 0.9))
        if i % 3 == 0:
            s.stroke(Stroke(width=r.uniform(1.0, 20.0)), None, Brush.solid(col), None, Path.circle(cx, cy, rad))
        elif i % 3 == 1:
            s.fill(Fill.NonZero, None, Brush.solid(col), None, Path.circle(cx, cy, rad))
        else:
            s.fill(Fill.EvenOdd, None, Brush.solid(col), None, Path.rect(cx - rad, cy - rad * 0.3, cx + rad, cy + rad * 0.3))
    s.pop_layer()
    return s, RenderParams(size, size, base_color=(0, 0, 0, 1))


def scene_dense_polygon(n_pts=40000, size=256, seed=SEED + 12):
    """One even-odd polygon whose n_pts vertices zigzag radially (a dense outline zoomed far out): thousands of tile
    crossings of ONE path in every tile of a ring -- per-tile lists far longer than a wave, under a few long thin
    strokes from edge to edge (every line crosses dozens of tiles)."""
    r = SplitMix64(seed)
    s = Scene()
    p = Path()
    c = size * 0.5
    for i in range(n_pts):
        a = 2.0 * math.pi * i / n_pts
        rad = size * (0.25 + 0.2 * r.uniform())
        x, y = c + rad * math.cos(a), c + rad * math.sin(a)
        p = p.move_to(x, y) if i == 0 else p.line_to(x, y)
    p.close()
    s.fill(Fill.EvenOdd, None, Brush.solid((0.2, 0.5, 0.3, 1.0)), None, p)
    for i in range(24):
        q = Path().move_to(0.0, r.uniform(0, size)).line_to(float(size), r.uniform(0, size)) if i % 2 == 0 else \
            Path().move_to(r.uniform(0, size), 0.0).line_to(r.uniform(0, size), float(size))
        s.stroke(Stroke(width=r.uniform(0.5, 3.0)), None, Brush.solid((r.uniform(), r.uniform(), r.uniform(), 0.8)), None, q)
    return s, RenderParams(size, size, base_color=(1, 1, 1, 1))


def scene_big_path(size=1024, n_zig=600, seed=SEED + 9):
    """One path with far more tile crossings than PC_BIG_PATH (a long zigzag polyline, filled even-odd and stroked)
    on top of a few small shapes: path_count's list-based route for big paths next to the atomics-free one."""
    u = splitmix64_array(40 * 6, seed).reshape(40, 6)
    s = Scene()
    for i in range(40):
        cx, cy, r = u[i, 0] * size, u[i, 1] * size, 10 + u[i, 2] * 50
        s.fill(Fill.NonZero, None, Brush.solid((u[i, 3], u[i, 4], u[i, 5], 0.8)), None, Path.circle(cx, cy, r))
    p = Path().move_to(8.5, 20.25)
    for k in range(n_zig):
        x = size - 9.25 if (k % 2 == 0) else 8.5
        y = 20.25 + (size - 40.0) * (k + 1) / n_zig
        p.line_to(x, y)
    p.close()
    s.fill(Fill.EvenOdd, None, Brush.solid((0.9, 0.2, 0.1, 0.6)), None, p)
    s.stroke(Stroke(1.5, Join.Bevel, 4, Cap.Butt, Cap.Butt), None, Brush.solid((0.0, 0.0, 0.0, 1.0)), None, p)
    return s, RenderParams(size, size, base_color=(1, 1, 1, 1))


def scene_fuzz(seed, size=256, n=40, extreme=False):
    """Random mixture of everything the pipeline handles: fills (both rules) and strokes (all joins / caps / widths,
    closed and open, lines / quads / cubics, degenerate segments), per-draw affine transforms, solid / linear / radial /
    sweep / image brushes with all extend modes, nested clip layers with every mix mode.  Drives the parity fuzz test.
    extreme: coordinates far outside the target, scales from 0.02 to 40, stroke widths up to 300, up to 9 nested
    layers (more than the 4 blend-stack entries fine keeps in registers: the spill buffer), twice the draws."""
    r = SplitMix64(SEED + (5000 if extreme else 1000) + seed)
    span_lo, span_hi = (-2.0 * size, 3.0 * size) if extreme else (0.0, float(size))
    max_layers = 9 if extreme else 5
    if extreme:
        n *= 2
    s = Scene()
    img = (splitmix64_array(16 * 16 * 4, SEED + 2000 + seed) * 256.0).astype(np.uint8).reshape(16, 16, 4)
    open_layers = 0

    def rnd_affine():
        k = int(r.uniform(0, 4))
        if k == 0:
            return None
        a, sc = r.uniform(0, 2 * math.pi), r.uniform(0.4, 2.0)
        if extreme and r.uniform() < 0.3:
            sc = 10.0 ** r.uniform(-1.7, 1.6)
        c, sn = math.cos(a) * sc, math.sin(a) * sc
        return (c, sn, -sn, c * r.uniform(0.5, 1.5), r.uniform(0, size * 0.5), r.uniform(0, size * 0.5))

    def rnd_path(closed):
        p = Path()
        x, y = r.uniform(span_lo, span_hi), r.uniform(span_lo, span_hi)
        if extreme and r.uniform() < 0.6:
            x, y = r.uniform(0, size), r.uniform(0, size)
        p.move_to(x, y)
        for _ in range(1 + int(r.uniform(0, 5))):
            k = int(r.uniform(0, 4))
            ext = r.uniform(2, size * 0.4) if not extreme else 10.0 ** r.uniform(-2.0, math.log10(size * 2.0))
            nx, ny = x + r.uniform(-ext, ext), y + r.uniform(-ext, ext)
            if k == 0:
                p.line_to(nx, ny)
            elif k == 1:
                p.quad_to(x + r.uniform(-ext, ext), y + r.uniform(-ext, ext), nx, ny)
            elif k == 2:
                p.cubic_to(x + r.uniform(-ext, ext), y + r.uniform(-ext, ext), nx + r.uniform(-ext, ext), ny + r.uniform(-ext, ext), nx, ny)
            else:
                nx, ny = x, y          # zero-length segment
                p.line_to(nx, ny)
            x, y = nx, ny
        if closed:
            p.close()
        return p

    def rnd_brush():
        k = int(r.uniform(0, 6))
        col = lambda: (r.uniform(), r.uniform(), r.uniform(), r.uniform(0.2, 1.0))
        stops = [ColorStop(0.0, col()), ColorStop(r.uniform(0.2, 0.8), col()), ColorStop(1.0, col())]
        ext = Extend(int(r.uniform(0, 3)))
        if k <= 1:
            return Brush.solid(col())
        if k == 2:
            return Brush.linear((r.uniform(0, size), r.uniform(0, size)), (r.uniform(0, size), r.uniform(0, size)), stops, ext)
        if k == 3:
            c0 = (r.uniform(0, size), r.uniform(0, size))
            return Brush.radial(c0, r.uniform(0, 20), (c0[0] + r.uniform(-30, 30), c0[1] + r.uniform(-30, 30)), r.uniform(25, 120), stops, ext)
        if k == 4:
            return Brush.sweep((r.uniform(0, size), r.uniform(0, size)), 0.0, r.uniform(0.3, 1.0), stops, ext)
        return Brush.image(img, key=3000 + seed)

    for i in range(n):
        act = int(r.uniform(0, 10))
        if (act == 0 or (extreme and act == 2)) and open_layers < max_layers:
            s.push_layer(Mix(int(r.uniform(0, 16))) if r.uniform() < 0.7 else Mix.Clip, Compose.SrcOver, r.uniform(0.3, 1.0), rnd_affine(),
                         Path.circle(r.uniform(0.2 * size, 0.8 * size), r.uniform(0.2 * size, 0.8 * size), r.uniform(0.3 * size, 0.8 * size)))
            open_layers += 1
        elif act == 1 and open_layers > 0:
            s.pop_layer()
            open_layers -= 1
        elif act <= 5:
            s.fill(Fill.EvenOdd if r.uniform() < 0.3 else Fill.NonZero, rnd_affine(), rnd_brush(), rnd_affine() if r.uniform() < 0.3 else None,
                   rnd_path(True))
        else:
            wd = r.uniform(0.3, 12.0) if not extreme else 10.0 ** r.uniform(-1.5, 2.5)
            st = Stroke(wd, Join(int(r.uniform(0, 3))), r.uniform(1.0, 8.0), Cap(int(r.uniform(0, 3))), Cap(int(r.uniform(0, 3))))
            s.stroke(st, rnd_affine(), rnd_brush(), None, rnd_path(r.uniform() < 0.4))
    while open_layers > 0:
        s.pop_layer()
        open_layers -= 1
    p = RenderParams(size, size, base_color=(r.uniform(), r.uniform(), r.uniform(), 1.0))
    return s, p


def scene_clip_torture(kind, size=256, seed=SEED + 31):
    """Clip-layer patterns that stress coarse's per-tile state machine (kernels_coarse.hip, the walk with lanes = elements) and
    fine's lazy layers far beyond what the C4 recipes produce -- small targets, so that the oracle finishes in seconds:
      deep       one nest of 120 layers around the centre, every fifth one with a mix mode, content at several depths
      siblings   1500 sibling layers over the same tiles, alternately covering them, missing them (empty: everything inside is
                 skipped) and cutting through them, each with two fills inside -- hundreds of elements per tile and batch
      comb       layers that open in one batch of 256 draw objects and close several batches later: 40 long-lived layers
                 interleaved with 1200 fills, some of the layers empty on half of the target
      mixed      random interleavings of pushes, pops and fills, up to 60 layers open at once"""
    r = SplitMix64(seed + {"deep": 0, "siblings": 1, "comb": 2, "mixed": 3}[kind])
    s = Scene()
    col = lambda: Brush.solid((r.uniform(), r.uniform(), r.uniform(), r.uniform(0.3, 1.0)))

    def blob(cx, cy, rad):
        p = Path().move_to(cx + r.uniform(-rad, rad), cy + r.uniform(-rad, rad))
        for _ in range(3):
            p.line_to(cx + r.uniform(-rad, rad), cy + r.uniform(-rad, rad))
        return p.close()
    c = size * 0.5
    if kind == "deep":
        depth = 120
        for d in range(depth):
            rad = size * 0.48 - d * (size * 0.4 / depth)
            mix = Mix((d // 5) % 16) if d % 5 == 0 else Mix.Clip
            s.push_layer(mix, Compose.SrcOver, 0.9 if d % 5 == 0 else 1.0, None, Path.circle(c + r.uniform(-2, 2), c + r.uniform(-2, 2), rad))
            if d % 7 == 0:
                s.fill(Fill.NonZero, None, col(), None, blob(c, c, size * 0.3))
        s.fill(Fill.EvenOdd, None, col(), None, blob(c, c, size * 0.2))
        for _ in range(depth):
            s.pop_layer()
    elif kind == "siblings":
        for i in range(1500):
            k = i % 3
            if k == 0:
                clip = Path.rect(-10, -10, size + 10, size + 10)                      # covers every tile
            elif k == 1:
                clip = Path.circle(r.uniform(0, size), r.uniform(0, size), 6.0)       # misses almost every tile
            else:
                clip = blob(r.uniform(0, size), r.uniform(0, size), size * 0.4)       # cuts through
            s.push_layer(Mix(i % 16) if i % 4 == 0 else Mix.Clip, Compose.SrcOver, 0.8, None, clip)
            s.fill(Fill.NonZero, None, col(), None, blob(r.uniform(0, size), r.uniform(0, size), 30.0))
            s.fill(Fill.NonZero, None, col(), None, blob(r.uniform(0, size), r.uniform(0, size), 12.0))
            s.pop_layer()
    elif kind == "comb":
        open_ = 0
        for i in range(1200):
            if i % 30 == 0 and open_ < 40:
                half = Path.rect(0, 0, size * (0.5 if (i // 30) % 2 else 1.0), size)
                s.push_layer(Mix((i // 30) % 16), Compose.SrcOver, 0.7, None, half)
                open_ += 1
            s.fill(Fill.NonZero, None, col(), None, blob(r.uniform(0, size), r.uniform(0, size), 25.0))
            if i % 97 == 96 and open_ > 0:
                s.pop_layer()
                open_ -= 1
        for _ in range(open_):
            s.pop_layer()
    else:
        open_ = 0
        for i in range(2500):
            u = r.uniform()
            if u < 0.25 and open_ < 60:
                rad = 10.0 ** r.uniform(0.3, math.log10(size * 0.7))
                s.push_layer(Mix(int(r.uniform(0, 16))) if r.uniform() < 0.3 else Mix.Clip, Compose.SrcOver, r.uniform(0.5, 1.0), None,
                             Path.circle(r.uniform(0, size), r.uniform(0, size), rad))
                open_ += 1
            elif u < 0.5:
                if open_ > 0:
                    s.pop_layer()
                    open_ -= 1
            else:
                s.fill(Fill.NonZero if r.uniform() < 0.8 else Fill.EvenOdd, None, col(), None, blob(r.uniform(0, size), r.uniform(0, size), 10.0 ** r.uniform(0.5, 2.0)))
        for _ in range(open_):
            s.pop_layer()
    return s, RenderParams(size, size, base_color=(0.1, 0.1, 0.1, 1.0))
