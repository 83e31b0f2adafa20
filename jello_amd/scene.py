"""Python face of the host-side Scene API (C++ in jello_amd/host/scene.{h,cpp}, which mirrors
scene.go of the reference).  Method names and argument meaning follow the reference:
Scene.fill / stroke / push_layer / pop_layer / append / apply_transform."""
import ctypes
import enum
import itertools

import numpy as np

from . import _lib
from ._lib import CBrush, CBumpSizes, CColorStop, CRenderParams, CStroke, PathEl

IDENTITY = (1.0, 0.0, 0.0, 1.0, 0.0, 0.0)


class Fill(enum.IntEnum):
    NonZero = 0
    EvenOdd = 1


class Join(enum.IntEnum):
    Bevel = 0
    Miter = 1
    Round = 2


class Cap(enum.IntEnum):
    Butt = 0
    Square = 1
    Round = 2


class Extend(enum.IntEnum):
    Pad = 0
    Repeat = 1
    Reflect = 2


class Aa(enum.IntEnum):
    Area = 0
    Msaa8 = 1
    Msaa16 = 2


class Mix(enum.IntEnum):
    Normal = 0; Multiply = 1; Screen = 2; Overlay = 3; Darken = 4; Lighten = 5; ColorDodge = 6; ColorBurn = 7
    HardLight = 8; SoftLight = 9; Difference = 10; Exclusion = 11; Hue = 12; Saturation = 13; Color = 14
    Luminosity = 15; Clip = 128


class Compose(enum.IntEnum):
    SrcOver = 0; Copy = 1; Dest = 2; Clear = 3; DestOver = 4; SrcIn = 5; DestIn = 6; SrcOut = 7; DestOut = 8
    SrcAtop = 9; DestAtop = 10; Xor = 11; Plus = 12; PlusLighter = 13


class Color(tuple):
    """Linear-sRGB, un-premultiplied (r, g, b, a)."""
    def __new__(cls, r, g, b, a=1.0):
        return super().__new__(cls, (float(r), float(g), float(b), float(a)))


class ColorStop:
    def __init__(self, offset, color):
        self.offset, self.color = float(offset), Color(*color)


class Path:
    """curve.BezPath: a list of elements built with move_to / line_to / quad_to / cubic_to / close."""
    MOVE, LINE, QUAD, CUBIC, CLOSE = 0, 1, 2, 3, 4

    def __init__(self):
        self.els = []

    def move_to(self, x, y): self.els.append((0, (x, y, 0, 0, 0, 0))); return self
    def line_to(self, x, y): self.els.append((1, (x, y, 0, 0, 0, 0))); return self
    def quad_to(self, x1, y1, x2, y2): self.els.append((2, (x1, y1, x2, y2, 0, 0))); return self
    def cubic_to(self, x1, y1, x2, y2, x3, y3): self.els.append((3, (x1, y1, x2, y2, x3, y3))); return self
    def close(self): self.els.append((4, (0, 0, 0, 0, 0, 0))); return self

    @staticmethod
    def rect(x0, y0, x1, y1):
        return Path().move_to(x0, y0).line_to(x1, y0).line_to(x1, y1).line_to(x0, y1).close()

    @staticmethod
    def circle(cx, cy, r):
        k = 0.5522847498307936 * r
        p = Path().move_to(cx + r, cy)
        p.cubic_to(cx + r, cy + k, cx + k, cy + r, cx, cy + r)
        p.cubic_to(cx - k, cy + r, cx - r, cy + k, cx - r, cy)
        p.cubic_to(cx - r, cy - k, cx - k, cy - r, cx, cy - r)
        p.cubic_to(cx + k, cy - r, cx + r, cy - k, cx + r, cy)
        return p.close()

    def _c(self):
        arr = (PathEl * len(self.els))()
        for i, (k, pts) in enumerate(self.els):
            arr[i].kind = k
            for j in range(6):
                arr[i].pts[j] = pts[j]
        return arr


class Brush:
    SOLID, LINEAR, RADIAL, SWEEP, IMAGE = 0, 1, 2, 3, 4

    def __init__(self, kind, **kw):
        self.kind = kind
        self.kw = kw

    @staticmethod
    def solid(color): return Brush(Brush.SOLID, color=Color(*color))
    @staticmethod
    def linear(p0, p1, stops, extend=Extend.Pad): return Brush(Brush.LINEAR, p0=p0, p1=p1, stops=stops, extend=extend)
    @staticmethod
    def radial(c0, r0, c1, r1, stops, extend=Extend.Pad): return Brush(Brush.RADIAL, p0=c0, p1=c1, r0=r0, r1=r1, stops=stops, extend=extend)
    @staticmethod
    def sweep(center, t0, t1, stops, extend=Extend.Pad): return Brush(Brush.SWEEP, p0=center, t0=t0, t1=t1, stops=stops, extend=extend)
    @staticmethod
    def image(pixels_rgba8, key=None):
        """`key` is the image's identity for de-duplication (the Go code keys on the image.Image pointer).  The default
        is derived from the CONTENTS (a 63-bit BLAKE2b of shape + bytes, top bit set to mark it): the same pixels wrapped in
        a Brush N times are one atlas entry, as one image.Image drawn N times is in the reference -- and never the array's
        address, which another array can reuse after this one is freed.  The C++ Scene copies the pixels inside fill /
        stroke, so the array only has to live until that call returns."""
        px = np.ascontiguousarray(pixels_rgba8, dtype=np.uint8)
        if px.ndim != 3 or px.shape[2] != 4:
            raise ValueError("image pixels must be (height, width, 4) uint8")
        if key is None:
            import hashlib
            h = hashlib.blake2b(digest_size=8)
            h.update(np.asarray(px.shape, dtype=np.uint64).tobytes())
            h.update(px.tobytes())
            key = int.from_bytes(h.digest(), "little") | (1 << 63)
        elif int(key) >> 63:
            raise ValueError("image keys with the top bit set are reserved for content-derived keys")
        return Brush(Brush.IMAGE, pixels=px, key=int(key))

    def _c(self):
        b = CBrush()
        b.kind = self.kind
        kw = self.kw
        b.extend = int(kw.get("extend", 0))
        for i, v in enumerate(kw.get("color", (0, 0, 0, 0))):
            b.color[i] = v
        for i in range(2):
            b.p0[i] = kw.get("p0", (0, 0))[i]
            b.p1[i] = kw.get("p1", (0, 0))[i]
        b.r0, b.r1, b.t0, b.t1 = kw.get("r0", 0), kw.get("r1", 0), kw.get("t0", 0), kw.get("t1", 0)
        keep = []
        stops = kw.get("stops") or []
        if stops:
            arr = (CColorStop * len(stops))()
            for i, s in enumerate(stops):
                arr[i].offset = s.offset
                for j in range(4):
                    arr[i].rgba[j] = s.color[j]
            b.stops = arr
            b.n_stops = len(stops)
            keep.append(arr)
        if "pixels" in kw:
            px = kw["pixels"]
            b.image_height, b.image_width = px.shape[0], px.shape[1]
            b.image_pixels = px.ctypes.data
            b.image_key = kw["key"]
            keep.append(px)
        return b, keep


class Stroke:
    """curve.Stroke subset: width, join, miter_limit, caps (dashes unsupported, see host/gfx.h)."""
    def __init__(self, width=1.0, join=Join.Round, miter_limit=4.0, start_cap=Cap.Round, end_cap=Cap.Round):
        self.width, self.join, self.miter_limit, self.start_cap, self.end_cap = width, join, miter_limit, start_cap, end_cap

    def _c(self):
        s = CStroke()
        s.width, s.join, s.start_cap, s.end_cap, s.miter_limit = self.width, int(self.join), int(self.start_cap), int(self.end_cap), self.miter_limit
        return s


class BumpSizes:
    """Element counts of the bump-allocated buffers; defaults are the reference's constants (config.go:144-151)."""
    FIELDS = ("bin_data", "tiles", "lines", "seg_counts", "segments", "blend_spill", "ptcl")

    def __init__(self, bin_data=1 << 18, tiles=1 << 21, lines=1 << 21, seg_counts=1 << 21, segments=1 << 21, blend_spill=1 << 21, ptcl=1 << 23):
        self.bin_data, self.tiles, self.lines, self.seg_counts, self.segments, self.blend_spill, self.ptcl = (
            bin_data, tiles, lines, seg_counts, segments, blend_spill, ptcl)

    def as_dict(self):
        return {f: getattr(self, f) for f in self.FIELDS}


class RenderParams:
    """renderer.RenderParams (render.go:58-63) + bump buffer sizes."""
    def __init__(self, width, height, base_color=(0, 0, 0, 0), aa=Aa.Area, bump=None):
        self.width, self.height, self.base_color, self.aa, self.bump = width, height, base_color, aa, bump or BumpSizes()

    def _c(self):
        p = CRenderParams()
        for i in range(4):
            p.base_color[i] = self.base_color[i]
        p.width, p.height, p.aa = self.width, self.height, int(self.aa)
        for f in BumpSizes.FIELDS:
            setattr(p.bump, f, int(getattr(self.bump, f)))
        return p


def _aff(t):
    if t is None:
        t = IDENTITY
    return (ctypes.c_double * 6)(*t)


class Scene:
    def __init__(self):
        self._L = _lib.load_host()
        self._h = self._L.jl_scene_new()

    def __del__(self):
        try:
            self._L.jl_scene_free(self._h)
        except Exception:
            pass

    def reset(self):
        self._L.jl_scene_reset(self._h)

    def _check(self, rc):
        if rc != 0:
            raise ValueError(self._L.jl_last_error().decode())

    def fill(self, style, transform, brush, brush_transform, path):
        b, keep = brush._c()
        els = path._c()
        self._check(self._L.jl_scene_fill(self._h, int(style), _aff(transform), ctypes.byref(b), _aff(brush_transform), els, len(path.els)))

    def stroke(self, style, transform, brush, brush_transform, path):
        b, keep = brush._c()
        els = path._c()
        s = style._c()
        self._check(self._L.jl_scene_stroke(self._h, ctypes.byref(s), _aff(transform), ctypes.byref(b), _aff(brush_transform), els, len(path.els)))

    def push_layer(self, mix, compose, alpha, transform, clip):
        els = clip._c()
        self._check(self._L.jl_scene_push_layer(self._h, int(mix), int(compose), float(alpha), _aff(transform), els, len(clip.els)))

    def pop_layer(self):
        self._L.jl_scene_pop_layer(self._h)

    def append(self, other, transform=None):
        self._L.jl_scene_append(self._h, other._h, _aff(transform))

    def apply_transform(self, transform):
        self._L.jl_scene_apply_transform(self._h, _aff(transform))

    # ---- raw encoding streams (encoding.Encoding fields) ----
    def stream(self, which):
        names = {"path_tags": 0, "path_data": 1, "draw_tags": 2, "draw_data": 3, "transforms": 4, "styles": 5}
        p = ctypes.c_void_p()
        n = self._L.jl_scene_stream(self._h, names[which], ctypes.byref(p))
        if n == 0:
            return b""
        return ctypes.string_at(p.value, n)

    def bump_sizes(self, width, height):
        """Buffer sizes for a width x height render from the scene's BumpEstimator (scene.go:36-43, renderer/estimate.go)
        plus the bounding-box bounds for tiles / bin data / PTCL: the first attempt normally fits."""
        from ._lib import CBumpSizes
        out = CBumpSizes()
        self._L.jl_scene_bump_sizes(self._h, int(width), int(height), ctypes.byref(out))
        return BumpSizes(**{f: getattr(out, f) for f in BumpSizes.FIELDS})

    def bump_sizes_clamped(self, width, height):
        """The fields of bump_sizes() that were held below the estimator's bounds (the first attempt is capped at 16 x the
        reference's constants).  Empty: the sizes are the bounds.  Otherwise a caller without the regrow loop (hipGraph capture, a
        timed loop) should render once with robust=True and keep the sizes that render ended with."""
        mask = int(self._L.jl_scene_bump_sizes_clamped(self._h, int(width), int(height)))
        return [f for i, f in enumerate(BumpSizes.FIELDS) if mask >> i & 1]

    def bump_estimate(self, transform=None):
        """The raw BumpEstimator tally (renderer/estimate.go:173-197)."""
        out = (ctypes.c_uint32 * 7)()
        self._L.jl_scene_bump_estimate(self._h, None if transform is None else _aff(transform), out)
        return dict(zip(["binning", "ptcl", "tile", "blend", "seg_counts", "segments", "lines"], out))

    def counts(self):
        out = (ctypes.c_uint32 * 4)()
        self._L.jl_scene_counts(self._h, out)
        return {"num_paths": out[0], "num_path_segments": out[1], "num_clips": out[2], "num_open_clips": out[3]}
