"""Pixel-level known answers of the fine stage, worked "by hand": every function here evaluates ONE pixel with one IEEE
binary32 operation per line (numpy.float32 scalars: no fused multiply-add, round-to-nearest-even), transcribed from the
WGSL text and nothing else -- it shares no code with oracle/oracle.cpp or jello_amd/csrc/kernels_fine.hip, handles no
tiles, no PTCL and no lanes, and logs every intermediate so that tests/golden/kat_pixels.json can show the derivation.

Sources: engine/wgpu_engine/shaders/original/fine.wgsl:824-878 (fill_path), :923-926 (CMD_COLOR), :978-987 (CMD_LIN_GRAD),
:800-812 (extend_mode), :1092-1102 (un-premultiply + rgba16float store), shared/blend.wgsl:142-195 (blend_mix),
:216-310 (blend_compose / blend_mix_compose).
"""
import numpy as np

f32 = np.float32


def bits(x):
    return "0x%08x" % int(np.float32(x).view(np.uint32))


class Log:
    def __init__(self):
        self.steps = []

    def __call__(self, name, v):
        v = f32(v)
        self.steps.append("%s = %r (%s)" % (name, float(v), bits(v)))
        return v


def clamp01(x):  # WGSL clamp(x, 0, 1) = min(max(x, 0), 1)
    return f32(min(max(f32(x), f32(0.0)), f32(1.0)))


def fill_area(segments, backdrop, px, py, even_odd=False, log=None):
    """fine.wgsl:824-878 for the pixel in column px, row py of the tile (segments are tile relative:
    (p0x, p0y, p1x, p1y, y_edge)).  xy.x is the first column of the pixel's group of four, i the pixel inside it."""
    log = log or Log()
    xy_x, i_f, xy_y = f32(4 * (px // 4)), f32(px % 4), f32(py)
    area = log("area = f32(backdrop)", f32(backdrop))
    for n, (p0x, p0y, p1x, p1y, y_edge) in enumerate(segments):
        p0x, p0y, p1x, p1y, y_edge = f32(p0x), f32(p0y), f32(p1x), f32(p1y), f32(y_edge)
        t = "seg%d " % n
        y = log(t + "y = p0.y - xy.y", p0y - xy_y)
        dx, dy_ = f32(p1x - p0x), f32(p1y - p0y)
        y0 = log(t + "y0 = clamp(y, 0, 1)", clamp01(y))
        y1 = log(t + "y1 = clamp(y + delta.y, 0, 1)", clamp01(f32(y + dy_)))
        dy = log(t + "dy = y0 - y1", y0 - y1)
        if dy != f32(0.0):
            recip = log(t + "vec_y_recip = 1 / delta.y", f32(1.0) / dy_)
            t0 = log(t + "t0 = (y0 - y) * vec_y_recip", f32(y0 - y) * recip)
            t1 = log(t + "t1 = (y1 - y) * vec_y_recip", f32(y1 - y) * recip)
            startx = log(t + "startx = p0.x - xy.x", p0x - xy_x)
            x0 = log(t + "x0 = startx + t0 * delta.x", startx + f32(t0 * dx))
            x1 = log(t + "x1 = startx + t1 * delta.x", startx + f32(t1 * dx))
            xmin0, xmax0 = f32(min(x0, x1)), f32(max(x0, x1))
            xmin = log(t + "xmin = min(xmin0 - i, 1) - 1e-6", f32(min(f32(xmin0 - i_f), f32(1.0))) - f32(1.0e-6))
            xmax = log(t + "xmax = xmax0 - i", xmax0 - i_f)
            b = log(t + "b = min(xmax, 1)", f32(min(xmax, f32(1.0))))
            c = log(t + "c = max(b, 0)", f32(max(b, f32(0.0))))
            d = log(t + "d = max(xmin, 0)", f32(max(xmin, f32(0.0))))
            num = log(t + "b + 0.5 * (d*d - c*c) - xmin", f32(f32(b + f32(f32(0.5) * f32(f32(d * d) - f32(c * c)))) - xmin))
            a = log(t + "a = that / (xmax - xmin)", num / f32(xmax - xmin))
            area = log(t + "area += a * dy", area + f32(a * dy))
        sgn = f32(1.0) if dx > 0 else (f32(-1.0) if dx < 0 else f32(0.0))
        ye = f32(sgn * clamp01(f32(f32(xy_y - y_edge) + f32(1.0))))
        if ye != f32(0.0):
            area = log(t + "area += sign(delta.x) * clamp(xy.y - y_edge + 1, 0, 1)", area + ye)
    if even_odd:
        area = log("area = abs(area - 2 * round(0.5 * area))", abs(f32(area - f32(f32(2.0) * f32(np.rint(f32(f32(0.5) * area)))))))
    else:
        area = log("area = min(abs(area), 1)", f32(min(abs(area), f32(1.0))))
    return area, log


def over(bg, fg, area, log=None):
    """fine.wgsl:923-926: fg_i = fg * area; rgba = rgba * (1 - fg_i.a) + fg_i"""
    log = log or Log()
    fg_i = [f32(f32(c) * f32(area)) for c in fg]
    k = log("1 - fg_i.a", f32(1.0) - fg_i[3])
    out = [log("rgba.%s = rgba.%s * k + fg_i.%s" % (n, n, n), f32(f32(b) * k) + fg_i[j]) for j, (n, b) in enumerate(zip("rgba", bg))]
    return out, log


def store_rgba16f(rgba, log=None):
    """fine.wgsl:1092-1102: a_inv = 1 / max(a, 1e-6); (rgb * a_inv, a) stored as rgba16float (round to nearest even)."""
    log = log or Log()
    a_inv = log("a_inv = 1 / max(a, 1e-6)", f32(1.0) / f32(max(f32(rgba[3]), f32(1.0e-6))))
    vals = [log("%s * a_inv" % n, f32(f32(rgba[j]) * a_inv)) for j, n in enumerate("rgb")] + [f32(rgba[3])]
    halves = [int(np.float16(v).view(np.uint16)) for v in vals]
    log.steps.append("rgba16float bits = " + " ".join("0x%04x" % h for h in halves))
    return halves, log


def extend_mode(t, mode):  # fine.wgsl:800-812: 0 pad, 1 repeat, 2 reflect
    t = f32(t)
    if mode == 0:
        return clamp01(t)
    if mode == 1:
        return f32(t - f32(np.floor(t)))
    return f32(abs(f32(t - f32(f32(2.0) * f32(np.rint(f32(f32(0.5) * t)))))))


def lin_grad_ramp_x(line_x, line_y, line_c, gx, gy, mode, log=None):
    """fine.wgsl:978-983 for the pixel at GLOBAL column gx, row gy: d from the first pixel of its group of four."""
    log = log or Log()
    xy_x, i_f, xy_y = f32(4 * (gx // 4)), f32(gx % 4), f32(gy)
    d = log("d = line_x * xy.x + line_y * xy.y + line_c", f32(f32(f32(line_x) * xy_x) + f32(f32(line_y) * xy_y)) + f32(line_c))
    my_d = log("my_d = d + line_x * i", d + f32(f32(line_x) * i_f))
    e = log("extend_mode(my_d)", extend_mode(my_d, mode))
    x = int(np.rint(log("extend * 511", e * f32(511.0))))
    log.steps.append("ramp x = round(...) = %d" % x)
    return x, log


def lum(c):  # blend.wgsl: dot(c, vec3(0.3, 0.59, 0.11)) evaluated left to right
    return f32(f32(f32(f32(c[0]) * f32(0.3)) + f32(f32(c[1]) * f32(0.59))) + f32(f32(c[2]) * f32(0.11)))


def clip_color(c):
    L = lum(c)
    n, x = f32(min(c)), f32(max(c))
    c = [f32(v) for v in c]
    if n < 0:
        c = [f32(L + f32(f32(f32(v - L) * L) / f32(L - n))) for v in c]
    if x > 1:
        c = [f32(L + f32(f32(f32(v - L) * f32(f32(1.0) - L)) / f32(x - L))) for v in c]
    return c


def set_lum(c, l):
    d = f32(f32(l) - lum(c))
    return clip_color([f32(f32(v) + d) for v in c])


def blend_mix_compose_srcover(backdrop, src, mix, log=None):
    """shared/blend.wgsl:288-310 with compose = SrcOver, for mix = "multiply" (1) or "luminosity" (15)."""
    log = log or Log()
    inv_src_a = log("inv_src_a = 1 / max(src.a, 1e-15)", f32(1.0) / f32(max(f32(src[3]), f32(1e-15))))
    cs = [log("cs.%s = src.%s * inv_src_a" % (n, n), f32(f32(src[j]) * inv_src_a)) for j, n in enumerate("rgb")]
    inv_b_a = log("inv_backdrop_a = 1 / max(backdrop.a, 1e-15)", f32(1.0) / f32(max(f32(backdrop[3]), f32(1e-15))))
    cb = [log("cb.%s = backdrop.%s * inv_backdrop_a" % (n, n), f32(f32(backdrop[j]) * inv_b_a)) for j, n in enumerate("rgb")]
    if mix == "multiply":
        mixed = [log("mixed.%s = cb * cs" % n, f32(cb[j] * cs[j])) for j, n in enumerate("rgb")]
    elif mix == "luminosity":
        l = log("lum(cs)", lum(cs))
        mixed = [log("mixed.%s = set_lum(cb, lum(cs))" % n, v) for n, v in zip("rgb", set_lum(cb, l))]
    else:
        raise ValueError(mix)
    ba, sa = f32(backdrop[3]), f32(src[3])
    cs2 = [log("cs'.%s = mix(cs, mixed, backdrop.a)" % n, f32(f32(cs[j] * f32(f32(1.0) - ba)) + f32(mixed[j] * ba))) for j, n in enumerate("rgb")]
    out = [log("out.%s = mix(backdrop, cs', src.a)" % n, f32(f32(f32(backdrop[j]) * f32(f32(1.0) - sa)) + f32(cs2[j] * sa))) for j, n in enumerate("rgb")]
    out.append(log("out.a = src.a + backdrop.a * (1 - src.a)", sa + f32(ba * f32(f32(1.0) - sa))))
    return out, log
