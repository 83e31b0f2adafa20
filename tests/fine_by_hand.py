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


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the remaining mix modes, gradients, images.  Sources: shared/blend.wgsl:24-195 (mix functions), :288-310;
# draw_leaf.wgsl:151-247 (radial / sweep / image info words), shared/transform.wgsl, fine.wgsl:988-1087.
# ---------------------------------------------------------------------------------------------------------------------
def _screen(cb, cs):  # cb + cs - (cb * cs)
    return f32(f32(cb + cs) - f32(cb * cs))


def _hard_light(cb, cs):  # select(screen(cb, 2 cs - 1), cb * 2 * cs, cs <= 0.5)
    if cs <= f32(0.5):
        return f32(f32(cb * f32(2.0)) * cs)
    return _screen(cb, f32(f32(f32(2.0) * cs) - f32(1.0)))


def _color_dodge(cb, cs):
    if cb == f32(0.0):
        return f32(0.0)
    if cs == f32(1.0):
        return f32(1.0)
    return f32(min(f32(1.0), f32(cb / f32(f32(1.0) - cs))))


def _color_burn(cb, cs):
    if cb == f32(1.0):
        return f32(1.0)
    if cs == f32(0.0):
        return f32(0.0)
    return f32(f32(1.0) - f32(min(f32(1.0), f32(f32(f32(1.0) - cb) / cs))))


def _soft_light(cb, cs):
    if cb <= f32(0.25):
        d = f32(f32(f32(f32(f32(f32(16.0) * cb) - f32(12.0)) * cb) + f32(4.0)) * cb)
    else:
        d = f32(np.sqrt(cb))
    if cs <= f32(0.5):
        return f32(cb - f32(f32(f32(f32(1.0) - f32(f32(2.0) * cs)) * cb) * f32(f32(1.0) - cb)))
    return f32(cb + f32(f32(f32(f32(2.0) * cs) - f32(1.0)) * f32(d - cb)))


def sat(c):  # max(c.x, max(c.y, c.z)) - min(c.x, min(c.y, c.z))
    return f32(f32(max(c[0], max(c[1], c[2]))) - f32(min(c[0], min(c[1], c[2]))))


def set_sat(c, s):
    """shared/blend.wgsl:92-141: the smallest channel becomes 0, the largest s, the middle one scales."""
    c = [f32(v) for v in c]
    s = f32(s)

    def inner(imin, imid, imax):
        if c[imax] > c[imin]:
            c[imid] = f32(f32(f32(c[imid] - c[imin]) * s) / f32(c[imax] - c[imin]))
            c[imax] = s
        else:
            c[imid] = f32(0.0)
            c[imax] = f32(0.0)
        c[imin] = f32(0.0)
    r, g, b = 0, 1, 2
    if c[r] <= c[g]:
        if c[g] <= c[b]:
            inner(r, g, b)
        elif c[r] <= c[b]:
            inner(r, b, g)
        else:
            inner(b, r, g)
    else:
        if c[r] <= c[b]:
            inner(g, r, b)
        elif c[g] <= c[b]:
            inner(g, b, r)
        else:
            inner(b, g, r)
    return c


MIX_NAMES = ["normal", "multiply", "screen", "overlay", "darken", "lighten", "color_dodge", "color_burn", "hard_light", "soft_light",
             "difference", "exclusion", "hue", "saturation", "color", "luminosity"]


def blend_mix(cb, cs, mix):
    """shared/blend.wgsl:142-195, per channel where the mode is separable."""
    cb, cs = [f32(v) for v in cb], [f32(v) for v in cs]
    per = {"multiply": lambda b, s: f32(b * s), "screen": _screen, "overlay": lambda b, s: _hard_light(s, b),
           "darken": lambda b, s: f32(min(b, s)), "lighten": lambda b, s: f32(max(b, s)), "color_dodge": _color_dodge,
           "color_burn": _color_burn, "hard_light": _hard_light, "soft_light": _soft_light,
           "difference": lambda b, s: f32(abs(f32(b - s))),
           "exclusion": lambda b, s: f32(f32(b + s) - f32(f32(f32(2.0) * b) * s))}
    if mix in per:
        return [per[mix](b, s) for b, s in zip(cb, cs)]
    if mix == "hue":
        return set_lum(set_sat(cs, sat(cb)), lum(cb))
    if mix == "saturation":
        return set_lum(set_sat(cb, sat(cs)), lum(cb))
    if mix == "color":
        return set_lum(cs, lum(cb))
    if mix == "luminosity":
        return set_lum(cb, lum(cs))
    return cs


def blend_mix_compose(backdrop, src, mix, log=None):
    """shared/blend.wgsl:288-310 with compose = SrcOver and any mix mode ("normal" takes the early return of :291-294)."""
    log = log or Log()
    if mix == "normal":
        k = log("1 - src.a", f32(1.0) - f32(src[3]))
        return [log("out.%s = backdrop.%s * (1 - src.a) + src.%s" % (n, n, n), f32(f32(f32(backdrop[j]) * k) + f32(src[j]))) for j, n in enumerate("rgba")], log
    inv_src_a = log("inv_src_a = 1 / max(src.a, 1e-15)", f32(1.0) / f32(max(f32(src[3]), f32(1e-15))))
    cs = [log("cs.%s = src.%s * inv_src_a" % (n, n), f32(f32(src[j]) * inv_src_a)) for j, n in enumerate("rgb")]
    inv_b_a = log("inv_backdrop_a = 1 / max(backdrop.a, 1e-15)", f32(1.0) / f32(max(f32(backdrop[3]), f32(1e-15))))
    cb = [log("cb.%s = backdrop.%s * inv_backdrop_a" % (n, n), f32(f32(backdrop[j]) * inv_b_a)) for j, n in enumerate("rgb")]
    mixed = [log("mixed.%s (%s)" % (n, mix), v) for n, v in zip("rgb", blend_mix(cb, cs, mix))]
    ba, sa = f32(backdrop[3]), f32(src[3])
    cs2 = [log("cs'.%s = mix(cs, mixed, backdrop.a)" % n, f32(f32(cs[j] * f32(f32(1.0) - ba)) + f32(mixed[j] * ba))) for j, n in enumerate("rgb")]
    out = [log("out.%s = mix(backdrop, cs', src.a)" % n, f32(f32(f32(backdrop[j]) * f32(f32(1.0) - sa)) + f32(cs2[j] * sa))) for j, n in enumerate("rgb")]
    out.append(log("out.a = src.a + backdrop.a * (1 - src.a)", sa + f32(ba * f32(f32(1.0) - sa))))
    return out, log


# ---- shared/transform.wgsl on (matrx[4], translate[2]) tuples ----
def xf_inverse(t):
    m, tr = [f32(v) for v in t[0]], [f32(v) for v in t[1]]
    inv_det = f32(f32(1.0) / f32(f32(m[0] * m[3]) - f32(m[1] * m[2])))
    im = [f32(inv_det * m[3]), f32(inv_det * f32(-m[1])), f32(inv_det * f32(-m[2])), f32(inv_det * m[0])]
    ntx, nty = f32(-tr[0]), f32(-tr[1])  # mat2x2(inv_mat.xy, inv_mat.zw) * -translate = xy * ntx + zw * nty
    return (im, [f32(f32(im[0] * ntx) + f32(im[2] * nty)), f32(f32(im[1] * ntx) + f32(im[3] * nty))])


def xf_mul(a, b):
    am, at, bm, bt = a[0], a[1], b[0], b[1]
    m = [f32(f32(am[0] * bm[0]) + f32(am[2] * bm[1])), f32(f32(am[1] * bm[0]) + f32(am[3] * bm[1])),
         f32(f32(am[0] * bm[2]) + f32(am[2] * bm[3])), f32(f32(am[1] * bm[2]) + f32(am[3] * bm[3]))]
    t = [f32(f32(f32(am[0] * bt[0]) + f32(am[2] * bt[1])) + at[0]), f32(f32(f32(am[1] * bt[0]) + f32(am[3] * bt[1])) + at[1])]
    return (m, t)


def xf_apply(t, x, y):  # matrx.xy * p.x + matrx.zw * p.y + translate
    m, tr = t
    return f32(f32(f32(m[0] * f32(x)) + f32(m[2] * f32(y))) + tr[0]), f32(f32(f32(m[1] * f32(x)) + f32(m[3] * f32(y))) + tr[1])


IDENT = ([f32(1.0), f32(0.0), f32(0.0), f32(1.0)], [f32(0.0), f32(0.0)])


def _from_poly2(p0, p1):
    return ([f32(p1[1] - p0[1]), f32(p0[0] - p1[0]), f32(p1[0] - p0[0]), f32(p1[1] - p0[1])], [f32(p0[0]), f32(p0[1])])


def _two_point_to_unit_line(p0, p1):
    return xf_mul(_from_poly2([f32(0.0), f32(0.0)], [f32(1.0), f32(0.0)]), xf_inverse(_from_poly2(p0, p1)))


def _distance(a, b):
    dx, dy = f32(a[0] - b[0]), f32(a[1] - b[1])
    return f32(np.sqrt(f32(f32(dx * dx) + f32(dy * dy))))


def rad_grad_info(p0, p1, r0, r1, log=None):
    """draw_leaf.wgsl:151-222 under the identity transform: (xform, focal_x, radius, kind, flags).
    kind (shared/config.wgsl:67-70): 1 circular, 2 strip, 3 focal-on-circle, 4 cone; flags bit 0 = swapped (:73)."""
    log = log or Log()
    EPS = f32(1.0) / f32(4096.0)
    p0, p1, r0, r1 = [f32(v) for v in p0], [f32(v) for v in p1], f32(r0), f32(r1)
    user_to_gradient = xf_inverse(IDENT)
    flags = 0
    if abs(f32(r0 - r1)) <= EPS:
        scaled = f32(r0 / _distance(p0, p1))
        return xf_mul(_two_point_to_unit_line(p0, p1), user_to_gradient), f32(0.0), f32(scaled * scaled), 2, 0, log
    kind = 4
    if p0[0] == p1[0] and p0[1] == p1[1]:
        kind = 1
        p0 = [f32(p0[0] + EPS), f32(p0[1] + EPS)]
    if r1 == f32(0.0):
        flags |= 1
        p0, p1, r0, r1 = p1, p0, r1, r0
    focal_x = log("focal_x = r0 / (r0 - r1)", f32(r0 / f32(r0 - r1)))
    omf = f32(f32(1.0) - focal_x)
    cf = [f32(f32(omf * p0[0]) + f32(focal_x * p1[0])), f32(f32(omf * p0[1]) + f32(focal_x * p1[1]))]
    radius = log("radius = r1 / distance(cf, p1)", f32(r1 / _distance(cf, p1)))
    unit = xf_mul(_two_point_to_unit_line(cf, p1), user_to_gradient)
    if abs(f32(radius - f32(1.0))) <= EPS:
        kind = 3
        scale = f32(f32(0.5) * abs(f32(f32(1.0) - focal_x)))
        xform = xf_mul(([scale, f32(0.0), f32(0.0), scale], [f32(0.0), f32(0.0)]), unit)
    else:
        a = f32(f32(radius * radius) - f32(1.0))
        scale_ratio = f32(abs(f32(f32(1.0) - focal_x)) / a)
        scale_x = f32(radius * scale_ratio)
        scale_y = f32(f32(np.sqrt(abs(a))) * scale_ratio)
        xform = xf_mul(([scale_x, f32(0.0), f32(0.0), scale_y], [f32(0.0), f32(0.0)]), unit)
    for n, v in zip(("m0", "m1", "m2", "m3"), xform[0]):
        log("xform." + n, v)
    log("xform.tx", xform[1][0]); log("xform.ty", xform[1][1])
    return xform, focal_x, radius, kind, flags, log


def rad_grad_ramp_x(xform, focal_x, radius, kind, flags, gx, gy, mode, log=None):
    """fine.wgsl:991-1036 for the pixel at GLOBAL (gx, gy): the ramp column, or None where the gradient is not valid."""
    log = log or Log()
    is_swapped = (flags & 1) != 0
    r1_recip = f32(0.0) if kind == 1 else f32(f32(1.0) / radius)
    less_scale = f32(-1.0) if (is_swapped or f32(f32(1.0) - focal_x) < 0) else f32(1.0)
    omf = f32(f32(1.0) - focal_x)
    t_sign = f32(1.0) if omf > 0 else (f32(-1.0) if omf < 0 else f32(0.0))
    x, y = xf_apply(xform, gx, gy)
    log("local x", x); log("local y", y)
    xx, yy = f32(x * x), f32(y * y)
    valid = True
    if kind == 2:
        a = f32(radius - yy)
        with np.errstate(invalid="ignore"):
            t = f32(f32(np.sqrt(a)) + x)
        valid = a >= 0
    elif kind == 3:
        t = f32(f32(xx + yy) / x)
        valid = t >= 0 and x != 0
    elif radius > f32(1.0):
        t = f32(f32(np.sqrt(f32(xx + yy))) - f32(x * r1_recip))
    else:
        a = f32(xx - yy)
        with np.errstate(invalid="ignore"):
            t = f32(f32(less_scale * f32(np.sqrt(a))) - f32(x * r1_recip))
        valid = a >= 0 and t >= 0
    if not valid:
        log.steps.append("not valid: the pixel keeps its backdrop")
        return None, log
    log("t", t)
    t = log("extend_mode(focal_x + t_sign * t)", extend_mode(f32(focal_x + f32(t_sign * t)), mode))
    if is_swapped:
        t = log("1 - t (swapped)", f32(f32(1.0) - t))
    xi = int(np.rint(log("t * 511", f32(t * f32(511.0)))))
    log.steps.append("ramp x = %d" % xi)
    return xi, log


def sweep_info(center, log=None):
    """draw_leaf.wgsl:223-235 under the identity transform: inverse of translate(center)."""
    return xf_inverse(xf_mul(IDENT, ([f32(1.0), f32(0.0), f32(0.0), f32(1.0)], [f32(center[0]), f32(center[1])])))


def sweep_ramp_x(inv, t0, t1, gx, gy, mode, log=None):
    """fine.wgsl:1038-1066."""
    log = log or Log()
    scale = f32(f32(1.0) / f32(f32(t1) - f32(t0)))
    x, y = xf_apply(inv, gx, gy)
    xabs, yabs = f32(abs(x)), f32(abs(y))
    slope = log("slope = min / max", f32(f32(min(xabs, yabs)) / f32(max(xabs, yabs))))
    s = f32(slope * slope)
    c0, c1, c2, c3 = f32(0.15912117063999176025390625), f32(-5.185396969318389892578125e-2), f32(2.476101927459239959716796875e-2), f32(-7.0547382347285747528076171875e-3)
    phi = f32(slope * f32(c0 + f32(s * f32(c1 + f32(s * f32(c2 + f32(s * c3)))))))
    log("phi (first octant)", phi)
    if xabs < yabs:
        phi = f32(f32(0.25) - phi)
    if x < 0:
        phi = f32(f32(0.5) - phi)
    if y < 0:
        phi = f32(f32(1.0) - phi)
    if phi != phi:
        phi = f32(0.0)
    log("phi", phi)
    phi = log("(phi - t0) * scale", f32(f32(phi - f32(t0)) * scale))
    t = log("extend_mode", extend_mode(phi, mode))
    xi = int(np.rint(log("t * 511", f32(t * f32(511.0)))))
    log.steps.append("ramp x = %d" % xi)
    return xi, log


def srgb8_to_linear(v):
    """IEC 61966-2-1 decoding of one 8-bit channel, evaluated in binary64 and rounded once (an rgba8unorm-srgb texel)."""
    c = v / 255.0
    return f32(c / 12.92 if c <= 0.04045 else ((c + 0.055) / 1.055) ** 2.4)


def image_pixel(pixels, inv, gx, gy, log=None):
    """fine.wgsl:1068-1087 for the pixel at GLOBAL (gx, gy) with area 1: the premultiplied bilinear sample, or None outside the
    image.  pixels: (h, w, 4) uint8, sRGB-encoded colour + linear alpha (render.go:137 uploads scene images as Rgba8Srgb)."""
    log = log or Log()
    h, w = pixels.shape[0], pixels.shape[1]
    u, v = xf_apply(inv, gx, gy)
    log("uv.x", u); log("uv.y", v)
    if not (u < f32(w) and v < f32(h)):
        return None, log
    x0, y0, x1, y1 = int(np.floor(u)), int(np.floor(v)), int(np.ceil(u)), int(np.ceil(v))
    fx, fy = log("fract(uv.x)", f32(u - f32(np.floor(u)))), log("fract(uv.y)", f32(v - f32(np.floor(v))))

    def texel(x, y):  # textureLoad is zero outside the texture (robust access); premul_alpha
        if not (0 <= x < w and 0 <= y < h):
            return [f32(0.0)] * 4
        p = pixels[y, x]
        a = f32(f32(int(p[3])) / f32(255.0))
        return [f32(srgb8_to_linear(int(p[0])) * a), f32(srgb8_to_linear(int(p[1])) * a), f32(srgb8_to_linear(int(p[2])) * a), a]

    def mix4(p, q, t):
        return [f32(f32(p[j] * f32(f32(1.0) - t)) + f32(q[j] * t)) for j in range(4)]
    a, b, c, d = texel(x0, y0), texel(x0, y1), texel(x1, y0), texel(x1, y1)
    out = mix4(mix4(a, b, fy), mix4(c, d, fy), fx)
    for n, val in zip("rgba", out):
        log("fg." + n, val)
    return out, log
