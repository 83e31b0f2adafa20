"""Writes tests/golden/kat_pixels.json: the pixel-level known answers of the fine stage with their derivations, every number
produced by tests/fine_by_hand.py (one IEEE binary32 operation per line, transcribed from the WGSL) or by the integer /
dyadic arithmetic written out in the "derivation" strings -- never by the oracle or the HIP kernels.
   python tests/golden/make_kat_pixels.py
tests/test_kat_pixels.py re-derives the file on every run (so it cannot drift from the derivation) and holds the oracle
to it; tests/test_gpu_kat.py holds the HIP image to it."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fine_by_hand as H  # noqa: E402


def kat_image_pixels():
    """The 4x4 RGBA8 image of the image-brush known answer."""
    import numpy as np
    px = np.zeros((4, 4, 4), np.uint8)
    for y in range(4):
        for x in range(4):
            px[y, x] = (60 * x + 15, 50 * y + 30, 255 - 40 * x - 20 * y, 128 if (x + y) & 1 else 255)
    return px


def h16(hs):
    return ["0x%04x" % v for v in hs]


def build():
    out = {"_about": "Pixel-level known answers for fine (and two flatten join cases), worked by hand: see tests/fine_by_hand.py. "
                     "Colours are premultiplied linear RGBA; pixel values are rgba16float bit patterns of the stored, un-premultiplied pixel."}
    # ---- 1. rectangle with edges at x.5 / y.25 / y.75 ---------------------------------------------------------------
    segs = [[2.5, 3.25, 9.5, 3.25, 1e9], [9.5, 3.25, 9.5, 7.75, 1e9], [9.5, 7.75, 2.5, 7.75, 1e9], [2.5, 7.75, 2.5, 3.25, 1e9]]
    fg = [1.0, 0.5, 0.25, 1.0]
    px, logs, areas = {}, {}, {}
    for (x, y) in [(5, 5), (2, 5), (5, 3), (2, 3), (1, 5), (9, 5), (5, 7), (9, 7), (10, 5), (5, 8)]:
        a, log = H.fill_area(segs, 0, x, y)
        rgba, log = H.over([0, 0, 0, 0], fg, a, log)
        hs, log = H.store_rgba16f(rgba, log)
        px["%d,%d" % (x, y)] = h16(hs)
        areas["%d,%d" % (x, y)] = float(a)
        logs["%d,%d" % (x, y)] = log.steps
    out["rect_fractional_edges"] = {
        "scene": "16x16 target, transparent base; Fill(NonZero, solid straight (1, .5, .25, 1)) of the rectangle (2.5, 3.25)-(9.5, 7.75)",
        "derivation": [
            "flatten: four LineTo segments, each degree-raised to a collinear cubic and accepted whole (SURVEY appendix D): 4 lines in path order.",
            "path_count / path_tiling: every line lies inside tile (0,0) and touches neither x = 0 nor a tile edge: one segment per line with the "
            "line's own end points (tile origin is (0,0)) and y_edge = 1e9 ('none', path_tiling.wgsl:166-169).",
            "fine.wgsl:824-878 per pixel (steps_by_pixel): the horizontal segments have dy = 0 in every row; the vertical ones contribute "
            "a * dy with a = the covered fraction of the pixel's width, dy = -+ the covered fraction of its height.",
            "expected areas: interior 1, left / right edge columns (x = 2, 9) 0.5, top / bottom edge rows (y = 3, 7) 0.75, corners 0.375, outside 0.",
            "colour: rgba = 0 * (1 - fg.a * area) + fg * area; stored (rgb / a, a): the un-premultiplied colour is (1, .5, .25) wherever "
            "area > 0 and alpha = area."],
        "segments_p0x_p0y_p1x_p1y_yedge": segs, "areas": areas, "pixels_rgba16f": px, "steps_by_pixel": {k: logs[k] for k in ("2,3", "9,5")}}
    # ---- 2. translucent colour over a non-zero base colour ---------------------------------------------------------
    base = [0.25, 0.125, 0.0625, 0.5]   # RenderParams.base_color (0.5, .25, .125, .5) premultiplied by the host
    fg = [0.25, 0.125, 0.0625, 0.5]     # Brush.solid((0.5, .25, .125, .5)) premultiplied by the encoder
    rgba, log = H.over(base, fg, 1.0)
    hs, log = H.store_rgba16f(rgba, log)
    out["translucent_over_base"] = {
        "scene": "16x16 target, base colour straight (0.5, .25, .125, .5); Fill(solid straight (0.5, .25, .125, .5)) of (0,0)-(16,16)",
        "derivation": ["the path covers the tile: coarse writes CMD_SOLID (area = 1) + CMD_COLOR with the premultiplied colour (.25, .125, .0625, .5)",
                       "config.base_color is the premultiplied base (.25, .125, .0625, .5); fine.wgsl:923-926 with area = 1:",
                       "k = 1 - 0.5 = 0.5; rgba = base * 0.5 + fg = (.375, .1875, .09375, .75): all exact in binary32",
                       "store: a_inv = 1 / 0.75; rgb * a_inv = (.5, .25, .125) after rounding to binary16; a = .75"],
        "ptcl_words_1_to_7": ["0x00000003", "0x00000005", "0x3e800000", "0x3e000000", "0x3d800000", "0x3f000000", "0x00000000"],
        "pixels_rgba16f": {"7,7": h16(hs), "0,0": h16(hs), "15,15": h16(hs)}, "steps": log.steps}
    # ---- 3. linear gradient: ramp index under Pad / Repeat / Reflect -------------------------------------------------
    ramp_x, steps = {}, {}
    for (gx, gy, mode) in [(3, 2, 0), (10, 2, 0), (10, 7, 1), (10, 12, 2), (13, 12, 2)]:
        x, log = H.lin_grad_ramp_x(0.125, 0.0, 0.0, gx, gy, mode)
        ramp_x["%d,%d" % (gx, gy)] = x
        steps["%d,%d" % (gx, gy)] = log.steps
    out["linear_gradient_extend"] = {
        "scene": "16x16 target; three Fill(linear gradient p0 = (0,0), p1 = (8,0), red -> blue) of the row bands y in [0,5) Pad, [5,10) Repeat, "
                 "[10,16) Reflect",
        "derivation": ["draw_leaf.wgsl (linear gradient info): dxy = p1 - p0 = (8, 0); scale = 1 / dot(dxy, dxy) = 1/64; line_xy = dxy * scale = "
                       "(0.125, 0); line_c = -dot(p0, line_xy) = -0",
                       "fine.wgsl:978-983: my_d = 0.125 * X for pixel column X (exact); ramp x = round(extend(my_d) * 511):",
                       "X = 3 Pad: 0.375 * 511 = 191.625 -> 192;  X = 10 Pad: clamp(1.25) = 1 -> 511;  X = 10 Repeat: fract(1.25) = 0.25 -> "
                       "round(127.75) = 128;  X = 10 Reflect: |1.25 - 2 * round(0.625)| = 0.75 -> round(383.25) = 383;  X = 13 Reflect: "
                       "|1.625 - 2 * round(0.8125)| = 0.375 -> 192",
                       "rows 2, 7, 12 are strictly inside their bands: area = 1; the ramp texels are opaque, so the stored pixel equals the texel "
                       "(the test reads the ramp row from the recording's own upload)"],
        "info_line_x_line_y_line_c": ["0x3e000000", "0x00000000", "0x80000000"], "ramp_x": ramp_x, "steps_by_pixel": steps}
    # ---- 4. END_CLIP blends ------------------------------------------------------------------------------------------
    backdrop, src = [0.5, 0.25, 0.75, 1.0], [0.5, 0.5, 0.25, 1.0]
    for mix in ("multiply", "luminosity"):
        o, log = H.blend_mix_compose_srcover(backdrop, src, mix)
        hs, log = H.store_rgba16f(o, log)
        out["blend_" + mix] = {
            "scene": "16x16 target, opaque base colour (.5, .25, .75, 1); PushLayer(%s, SrcOver, alpha 1, clip (0,0)-(16,16)); "
                     "Fill(solid (.5, .5, .25, 1)) of (0,0)-(16,16); PopLayer" % mix,
            "derivation": ["BEGIN_CLIP saves rgba = base and restarts from 0; the fill leaves rgba = (.5, .5, .25, 1); END_CLIP: "
                           "fg = rgba * area(1) * alpha(1), rgba = blend_mix_compose(saved base, fg, mix << 8 | SrcOver) (fine.wgsl:951-972)",
                           "shared/blend.wgsl:288-310 step by step in `steps`" +
                           ("; multiply: mixed = cb * cs = (.25, .125, .1875) exactly, and with backdrop.a = src.a = 1 both mix() calls return their "
                            "second argument: out = (.25, .125, .1875, 1)" if mix == "multiply" else
                            "; luminosity: mixed = set_lum(cb, lum(cs)) = cb + (lum(cs) - lum(cb)) per channel (clip_color leaves it: min >= 0, max <= 1)")],
            "result_premultiplied": [float(v) for v in o], "pixels_rgba16f": {"7,7": h16(hs)}, "steps": log.steps}
    # ---- 5. MSAA8: a vertical edge through the middle of a pixel --------------------------------------------------------
    pattern = [0, 5, 3, 7, 1, 4, 6, 2]
    xs = [(p + 0.5) / 8 for p in pattern]
    inside = [x > 0.5 for x in xs]
    out["msaa8_half_pixel"] = {
        "scene": "16x16 target, 8-sample coverage; Fill(opaque white) of (4.5, 0)-(12, 16)",
        "derivation": ["sample k of a pixel sits at x = (pattern[k] + 0.5) / 8 with pattern = [0,5,3,7,1,4,6,2] (renderer/mask.go:43-61): "
                       + ", ".join("%.4f" % x for x in xs),
                       "pixel (4, 8) spans x in [4, 5]; the path covers x > 4.5: samples %s are inside = %d of 8 -> area = %d/8"
                       % ([k for k, i in enumerate(inside) if i], sum(inside), sum(inside)),
                       "rgba = white * 0.5 = (.5, .5, .5, .5); stored (rgb / a, a) = (1, 1, 1, .5); pixel (5, 8) is fully inside, (3, 8) outside"],
        "pixels_rgba16f": {"4,8": ["0x3c00", "0x3c00", "0x3c00", "0x3800"], "5,8": ["0x3c00"] * 4, "3,8": ["0x0000"] * 4}}
    # ---- 6. flatten: the EPS = 1e-12 start-tangent rule at a round join ---------------------------------------------------
    import math
    theta = 2 * math.acos(1 - 0.25 / 4)
    c, s = math.cos(theta), math.sin(theta)
    p1 = (30 + 4 * c, 20 - 4 * s)
    p2 = (30 + 4 * (c * c - s * s), 20 - 4 * (2 * s * c))
    out["eps_tangent_round_join"] = {
        "scene": "Stroke(width 8, round join, butt caps) of MoveTo(20,20) LineTo(30,20) CubicTo((30, 20 + h), (40,20), (50,20)) with "
                 "h = 1e-4 and with h = 1e-7, identity transform",
        "derivation": ["flatten.wgsl:294-299 cubic_start_tangent: d01 = p1 - p0 = (0, h) is used when dot(d01, d01) > 1e-12 (the Go twin's "
                       "threshold is 2e-7, cpu/flatten.go:54-73).  h = 1e-4: 1e-8 > 1e-12, the cubic starts along +y; h = 1e-7: 1e-14 < 1e-12, "
                       "the rule falls through to d02 = (10, -h'): along +x, like the line in front of it.",
                       "h = 1e-4, join at p0 = (30,20) (flatten.wgsl:545-614): tan_prev = +x, tan_next = +y: n_prev = 4 * (0, 1), n_next = 4 * (-1, 0); "
                       "cr = 1 > 0 -> the arc runs on the back side from back0 = p0 - n_next = (34, 20) to back1 = p0 - n_prev = (30, 16) around p0 "
                       "with angle |atan2(cr, d)| = pi/2, the other side gets the line front0 = (30, 24) -> front1 = (26, 20).",
                       "flatten_arc (:490-517): radius 4, theta = 2 acos(1 - 0.25/4) = %.5f, n_lines = ceil((pi/2) / theta) = ceil(%.4f) = 3; "
                       "cos(theta) = 2 * 0.9375^2 - 1 = 0.7578125 exactly: the first interior point is (30 + 4 * 0.7578125, 20 - 4 sin(theta)) = "
                       "(33.03125, %.5f), the second (%.5f, %.5f); the third line ends at back1 exactly." % (theta, (math.pi / 2) / theta, p1[1], p2[0], p2[1]),
                       "lines are written per tag byte in order (the LineTo's two offset lines first: (20,24)->(30,24) and (30,16)->(20,16), then its "
                       "join with the next segment): the arc is lines[2..5), the other side lines[5].  Under the 2e-7 rule lines[2] and [3] would be the "
                       "degenerate pair (30,24)->(30,24), (30,16)->(30,16) that h = 1e-7 produces.",
                       "h = 1e-7: every piece is straight: LineTo 2 lines, join 2 (degenerate), the cubic (collinear, accepted whole) 2, end cap 1, "
                       "start cap 1 = 8 lines."],
        "arc_begin": [34.0, 20.0], "arc_end": [30.0, 16.0], "other_side_line": [30.0, 24.0, 26.0, 20.0],
        "arc_interior_points_approx": [[p1[0], p1[1]], [p2[0], p2[1]]], "lines_when_h_is_1e-7": 8,
        "lines_h_1e-7": [[20, 24, 30, 24], [30, 16, 20, 16], [30, 24, 30, 24], [30, 16, 30, 16], [30, 24, 50, 24], [50, 16, 30, 16],
                         [50, 24, 50, 16], [20, 16, 20, 24]]}
    # ================================= round 4 =========================================================================
    import numpy as np
    # ---- 7. the other mix modes at END_CLIP (shared/blend.wgsl:24-195), translucent source over an opaque backdrop ------
    backdrop, src_straight = [0.75, 0.5, 0.25, 1.0], [0.5, 0.875, 0.25, 0.5]
    src = [src_straight[0] * 0.5, src_straight[1] * 0.5, src_straight[2] * 0.5, 0.5]  # premultiplied by the encoder: exact
    for mix in ("screen", "overlay", "darken", "lighten", "color_dodge", "color_burn", "hard_light", "soft_light", "difference",
                "exclusion", "hue", "saturation", "color"):
        o, log = H.blend_mix_compose(backdrop, src, mix)
        hs, log = H.store_rgba16f(o, log)
        out["blend2_" + mix] = {
            "scene": "16x16 target, opaque base colour (.75, .5, .25, 1); PushLayer(%s, SrcOver, alpha 1, clip (0,0)-(16,16)); "
                     "Fill(solid straight (.5, .875, .25, .5)) of (0,0)-(16,16); PopLayer" % mix,
            "derivation": ["the fill leaves rgba = (.25, .4375, .125, .5) in the layer; END_CLIP: fg = rgba * 1 * 1; "
                           "blend_mix_compose(base, fg, %d << 8 | SrcOver): cs = fg.rgb * (1 / 0.5) = (.5, .875, .25), cb = (.75, .5, .25); "
                           "cs' = mix(cs, mixed, 1) = cs * 0 + mixed; out.rgb = base * 0.5 + cs' * 0.5, out.a = 1 (steps)" % H.MIX_NAMES.index(mix)],
            "result_premultiplied": [float(v) for v in o], "pixels_rgba16f": {"7,7": h16(hs), "0,15": h16(hs)}, "steps": log.steps}
    # ---- 8. two-point conical gradients: swapped cone with radius > 1 and with radius < 1 -------------------------------
    for name, r0 in (("radial_cone_swapped", 16.0), ("radial_cone_swapped_small", 4.0)):
        xform, focal_x, radius, kind, flags, ilog = H.rad_grad_info((4.0, 8.0), (12.0, 8.0), r0, 0.0)
        ramp_x, steps = {}, {}
        for (gx, gy) in [(6, 3), (12, 8), (1, 14), (15, 0), (9, 8), (13, 9)]:
            x, log = H.rad_grad_ramp_x(xform, focal_x, radius, kind, flags, gx, gy, 0)
            ramp_x["%d,%d" % (gx, gy)] = x
            steps["%d,%d" % (gx, gy)] = log.steps
        out[name] = {
            "scene": "16x16 target, transparent base; Fill(radial gradient c0 = (4,8) r0 = %g, c1 = (12,8) r1 = 0, red -> blue, Pad) of (0,0)-(16,16)" % r0,
            "derivation": ["draw_leaf.wgsl:151-222: r1 == 0 swaps points and radii (flags = SWAPPED): p0 = (12,8) r0 = 0, p1 = (4,8) r1 = %g; "
                           "focal_x = 0 / (0 - r1) = -0; cf = p0; radius = r1 / |cf - p1| = %g -> kind CONE (4, shared/config.wgsl:70); info[9] = flags << 3 | kind = 0xc" % (r0, r0 / 8.0),
                           "xform = scale(scale_x, scale_y) * two_point_to_unit_line(cf, p1) (steps_info), all from shared/transform.wgsl",
                           "fine.wgsl:991-1036 per pixel (steps_by_pixel): radius > 1: t = sqrt(xx + yy) - x / radius, always valid; radius < 1: "
                           "t = less_scale * sqrt(xx - yy) - x / radius with less_scale = -1 (swapped), valid iff xx >= yy and t >= 0 -- an invalid "
                           "pixel keeps its backdrop (transparent: all four halves 0); then extend(focal_x + t), 1 - t (swapped), round(t * 511)",
                           "the ramp texels are opaque: a valid pixel's stored value equals texel ramp_x of the recording's ramp upload"],
            "info_words_1_to_9": ["0x%08x" % int(np.float32(v).view(np.uint32)) for v in list(xform[0]) + list(xform[1]) + [focal_x, radius]] + ["0x%08x" % ((flags << 3) | kind)],
            "ramp_x": ramp_x, "steps_info": ilog.steps, "steps_by_pixel": steps}
    # ---- 9. sweep gradient -----------------------------------------------------------------------------------------------
    inv = H.sweep_info((8.0, 8.0))
    ramp_x, steps = {}, {}
    for (gx, gy) in [(12, 8), (12, 10), (5, 12), (3, 4), (13, 2), (8, 3), (9, 15)]:
        x, log = H.sweep_ramp_x(inv, 0.0, 1.0, gx, gy, 0)
        ramp_x["%d,%d" % (gx, gy)] = x
        steps["%d,%d" % (gx, gy)] = log.steps
    out["sweep_gradient"] = {
        "scene": "16x16 target; Fill(sweep gradient centre (8,8), angles 0 .. 2 pi, red -> blue, Pad) of (0,0)-(16,16)",
        "derivation": ["the encoder stores t0 = 0 / 2pi = 0 and t1 = f32(2 pi) / f32(2 pi) = 1 (encoding.go: angles as turn fractions); "
                       "draw_leaf.wgsl:223-235: info = inverse(translate(8,8)) = identity matrix, translate (-8,-8), then t0, t1",
                       "fine.wgsl:1038-1066: Skia's xy_to_unit_angle polynomial on slope = min(|x|,|y|) / max(|x|,|y|), octant fix-ups "
                       "(1/4 - phi, 1/2 - phi, 1 - phi), phi = (phi - t0) * 1 / (t1 - t0), Pad, round(t * 511) (steps_by_pixel)",
                       "on the +x axis (12,8): slope = 0 -> phi = 0 -> texel 0 (red); straight down (8,3) is y < 0: phi = 1 - 1/4 = 0.75 -> texel 383"],
        "info_words_1_to_8": ["0x%08x" % int(np.float32(v).view(np.uint32)) for v in list(inv[0]) + list(inv[1]) + [0.0, 1.0]],
        "ramp_x": ramp_x, "steps_by_pixel": steps}
    # ---- 10. image brush: bilinear sample of sRGB texels with linear alpha -------------------------------------------------
    px = kat_image_pixels()
    inv = H.xf_inverse(([np.float32(1), np.float32(0), np.float32(0), np.float32(1)], [np.float32(2.25), np.float32(3.5)]))
    pixels, steps = {}, {}
    for (gx, gy) in [(4, 5), (3, 4), (5, 6), (2, 3)]:
        fg, log = H.image_pixel(px, inv, gx, gy)
        rgba, log = H.over([0, 0, 0, 0], fg, 1.0, log)
        hs, log = H.store_rgba16f(rgba, log)
        pixels["%d,%d" % (gx, gy)] = h16(hs)
        steps["%d,%d" % (gx, gy)] = log.steps
    out["image_bilinear_srgb"] = {
        "scene": "16x16 target, transparent base; Fill(image brush 4x4 RGBA8, brush transform translate(2.25, 3.5)) of (0,0)-(16,16); texel (x, y) = "
                 "(60 x + 15, 50 y + 30, 255 - 40 x - 20 y, 255 except 128 where x + y is odd)",
        "derivation": ["draw_leaf.wgsl:236-247: info = inverse(translate(2.25, 3.5)): uv = (X - 2.25, Y - 3.5); extents (4, 4)",
                       "fine.wgsl:1068-1087: texels (floor, floor), (floor, ceil), (ceil, floor), (ceil, ceil) -- zero outside the texture (robust "
                       "access) --, each decoded like an rgba8unorm-srgb texel (IEC 61966-2-1 in binary64, rounded once; alpha = a / 255) and "
                       "premultiplied, then mix(mix(a, b, fy), mix(c, d, fy), fx) with mix(p, q, t) = p (1 - t) + q t",
                       "pixel (4,5): uv = (1.75, 1.5): texels (1,1), (1,2), (2,1), (2,2), fx = .75, fy = .5; pixel (2,3): uv = (-0.25, -0.5): "
                       "floor = -1 on both axes: three of the four texels are outside and read as zero",
                       "over a transparent base: rgba = fg; stored (rgb / a, a)"],
        "image_rgba8": [[int(v) for v in row.reshape(-1)] for row in px], "pixels_rgba16f": pixels, "steps_by_pixel": steps}
    # ---- 11. even-odd fill: a rectangle with a same-direction rectangle inside it ---------------------------------------------
    segs = [[2, 2, 14, 2, 1e9], [14, 2, 14, 14, 1e9], [14, 14, 2, 14, 1e9], [2, 14, 2, 2, 1e9],
            [4.5, 5, 10.5, 5, 1e9], [10.5, 5, 10.5, 11.25, 1e9], [10.5, 11.25, 4.5, 11.25, 1e9], [4.5, 11.25, 4.5, 5, 1e9]]
    fg = [0.25, 0.5, 1.0, 1.0]
    pixels, areas, steps = {}, {}, {}
    for (x, y) in [(3, 7), (4, 7), (7, 7), (10, 7), (11, 7), (7, 11), (4, 11), (7, 12), (1, 7)]:
        a, log = H.fill_area(segs, 0, x, y, even_odd=True)
        rgba, log = H.over([0, 0, 0, 0], fg, a, log)
        hs, log = H.store_rgba16f(rgba, log)
        pixels["%d,%d" % (x, y)] = h16(hs)
        areas["%d,%d" % (x, y)] = float(a)
        steps["%d,%d" % (x, y)] = log.steps
    out["even_odd_fill"] = {
        "scene": "16x16 target, transparent base; Fill(EvenOdd, solid (.25, .5, 1, 1)) of ONE path: rect (2,2)-(14,14) followed by rect "
                 "(4.5,5)-(10.5,11.25), both in the same direction",
        "derivation": ["eight lines, none touches x = 0 or a tile edge: eight segments with their own end points, y_edge = 1e9",
                       "fine.wgsl:824-870: the signed area sums to the winding number's coverage: 1 between the rectangles, 2 inside the inner "
                       "one, 1.5 / 1.25 / 1.125 where the inner rectangle's left / bottom edges cut a pixel; even-odd: "
                       "|area - 2 round(area / 2)| with round = ties to even (:865-870)",
                       "(3,7): |1 - 2 round(.5)| = |1 - 0| = 1;  (7,7): |2 - 2 round(1)| = 0;  (4,7): |1.5 - 2 round(.75)| = .5;  "
                       "(7,11): rows 11 .. 11.25 of the inner rectangle: 1.25 -> |1.25 - 2 round(.625)| = .75;  (4,11): 1 + .5 * .25 = 1.125 -> .875"],
        "areas": areas, "pixels_rgba16f": pixels, "steps_by_pixel": {k: steps[k] for k in ("4,7", "4,11")}}
    # ---- 12. five nested layers: the fifth level lives in blend_spill (fine.wgsl:938-973) ------------------------------------
    base = [0.25, 0.5, 0.75, 1.0]
    layers = [("normal", 1.0, [0.5, 0.25, 0.125, 0.5]), ("normal", 0.75, [0.125, 0.5, 0.25, 0.5]), ("screen", 0.5, [0.25, 0.125, 0.5, 0.5]),
              ("normal", 0.75, [0.5, 0.5, 0.125, 0.5]), ("multiply", 0.5, [0.125, 0.25, 0.5, 0.5])]  # (mix, layer alpha, premultiplied fill colour)
    log = H.Log()
    stack, rgba = [], [np.float32(v) for v in base]
    for k, (mix, alpha, col) in enumerate(layers):
        stack.append(rgba)
        log.steps.append("BEGIN_CLIP %d: save rgba, restart from 0; COLOR %r" % (k + 1, col))
        rgba, log = H.over([0, 0, 0, 0], col, 1.0, log)
    for k in range(4, -1, -1):
        mix, alpha, col = layers[k]
        bg = stack.pop()
        fg = [np.float32(np.float32(v * np.float32(1.0)) * np.float32(alpha)) for v in rgba]
        log.steps.append("END_CLIP %d (%s, alpha %g): fg = rgba * area * alpha = %r%s" % (k + 1, mix, alpha, [float(v) for v in fg],
                                                                                      " -- bg comes back from blend_spill" if k == 4 else ""))
        rgba, log = H.blend_mix_compose(bg, fg, mix, log)
    hs, log = H.store_rgba16f(rgba, log)
    out["five_layers_blend_spill"] = {
        "scene": "16x16 target, opaque base (.25, .5, .75, 1); five nested PushLayer(mix_k, SrcOver, alpha_k, clip (0,0)-(16,16)), each followed by "
                 "Fill(solid, premultiplied c_k) of (0,0)-(16,16); mix / alpha / c_k: " + "; ".join("%s %g %r" % l for l in layers),
        "derivation": ["fine.wgsl:938-949: BEGIN_CLIP saves rgba into blend_stack[clip_depth] for clip_depth < 4, into blend_spill behind "
                       "blend_offset for the fifth level, and restarts from 0; :951-972 END_CLIP: fg = rgba * area * alpha, rgba = "
                       "blend_mix_compose(saved, fg, blend)", "coarse: max_blend_depth = 5 > 4 -> this tile reserves (5 - 4) * 256 entries: bump.blend = 256, "
                       "ptcl[0] (blend_ix) = 0", "every step in `steps`"],
        "bump_blend": 256, "result_premultiplied": [float(v) for v in rgba], "pixels_rgba16f": {"7,7": h16(hs), "15,0": h16(hs)}, "steps": log.steps}
    return out


if __name__ == "__main__":
    json.dump(build(), open(os.path.join(HERE, "kat_pixels.json"), "w"), indent=1)
    print("wrote kat_pixels.json")
