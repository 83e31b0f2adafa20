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
    return out


if __name__ == "__main__":
    json.dump(build(), open(os.path.join(HERE, "kat_pixels.json"), "w"), indent=1)
    print("wrote kat_pixels.json")
