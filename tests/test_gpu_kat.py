"""-m gpu: the HIP buffers against the hand-derived word-level known answers of tests/golden/kat_words.json -- directly,
without the oracle in between (the same words pin the oracle in tests/test_kat_words.py)."""
import numpy as np
import pytest

import jello_amd
from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS

import kat_scenes as K

pytestmark = pytest.mark.gpu
BUMP = ["failed", "binning", "ptcl", "tile", "seg_counts", "segments", "blend", "lines"]


def run_gpu(engine, scene_params):
    s, p = scene_params[:2]
    p.bump = scene_params[2] if len(scene_params) > 2 else jello_amd.BumpSizes(blend_spill=1 << 14)
    rec = jello_amd.Host().record(s, p)
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    engine.sync()
    bufs = {}

    def get(name, dt):
        if name not in bufs:
            bufs[name] = engine.download(rec.buffer(name)[0], dtype=np.uint8).copy()
        return bufs[name].view(dt)
    bump = dict(zip(BUMP, [int(v) for v in get("bumpBuf", np.uint32)[:8]]))
    return get, rec, bump


def test_nested_plain_clips(engine):
    get, rec, bump = run_gpu(engine, K.nested_plain_clips())
    try:
        assert bump["failed"] == 0
        K.check_nested_plain_clips(get, rec.config)
    finally:
        engine.release(rec)


def test_blend_layer(engine):
    get, rec, bump = run_gpu(engine, K.blend_layer())
    try:
        K.check_blend_layer(get, rec.config)
    finally:
        engine.release(rec)


def test_five_blend_layers_spill_offsets(engine):
    get, rec, bump = run_gpu(engine, K.five_blend_layers())
    try:
        K.check_five_blend_layers(get, rec.config, bump)
    finally:
        engine.release(rec)


def test_bbox_extent_rule(engine):
    get, rec, bump = run_gpu(engine, K.bbox_extent_rule())
    try:
        assert bump["failed"] == 0
        K.check_bbox_extent_rule(get, rec.config)
    finally:
        engine.release(rec)


def test_rect_on_tile_boundaries(engine):
    get, rec, bump = run_gpu(engine, K.rect_on_tile_boundaries())
    try:
        K.check_rect_on_tile_boundaries(get, rec.config, bump)
    finally:
        engine.release(rec)


def test_radial_kinds(engine):
    get, rec, bump = run_gpu(engine, K.radial_kinds())
    try:
        assert bump["failed"] == 0
        K.check_radial_kinds(get, rec.config)
    finally:
        engine.release(rec)


def test_bevel_join_between_collinear_segments(engine):
    get, rec, bump = run_gpu(engine, K.bevel_join_collinear())
    try:
        K.check_bevel(bump)
    finally:
        engine.release(rec)


def test_lines_overflow_guard(engine):
    get, rec, bump = run_gpu(engine, K.lines_overflow_guard())
    try:
        K.check_lines_overflow_guard(get, bump)
    finally:
        engine.release(rec)


# ---- pixel-level known answers (tests/golden/kat_pixels.json): the HIP image itself, not a comparison with the oracle ----
def run_gpu_image(engine, scene_params):
    get, rec, bump = run_gpu(engine, scene_params)
    t = rec.target
    return get, engine.download_image(t["id"], t["width"], t["height"]), rec, bump


def test_px_rect_fractional_edges(engine):
    get, img, rec, bump = run_gpu_image(engine, K.px_rect_fractional_edges())
    try:
        K.check_px_rect_fractional_edges(get, img, bump)
    finally:
        engine.release(rec)


def test_px_translucent_over_base(engine):
    get, img, rec, bump = run_gpu_image(engine, K.px_translucent_over_base())
    try:
        k = K.PIX["translucent_over_base"]
        assert ["0x%08x" % int(v) for v in get("ptclBuf", np.uint32)[1:8]] == k["ptcl_words_1_to_7"]
        K.check_pixels(img, k["pixels_rgba16f"])
    finally:
        engine.release(rec)


def test_px_linear_gradient_extend_modes(engine):
    get, img, rec, bump = run_gpu_image(engine, K.px_linear_gradient_extend())
    try:
        K.check_px_linear_gradient(get, img, rec)
    finally:
        engine.release(rec)


@pytest.mark.parametrize("mix", ["multiply", "luminosity"])
def test_px_end_clip_blend(engine, mix):
    get, img, rec, bump = run_gpu_image(engine, K.px_blend({"multiply": jello_amd.Mix.Multiply, "luminosity": jello_amd.Mix.Luminosity}[mix]))
    try:
        K.check_pixels(img, K.PIX["blend_" + mix]["pixels_rgba16f"])
    finally:
        engine.release(rec)


def test_px_msaa8_half_pixel(engine):
    get, img, rec, bump = run_gpu_image(engine, K.px_msaa8_half_pixel())
    try:
        K.check_pixels(img, K.PIX["msaa8_half_pixel"]["pixels_rgba16f"])
    finally:
        engine.release(rec)


def test_px_eps_tangent_rule_at_a_round_join(engine):
    get, rec, bump = run_gpu(engine, K.px_eps_tangent(1e-4))
    try:
        K.check_px_eps_tangent_join(get, bump)
    finally:
        engine.release(rec)
    get, rec, bump = run_gpu(engine, K.px_eps_tangent(1e-7))
    try:
        k = K.PIX["eps_tangent_round_join"]
        assert bump["lines"] == k["lines_when_h_is_1e-7"]
        lines = get("linesBuf", np.float32)[:8 * 6].reshape(-1, 6)[:, 2:]
        assert [[float(v) for v in r] for r in lines] == [[float(v) for v in r] for r in k["lines_h_1e-7"]]
    finally:
        engine.release(rec)


# ---- round 4: the HIP image against the hand-derived pixels of the remaining mix modes, gradients, images, even-odd, blend spill ----
@pytest.mark.parametrize("mix", K.MIX2)
def test_px_end_clip_blend_other_modes(engine, mix):
    get, img, rec, bump = run_gpu_image(engine, K.px_blend2(mix))
    try:
        K.check_pixels(img, K.PIX["blend2_" + mix]["pixels_rgba16f"])
    finally:
        engine.release(rec)


@pytest.mark.parametrize("key,r0", [("radial_cone_swapped", 16.0), ("radial_cone_swapped_small", 4.0)])
def test_px_radial_cone_swapped(engine, key, r0):
    get, img, rec, bump = run_gpu_image(engine, K.px_radial(r0))
    try:
        K.check_px_ramp_gradient(get, img, rec, key, 9)
    finally:
        engine.release(rec)


def test_px_sweep_gradient(engine):
    get, img, rec, bump = run_gpu_image(engine, K.px_sweep())
    try:
        K.check_px_ramp_gradient(get, img, rec, "sweep_gradient", 8)
    finally:
        engine.release(rec)


def test_px_image_bilinear_srgb(engine):
    """Known texel values (not "two pixels differ"), from a pixel array that is freed before the render."""
    get, img, rec, bump = run_gpu_image(engine, K.px_image())
    try:
        K.check_pixels(img, K.PIX["image_bilinear_srgb"]["pixels_rgba16f"])
    finally:
        engine.release(rec)


def test_px_even_odd_fill(engine):
    get, img, rec, bump = run_gpu_image(engine, K.px_even_odd())
    try:
        assert bump["lines"] == 8 and bump["segments"] == 8
        K.check_pixels(img, K.PIX["even_odd_fill"]["pixels_rgba16f"])
    finally:
        engine.release(rec)


def test_px_five_layers_through_the_blend_spill(engine):
    get, img, rec, bump = run_gpu_image(engine, K.px_five_layers())
    try:
        assert bump["blend"] == K.PIX["five_layers_blend_spill"]["bump_blend"]
        K.check_pixels(img, K.PIX["five_layers_blend_spill"]["pixels_rgba16f"])
    finally:
        engine.release(rec)
