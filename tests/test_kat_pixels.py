"""CPU: the pixel-level known answers of tests/golden/kat_pixels.json.  (1) The fixture is re-derived by hand
(tests/fine_by_hand.py: one binary32 operation per line from the WGSL text) and must equal the committed file; (2) the
oracle's image, segments, info words and lines are held to it.  tests/test_gpu_kat.py holds the HIP buffers to the same file."""
import importlib.util
import json
import os

import numpy as np
import pytest

import jello_amd
from oracle.oracle_engine import OracleEngine

import kat_scenes as K

BUMP = ["failed", "binning", "ptcl", "tile", "seg_counts", "segments", "blend", "lines"]
HERE = os.path.dirname(os.path.abspath(__file__))


def run_oracle(scene_params):
    s, p = scene_params
    p.bump = jello_amd.BumpSizes(blend_spill=1 << 14)
    rec = jello_amd.Host().record(s, p)
    o = OracleEngine()
    o.run(rec)
    bump = dict(zip(BUMP, [int(v) for v in o.get(rec, "bumpBuf", np.uint32)[:8]]))
    return (lambda name, dt: o.get(rec, name, dt)), o.target(rec), rec, bump


def test_fixture_is_what_the_hand_derivation_gives():
    spec = importlib.util.spec_from_file_location("make_kat_pixels", os.path.join(HERE, "golden", "make_kat_pixels.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert json.loads(json.dumps(m.build())) == K.PIX


def test_hand_derived_areas_are_the_covered_fractions():
    """The by-hand evaluation of fill_path itself against plain geometry: these areas are exact in binary32."""
    a = K.PIX["rect_fractional_edges"]["areas"]
    assert a["5,5"] == 1.0 and a["2,5"] == 0.5 and a["9,5"] == 0.5 and a["5,3"] == 0.75 and a["5,7"] == 0.75
    assert a["2,3"] == 0.375 and a["9,7"] == 0.375 and a["1,5"] == 0.0 and a["10,5"] == 0.0 and a["5,8"] == 0.0


def test_rect_fractional_edges(built):
    get, img, rec, bump = run_oracle(K.px_rect_fractional_edges())
    K.check_px_rect_fractional_edges(get, img, bump)


def test_translucent_over_base(built):
    get, img, rec, bump = run_oracle(K.px_translucent_over_base())
    k = K.PIX["translucent_over_base"]
    assert ["0x%08x" % int(v) for v in get("ptclBuf", np.uint32)[1:8]] == k["ptcl_words_1_to_7"]
    assert [float(v) for v in rec.config["base_color"]] == [0.25, 0.125, 0.0625, 0.5]
    K.check_pixels(img, k["pixels_rgba16f"])


def test_linear_gradient_extend_modes(built):
    get, img, rec, bump = run_oracle(K.px_linear_gradient_extend())
    K.check_px_linear_gradient(get, img, rec)


@pytest.mark.parametrize("mix", ["multiply", "luminosity"])
def test_end_clip_blend(built, mix):
    get, img, rec, bump = run_oracle(K.px_blend({"multiply": jello_amd.Mix.Multiply, "luminosity": jello_amd.Mix.Luminosity}[mix]))
    K.check_pixels(img, K.PIX["blend_" + mix]["pixels_rgba16f"])


def test_msaa8_half_pixel(built):
    get, img, rec, bump = run_oracle(K.px_msaa8_half_pixel())
    K.check_pixels(img, K.PIX["msaa8_half_pixel"]["pixels_rgba16f"])


def test_eps_tangent_rule_at_a_round_join(built):
    get, img, rec, bump = run_oracle(K.px_eps_tangent(1e-4))
    K.check_px_eps_tangent_join(get, bump)
    get, img, rec, bump = run_oracle(K.px_eps_tangent(1e-7))
    k = K.PIX["eps_tangent_round_join"]
    assert bump["lines"] == k["lines_when_h_is_1e-7"]
    lines = get("linesBuf", np.float32)[:8 * 6].reshape(-1, 6)[:, 2:]
    assert [[float(v) for v in r] for r in lines] == [[float(v) for v in r] for r in k["lines_h_1e-7"]]


# ---- round 4: the remaining mix modes, conical / sweep gradients, an sRGB image brush, even-odd, the blend spill ----
@pytest.mark.parametrize("mix", K.MIX2)
def test_end_clip_blend_other_modes(built, mix):
    get, img, rec, bump = run_oracle(K.px_blend2(mix))
    K.check_pixels(img, K.PIX["blend2_" + mix]["pixels_rgba16f"])


@pytest.mark.parametrize("key,r0", [("radial_cone_swapped", 16.0), ("radial_cone_swapped_small", 4.0)])
def test_radial_cone_swapped(built, key, r0):
    get, img, rec, bump = run_oracle(K.px_radial(r0))
    K.check_px_ramp_gradient(get, img, rec, key, 9)


def test_sweep_gradient(built):
    get, img, rec, bump = run_oracle(K.px_sweep())
    K.check_px_ramp_gradient(get, img, rec, "sweep_gradient", 8)


def test_image_bilinear_srgb(built):
    get, img, rec, bump = run_oracle(K.px_image())
    K.check_pixels(img, K.PIX["image_bilinear_srgb"]["pixels_rgba16f"])


def test_even_odd_fill(built):
    get, img, rec, bump = run_oracle(K.px_even_odd())
    assert bump["lines"] == 8 and bump["segments"] == 8
    K.check_pixels(img, K.PIX["even_odd_fill"]["pixels_rgba16f"])


def test_five_layers_through_the_blend_spill(built):
    get, img, rec, bump = run_oracle(K.px_five_layers())
    assert bump["blend"] == K.PIX["five_layers_blend_spill"]["bump_blend"]
    K.check_pixels(img, K.PIX["five_layers_blend_spill"]["pixels_rgba16f"])


def test_hand_derived_even_odd_areas():
    a = K.PIX["even_odd_fill"]["areas"]
    assert a == {"3,7": 1.0, "4,7": 0.5, "7,7": 0.0, "10,7": 0.5, "11,7": 1.0, "7,11": 0.75, "4,11": 0.875, "7,12": 1.0, "1,7": 0.0}
