"""CPU: the oracle (and the host encoder / mask LUT builder) against the hand-derived word-level known answers of
tests/golden/kat_words.json -- one tiny scene per SURVEY 2.2 divergence row that can be pinned exactly.  The same
words are asserted against the HIP buffers in tests/test_gpu_kat.py."""
import numpy as np
import pytest

import jello_amd
from oracle.oracle_engine import OracleEngine

import kat_scenes as K

BUMP = ["failed", "binning", "ptcl", "tile", "seg_counts", "segments", "blend", "lines"]


def run_oracle(scene_params):
    s, p = scene_params[:2]
    p.bump = scene_params[2] if len(scene_params) > 2 else jello_amd.BumpSizes(blend_spill=1 << 14)
    rec = jello_amd.Host().record(s, p)
    o = OracleEngine()
    o.run(rec)
    bump = dict(zip(BUMP, [int(v) for v in o.get(rec, "bumpBuf", np.uint32)[:8]]))
    return (lambda name, dt: o.get(rec, name, dt)), rec.config, bump


def test_nested_plain_clips(built):
    get, cfg, bump = run_oracle(K.nested_plain_clips())
    assert bump["failed"] == 0
    K.check_nested_plain_clips(get, cfg)


def test_blend_layer(built):
    get, cfg, bump = run_oracle(K.blend_layer())
    K.check_blend_layer(get, cfg)


def test_five_blend_layers_spill_offsets(built):
    get, cfg, bump = run_oracle(K.five_blend_layers())
    K.check_five_blend_layers(get, cfg, bump)


def test_bevel_join_between_collinear_segments(built):
    get, cfg, bump = run_oracle(K.bevel_join_collinear())
    K.check_bevel(bump)


def test_bbox_extent_rule(built):
    get, cfg, bump = run_oracle(K.bbox_extent_rule())
    assert bump["failed"] == 0
    K.check_bbox_extent_rule(get, cfg)


def test_lines_overflow_guard(built):
    get, cfg, bump = run_oracle(K.lines_overflow_guard())
    K.check_lines_overflow_guard(get, bump)


def test_rect_on_tile_boundaries(built):
    get, cfg, bump = run_oracle(K.rect_on_tile_boundaries())
    K.check_rect_on_tile_boundaries(get, cfg, bump)


def test_radial_kinds(built):
    get, cfg, bump = run_oracle(K.radial_kinds())
    assert bump["failed"] == 0
    K.check_radial_kinds(get, cfg)


def test_gradient_in_clip_encoder_streams(built):
    k = K.KAT["gradient_in_clip_streams"]
    s, p = K.gradient_in_clip()
    assert s.stream("path_tags").hex(" ") == k["path_tags_hex"]
    assert list(np.frombuffer(s.stream("path_data"), np.float32)) == k["path_data_f32"]
    assert list(np.frombuffer(s.stream("draw_tags"), np.uint32)) == K.words(k["draw_tags"])
    assert list(np.frombuffer(s.stream("draw_data"), np.uint32)) == K.words(k["draw_data_u32"])
    assert list(np.frombuffer(s.stream("transforms"), np.float32)) == k["transforms_f32"]
    assert list(np.frombuffer(s.stream("styles"), np.uint32)) == K.words(k["styles_u32"])
    c = s.counts()
    for name, v in k["counts"].items():
        assert c[name] == v, name


def test_f32_bit_patterns_of_the_fixture():
    for val, bits in K.KAT["f32"].items():
        assert int(np.float32(float(val)).view(np.uint32)) == int(bits, 16)


def _mask_lut8_from_the_definition():
    """renderer/mask.go:43-61 restated independently (float64, vectorised)."""
    pattern = np.array([0, 5, 3, 7, 1, 4, 6, 2], np.float64)
    out = np.zeros(32 * 32, np.uint8)
    for i in range(32 * 32):
        u, v = i % 32, i // 32
        is_pos = v >= 16
        slope = (v % 16 + 0.5) / 16.0
        t = (u + 0.5) / 32.0
        if is_pos:
            t = 1.0 - t
        k = np.arange(8, dtype=np.float64)
        y = (k + 0.5) * 0.125
        x = (pattern + 0.5) * 0.125
        if not is_pos:
            y = 1.0 - y
        bits = ((x - (1.0 - t)) * (1.0 - slope) - (y - t) * slope) >= 0.0
        out[i] = int(sum(1 << int(j) for j in np.flatnonzero(bits)))
    return out


def _mask_lut16_from_the_definition():
    pattern = np.array([1, 8, 4, 11, 15, 7, 3, 12, 0, 9, 5, 13, 2, 10, 6, 14], np.float64)
    out = np.zeros(64 * 64, np.uint16)
    for i in range(64 * 64):
        u, v = i % 64, i // 64
        is_pos = v >= 32
        slope = (v % 32 + 0.5) / 32.0
        t = (u + 0.5) / 64.0
        if is_pos:
            t = 1.0 - t
        k = np.arange(16, dtype=np.float64)
        y = (k + 0.5) * 0.0625
        x = (pattern + 0.5) * 0.0625
        if not is_pos:
            y = 1.0 - y
        bits = ((x - (1.0 - t)) * (1.0 - slope) - (y - t) * slope) >= 0.0
        out[i] = int(sum(1 << int(j) for j in np.flatnonzero(bits)))
    return out


def test_mask_lut_matches_the_definition_and_hand_computed_entries(built):
    """The MSAA mask LUT the renderer uploads (host/renderer.cpp make_mask_lut8/16) against renderer/mask.go restated
    here, plus three entries computed by hand in the fixture."""
    s, p = K.bevel_join_collinear()
    luts = {}
    for aa in (jello_amd.Aa.Msaa8, jello_amd.Aa.Msaa16):
        p.aa = aa
        rec = jello_amd.Host().record(s, p)
        for c in rec.commands():
            if c["kind"] == jello_amd.CMD.UPLOAD and c["buf_name"].lower().startswith("mask"):
                luts[aa] = np.frombuffer(c["data"], np.uint8)
    lut8, lut16 = luts[jello_amd.Aa.Msaa8], luts[jello_amd.Aa.Msaa16].view(np.uint16)
    assert np.array_equal(lut8[:1024], _mask_lut8_from_the_definition())
    assert np.array_equal(lut16[:4096], _mask_lut16_from_the_definition())
    for ix, v in K.KAT["mask_lut"]["lut8"].items():
        assert int(lut8[int(ix)]) == int(v, 16), ix
