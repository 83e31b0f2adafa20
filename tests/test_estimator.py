"""CPU: the BumpEstimator (renderer/estimate.go restated in jello_amd/host/estimate.cpp) and the footprint bound for
tiles / bin data / PTCL size the bump buffers so that the FIRST attempt fits -- checked against what the oracle really
allocates on every scene family -- without being absurdly large."""
import numpy as np
import pytest

import jello_amd
from jello_amd import scenes
from oracle.oracle_engine import OracleEngine

BUMP = ["failed", "binning", "ptcl", "tile", "seg_counts", "segments", "blend", "lines"]


def _scenes():
    yield "c1", scenes.scene_c1()
    yield "c2", scenes.scene_c2(120, 512)
    yield "c3", scenes.scene_c3(1500, 512)
    yield "c3_dense", scenes.scene_c3(3000, 256)
    yield "c4", scenes.scene_c4(400, 384)
    yield "images", scenes.scene_images()
    yield "large_shapes", scenes.scene_large_shapes(size=512, n=30)
    for seed in (0, 3, 5):
        yield "fuzz%d" % seed, scenes.scene_fuzz(seed)
    for seed in (1, 7):
        yield "fuzz_extreme%d" % seed, scenes.scene_fuzz(seed, extreme=True)


@pytest.mark.parametrize("name,sp", list(_scenes()), ids=[n for n, _ in _scenes()])
def test_estimated_sizes_fit_on_the_first_attempt(built, name, sp):
    s, p = sp
    p.bump = s.bump_sizes(p.width, p.height)
    rec = jello_amd.Host().record(s, p)
    o = OracleEngine()
    o.run(rec)
    bump = dict(zip(BUMP, [int(v) for v in o.get(rec, "bumpBuf", np.uint32)[:8]]))
    assert bump["failed"] == 0, (name, bump, p.bump.as_dict())
    cfg = rec.config
    used = {"lines": bump["lines"], "seg_counts": bump["seg_counts"], "segments": bump["segments"], "tiles": bump["tile"],
            "bin_data": bump["binning"] + cfg["bin_data_start"], "blend_spill": bump["blend"],
            "ptcl": bump["ptcl"] + cfg["width_in_tiles"] * cfg["height_in_tiles"] * 64}
    for k, v in used.items():
        have = getattr(p.bump, k)
        assert have >= v, (name, k, have, v)


def test_c3_estimate_is_within_a_small_factor(built):
    """The headline generator: the estimate must not be a wild overshoot either (memory = what gets allocated)."""
    s, p = scenes.scene_c3(4000, 1024)
    est = s.bump_sizes(p.width, p.height)
    p.bump = est
    rec = jello_amd.Host().record(s, p)
    o = OracleEngine()
    o.run(rec)
    bump = dict(zip(BUMP, [int(v) for v in o.get(rec, "bumpBuf", np.uint32)[:8]]))
    assert bump["failed"] == 0
    assert est.lines < 6 * bump["lines"] and est.segments < 6 * bump["segments"] and est.tiles < 3 * bump["tile"]


def test_raw_tally_follows_the_reference_formulas(built):
    """estimate.go by hand: Fill of MoveTo(0,0) LineTo(100,0) LineTo(100,50) (identity): linetos = 2 + 1 close line = 3,
    no curves -> lines 3; segments: LineTo 1: ceil(100/16)=7 (+0 for dy -> ceil(0)=0) = 7; LineTo 2 measured from the
    FIRST point as the reference does (estimate.go:108: firstPt..lastPt): dx 100 -> 7, dy 50 -> 4 = 11; implicit close
    first..last = 11 -> 29."""
    s = jello_amd.Scene()
    path = jello_amd.Path().move_to(0, 0).line_to(100, 0).line_to(100, 50)
    s.fill(jello_amd.Fill.NonZero, None, jello_amd.Brush.solid((1, 0, 0, 1)), None, path)
    e = s.bump_estimate()
    assert e["lines"] == 3
    assert e["segments"] == 29 and e["seg_counts"] == 29 and e["binning"] == 29
    # Append under a uniform scale of 2: transform_scale = |(4,0)| + |(0,0)| = 4 -> segments x4, explicit lines unchanged
    s2 = jello_amd.Scene()
    s2.append(s, (2, 0, 0, 2, 0, 0))
    e2 = s2.bump_estimate()
    assert e2["lines"] == 3 and e2["segments"] == 29 * 4


def test_held_back_sizes_are_reported(built):
    """The first attempt is capped at 16 x the reference's constants (scene.cpp): a scene whose bounds ask for more must say so,
    so that a caller without the regrow loop (graph capture, a timed loop) knows to render robustly once."""
    from jello_amd import Brush, Fill, Path, Scene
    s, p = scenes.scene_c3(4000, 1024)
    assert s.bump_sizes_clamped(p.width, p.height) == []
    heavy = Scene()
    for i in range(3000):  # 3000 rectangles over a whole 4096^2 target: 3000 x 65536 tiles by the bounding-box bound
        heavy.fill(Fill.NonZero, None, Brush.solid((0.1, 0.2, 0.3, 0.5)), None, Path.rect(0, 0, 4096, 4096))
    held = heavy.bump_sizes_clamped(4096, 4096)
    assert "tiles" in held and "ptcl" in held, held
    sizes = heavy.bump_sizes(4096, 4096)
    assert sizes.tiles == 16 * (1 << 21)
    assert "lines" not in held
