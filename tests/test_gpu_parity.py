"""-m gpu: the HIP pipeline against the CPU oracle, through the C ABI, on the BASELINE.json configs
(C1 exact, C2/C3/C4 at sizes the oracle finishes in seconds).  Bar: bit-exact for every integer /
index buffer and every f32 bit pattern (lines, segments, PTCL), image identical in f16 bits."""
import numpy as np
import pytest

import jello_amd
from jello_amd import BumpSizes, scenes

from parity import compare

pytestmark = pytest.mark.gpu


def test_c1_rect_and_stroked_cubic(engine):
    s, p = scenes.scene_c1()
    r = compare(engine, s, p)
    assert r["bump"]["lines"] == 70
    img = r["image"].view(np.float16).astype(np.float32)
    assert tuple(img[50, 50]) == (1.0, 0.0, 0.0, 1.0)


@pytest.mark.parametrize("n,size", [(500, 256), (3000, 512), (20000, 1024)])
def test_c3_random_cubics(engine, n, size):
    s, p = scenes.scene_c3(n, size)
    p.bump = BumpSizes(lines=1 << 21, seg_counts=1 << 21, segments=1 << 21, tiles=1 << 21, ptcl=1 << 24, bin_data=1 << 20)
    r = compare(engine, s, p)
    assert r["bump"]["lines"] > n


def test_c3_headline_full_size(engine):
    """BASELINE.json configs[2] at full size: 100k stroked+filled cubics, 4096x4096 -- every buffer bit-exact."""
    s, p = scenes.scene_c3(100_000, 4096)
    p.bump = BumpSizes(lines=1 << 22, seg_counts=1 << 23, segments=1 << 23, tiles=1 << 21, ptcl=1 << 25, bin_data=1 << 20)
    r = compare(engine, s, p)
    assert r["bump"]["lines"] > 3_000_000 and r["bump"]["failed"] == 0


def test_c2_blobs_all_joins_caps_evenodd(engine):
    s, p = scenes.scene_c2(300, 1024)
    compare(engine, s, p)


def test_c4_clips_gradients_blends(engine):
    s, p = scenes.scene_c4(1500, 512)
    p.bump = BumpSizes(ptcl=1 << 24)
    compare(engine, s, p)


def test_images_srgb_bilinear(engine):
    s, p = scenes.scene_images()
    r = compare(engine, s, p)
    img = r["image"].view(np.float16).astype(np.float32)
    assert not np.array_equal(img[40, 30], img[41, 31])  # the image brush really painted texels


@pytest.mark.parametrize("aa", [jello_amd.Aa.Msaa8, jello_amd.Aa.Msaa16])
@pytest.mark.parametrize("which", ["c1", "c2", "c3", "c4"])
def test_msaa_matches_oracle(engine, aa, which):
    """fine_msaa8 / fine_msaa16 (fine.wgsl:148-711): integer SWAR coverage, image identical in f16 bits."""
    if which == "c1":
        s, p = scenes.scene_c1()
    elif which == "c2":
        s, p = scenes.scene_c2(120, 512)      # even-odd fills, all joins and caps
    elif which == "c3":
        s, p = scenes.scene_c3(2000, 512)
        p.bump = BumpSizes(ptcl=1 << 24)
    else:
        s, p = scenes.scene_c4(600, 256)      # clip layers, gradients, blends
        p.bump = BumpSizes(ptcl=1 << 24)
    p.aa = aa
    compare(engine, s, p)


@pytest.mark.parametrize("aa", [jello_amd.Aa.Area, jello_amd.Aa.Msaa8])
def test_large_shapes(engine, aa):
    """A full-target background, circles and bars of hundreds of tiles, a large clip layer: the device-wide tile
    clear, the wave-per-row backdrop route, bin lists with an element in every bin, the list route of path_count."""
    s, p = scenes.scene_large_shapes()
    p.bump = BumpSizes(ptcl=1 << 24)
    p.aa = aa
    r = compare(engine, s, p)
    assert r["bump"]["tile"] > 90000


@pytest.mark.parametrize("aa", [jello_amd.Aa.Area, jello_amd.Aa.Msaa8, jello_amd.Aa.Msaa16])
def test_dense_polygon_and_long_lines(engine, aa):
    """40 k radial zigzag edges of one path (more than 1024 crossings in a single tile: the wave-per-tile ranking of
    path_count in several LDS passes) and edge-to-edge strokes (lines handed to a whole wave in k_pc_emit).  With the
    multisampled modes: fills of hundreds of segments per tile, i.e. fills that span many batches of touched pixels (ms_build)."""
    s, p = scenes.scene_dense_polygon()
    p.bump = BumpSizes(lines=1 << 17, seg_counts=1 << 20, segments=1 << 20, ptcl=1 << 22)
    p.aa = aa
    r = compare(engine, s, p)
    assert r["bump"]["seg_counts"] > 90000
    assert r["max_tile_segments"] > 1024


def test_lines_without_extent(engine):
    """Segments whose lines collapse to a point (flatten.wgsl:893-899 merges a segment's box only if it has an extent):
    paths that are all points, points next to real segments, and strokes squeezed onto a vertical line, whose point-like
    lines belong to segments that do have an extent."""
    import kat_scenes as K
    from jello_amd import Brush, Cap, Fill, Join, Path, Stroke
    s, p = K.bbox_extent_rule()
    for i, (join, cap) in enumerate([(Join.Round, Cap.Round), (Join.Miter, Cap.Square), (Join.Bevel, Cap.Butt)]):
        path = Path().move_to(2, 5 + i).line_to(9, 5 + i).line_to(9, 30).cubic_to(20, 40, 0, 50, 7, 60 - i)
        s.stroke(Stroke(3.0 + i, join, 4.0, cap, cap), (0, 0, 0, 1, 30 + i, 0), Brush.solid((0.2, 0.3, 0.4, 1.0)), None, path)
        s.fill(Fill.EvenOdd, (0, 0, 0.5, 0, 10 * i, 7), Brush.solid((0.2, 0.3, 0.4, 1.0)), None, path)
    compare(engine, s, p)


def test_big_path_takes_the_list_route(engine):
    """A single path with far more tile crossings than PC_BIG_PATH next to small ones: both rank routes of path_count."""
    s, p = scenes.scene_big_path()
    p.bump = BumpSizes(ptcl=1 << 24)
    r = compare(engine, s, p)
    assert r["bump"]["seg_counts"] > 60000


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_mixed_scenes(engine, seed):
    """Random mixtures of every feature (scenes.scene_fuzz): all buffers and the image against the oracle; every third
    seed with multisampling."""
    s, p = scenes.scene_fuzz(seed)
    p.bump = BumpSizes(ptcl=1 << 22)
    if seed % 3 == 1:
        p.aa = jello_amd.Aa.Msaa8
    elif seed % 3 == 2:
        p.aa = jello_amd.Aa.Msaa16
    compare(engine, s, p)


@pytest.mark.parametrize("seed", [1, 7, 8, 13, 20, 24])
def test_fuzz_extreme_scenes(engine, seed):
    """scene_fuzz(extreme=True): coordinates far outside the target, scales 0.02..40, stroke widths up to 300, up to nine
    nested layers (seeds 1 and 7 reach the blend spill buffer of fine)."""
    s, p = scenes.scene_fuzz(seed, extreme=True)
    p.bump = BumpSizes(ptcl=1 << 22, blend_spill=1 << 18)
    p.aa = [jello_amd.Aa.Area, jello_amd.Aa.Msaa8, jello_amd.Aa.Msaa16][seed % 3]
    r = compare(engine, s, p)
    if seed in (1, 7):
        assert r["bump"]["blend"] > 0


@pytest.mark.parametrize("which,world", [("c3", 3), ("c4", 2), ("fuzz", 4)])
def test_band_mode_reassembles_the_unsharded_frame(engine, which, world):
    """SURVEY 8e, second way: every "rank" runs the element stages on the whole scene and coarse(write)+fine only for its
    band of bin rows (Engine.set_band); bump allocators and the PTCL words of the band are those of the unsharded run
    and the bands stitched together are the unsharded image, bit for bit.  (The ranks run one after the other here.)"""
    from jello_amd import sharding
    from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS
    if which == "c3":
        s, p = scenes.scene_c3(3000, 1024)
    elif which == "c4":
        s, p = scenes.scene_c4(600, 768)
    else:
        s, p = scenes.scene_fuzz(5, size=1100, n=120)
    p.bump = BumpSizes(ptcl=1 << 23, blend_spill=1 << 20)
    host = jello_amd.Host()
    rec = host.record(s, p)
    t = rec.target
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    engine.sync()
    full = engine.download_image(t["id"], t["width"], t["height"]).copy()
    full_bump = engine.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8].copy()
    full_ptcl = engine.download(rec.buffer("ptclBuf")[0], dtype=np.uint32).copy()
    assert full_bump[0] == 0
    from parity import ptcl_walk
    hb = (rec.config["height_in_tiles"] + 15) // 16
    live_full = ptcl_walk(full_ptcl, rec.config)
    live_union = np.zeros_like(live_full)
    stitched = np.zeros_like(full)
    try:
        for rank in range(world):
            y0, y1 = sharding.band_for_rank(hb, world, rank)
            engine.set_band(y0, y1)
            engine.clear(rec.buffer("ptclBuf")[0])  # whatever is live afterwards was written by this run
            engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
            engine.sync()
            img = engine.download_image(t["id"], t["width"], t["height"])
            stitched[y0 * 256:y1 * 256] = img[y0 * 256:y1 * 256]
            assert np.array_equal(engine.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8], full_bump)
            ptcl = engine.download(rec.buffer("ptclBuf")[0], dtype=np.uint32)
            live = ptcl_walk(ptcl, rec.config)  # tiles outside the band: an all-zero head, i.e. CMD_END at once
            wt, ht = rec.config["width_in_tiles"], rec.config["height_in_tiles"]
            live[:y0 * 16 * wt * 64] = False                       # ... whose two head words are not this band's business
            live[min(y1 * 16, ht) * wt * 64:wt * ht * 64] = False
            assert live.any() and not (live & ~live_full).any()
            assert np.array_equal(ptcl[live], full_ptcl[live])
            live_union |= live
    finally:
        engine.set_band()
    assert np.array_equal(live_union & live_full, live_full)  # every live word of the unsharded PTCL was written by some band
    assert np.array_equal(stitched.view(np.uint16), full.view(np.uint16))
    engine.release(rec)


def test_non_multiple_of_16_target(engine):
    s, p = scenes.scene_c3(400, 256)
    p.width, p.height = 250, 199
    compare(engine, s, p)


def test_empty_scene(engine):
    s = jello_amd.Scene()
    p = jello_amd.RenderParams(64, 64, base_color=(0.5, 0.25, 1.0, 1.0))
    host = jello_amd.Host()
    rec = host.record(s, p)
    engine.run(rec)
    t = rec.target
    # target image was freed with the frame in a full run; re-run retaining it
    from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    img = engine.download_image(t["id"], t["width"], t["height"]).view(np.float16).astype(np.float32)
    engine.release(rec)
    assert np.allclose(img[..., 0], 0.5) and np.allclose(img[..., 3], 1.0)


def test_empty_scene_after_a_failed_frame_sees_a_cleared_bump(engine):
    """the recording's bbox_clear and Clear(bump) are held back for flatten to absorb; an empty scene dispatches
    flatten with zero workgroups, which launches nothing -- the clear must then run as recorded.  The pooled 32-byte bump
    allocation is poisoned by a frame that fails (failed != 0, counters != 0) and is handed to the empty frame next."""
    s, p = scenes.scene_c3(400, 256)
    p.bump = BumpSizes(bin_data=256, tiles=512, lines=64, seg_counts=64, segments=64, blend_spill=256, ptcl=1 << 14)
    rec = jello_amd.Host().record(s, p)
    from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    engine.sync()
    dirty = engine.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8].copy()
    assert dirty[0] != 0 and dirty[7] != 0, dirty   # failed, lines
    engine.release(rec)                              # the 32-byte allocation goes back to the pool as it is
    for _ in range(2):
        e = jello_amd.Scene()
        rec, bump, attempts = engine.render(e, jello_amd.RenderParams(64, 64, base_color=(0.5, 0.25, 1.0, 1.0)), robust=True, retain=True)
        assert attempts == 1 and all(v == 0 for v in bump.values()), bump
        t = rec.target
        img = engine.download_image(t["id"], t["width"], t["height"]).view(np.float16).astype(np.float32)
        engine.release(rec)
        assert np.all(img[..., 0] == 0.5) and np.all(img[..., 1] == 0.25) and np.all(img[..., 3] == 1.0)


def test_regrow_loop_recovers_from_undersized_buffers(engine):
    """renderer/render.go:458-460 reads bump back; with every bump buffer far too small the first attempt must fail
    cleanly (no hang, no fault), the regrow loop must converge, and the final image must be the one the oracle gives
    with comfortable buffers."""
    s, p = scenes.scene_c3(800, 256)
    p.bump = BumpSizes(bin_data=256, tiles=512, lines=1024, seg_counts=1024, segments=1024, blend_spill=256, ptcl=1 << 14)
    rec, bump, attempts = engine.render(s, p, robust=True, retain=True)
    assert bump["failed"] == 0 and attempts > 1
    t = rec.target
    got = engine.download_image(t["id"], t["width"], t["height"])
    engine.release(rec)
    s2, p2 = scenes.scene_c3(800, 256)
    p2.bump = BumpSizes(ptcl=1 << 22)
    r = compare(engine, s2, p2)
    assert np.array_equal(got.view(np.uint16), r["image"].view(np.uint16))


def test_undersized_buffers_fail_cleanly_without_regrow(engine):
    s, p = scenes.scene_c3(800, 256)
    p.bump = BumpSizes(bin_data=256, tiles=512, lines=1024, seg_counts=1024, segments=1024, blend_spill=256, ptcl=1 << 14)
    # (a non-robust recording has no Download(bumpBuf), render.go:458-460: the allocators are read back by hand.  Until round 4
    # this test looked at the bump array engine.render returns for such a frame -- uninitialised memory that happened to be
    # non-zero; hip_engine::Frame::bump is zero-initialised now)
    from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS
    rec = jello_amd.Host().record(s, p)
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    engine.sync()
    bump = engine.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8].copy()
    engine.release(rec)
    assert bump[0] != 0, bump
    rec2, bump2, attempts = engine.render(s, p, robust=False, retain=False)
    assert attempts == 1 and all(v == 0 for v in bump2.values())  # nothing was downloaded: nothing is reported


@pytest.mark.parametrize("n,size", [(1500, 512), (6000, 1024)])
def test_c4_nested_clips_with_visible_paints(engine, n, size):
    """scene_c4_nested: the clip circles of a group are concentric and its paths lie inside them, so gradients and all 16
    mix modes are composited for real (scene_c4's independent circles clip nearly every paint away)."""
    s, p = scenes.scene_c4_nested(n, size)
    p.bump = s.bump_sizes(p.width, p.height)
    r = compare(engine, s, p)
    img = r["image"].view(np.float16).astype(np.float32)
    assert (img[:, :, :3] != 0).any(axis=2).mean() > 0.5  # most of the target is painted


def test_c4_nested_full_size(engine):
    """The nested variant at configs[3]'s size: 30 k paths, 9000 clip layers, 2048x2048, against the oracle (16 threads)."""
    from oracle import oracle_engine
    s, p = scenes.scene_c4_nested(30_000, 2048)
    p.bump = s.bump_sizes(p.width, p.height)
    oracle_engine.lib().oracle_set_threads(16)
    try:
        r = compare(engine, s, p)
    finally:
        oracle_engine.lib().oracle_set_threads(1)
    assert r["bump"]["failed"] == 0


def test_c4_full_size(engine):
    """BASELINE.json configs[3] at its stated size: 30 k paths, 9000 clip layers (all 16 mix modes), radial gradients,
    2048x2048 -- sized by the estimator, every buffer and the image bit-exact against the oracle (16 host threads)."""
    from oracle import oracle_engine
    s, p = scenes.scene_c4(30_000, 2048)
    p.bump = s.bump_sizes(p.width, p.height)
    oracle_engine.lib().oracle_set_threads(16)
    try:
        r = compare(engine, s, p)
    finally:
        oracle_engine.lib().oracle_set_threads(1)
    assert r["bump"]["failed"] == 0 and r["ptcl_live_words"] > 10_000_000


def test_poisoned_scratch_and_growing_scenes(engine):
    """The deterministic allocators keep counters in per-context scratch that a frame leaves zeroed for the next one
    ("self-cleaning", kcommon.h JH_CLEAN_*).  Nothing may depend on what else that memory held: poison it the way a fresh
    non-zero allocation would look, render a small scene (few workgroups touch few counters), then a larger one on the
    same, not regrown allocations -- every counter of flatten's block (the list counters, the region cursors of its temporary)
    must have been zeroed all the same, whatever the small frame touched -- and a clip scene after a second
    poisoning."""
    def sized(sp):
        s, p = sp
        p.bump = BumpSizes(lines=1 << 21, seg_counts=1 << 21, segments=1 << 21, tiles=1 << 21, ptcl=1 << 24, bin_data=1 << 20)
        return s, p
    big, small = sized(scenes.scene_c3(20000, 1024)), sized(scenes.scene_c3(300, 256))
    compare(engine, *big)  # grows every scratch slot to the size the big scene needs
    engine.debug_poison_scratch(0xA5)
    compare(engine, *small)
    r = compare(engine, *big)
    assert r["bump"]["lines"] > 20000
    engine.debug_poison_scratch(0xFF)
    s, p = scenes.scene_c4(1500, 512)
    compare(engine, s, p)
    compare(engine, *small)


@pytest.mark.parametrize("aa", [jello_amd.Aa.Area, jello_amd.Aa.Msaa8])
@pytest.mark.parametrize("kind", ["deep", "siblings", "comb", "mixed"])
def test_clip_torture(engine, kind, aa):
    """Clip-layer patterns far beyond the C4 recipes (scenes.scene_clip_torture): a nest 120 deep, 1500 sibling layers over the same tiles
    (covering, missing and cutting them), layers that stay open across many batches of 256 draw objects, random interleavings with up to
    60 open layers -- the per-tile state machine of coarse (lanes = the elements of a tile: skipped stretches, partners per nesting level,
    chunk boundaries inside a wave step, more than 64 elements of one batch on one tile) and fine's lazy layers / blend spill."""
    s, p = scenes.scene_clip_torture(kind)
    p.bump = BumpSizes(ptcl=1 << 24, blend_spill=1 << 23)
    p.aa = aa
    r = compare(engine, s, p)
    assert r["bump"]["failed"] == 0
