"""-m gpu: the pieces of the drop-in boundary beyond plain dispatch: estimator-sized first attempt,
WriteImage, image arrays larger than the inline descriptor set, nested profile groups, stale-graph detection, and the
size checks that keep a too-small buffer from being read out of bounds."""
import ctypes

import numpy as np
import pytest

import jello_amd
from jello_amd import BumpSizes, scenes
from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS

from parity import compare

pytestmark = pytest.mark.gpu


class Binding(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("count", ctypes.c_uint32), ("id", ctypes.c_uint64), ("ids", ctypes.POINTER(ctypes.c_uint64))]


def test_c3_headline_is_sized_by_the_estimator_in_one_attempt(engine):
    """renderer/estimate.go feeds the buffer sizes; the regrow loop must not be needed for C3."""
    s, p = scenes.scene_c3(100_000, 4096)
    p.bump = s.bump_sizes(p.width, p.height)
    rec, bump, attempts = engine.render(s, p, robust=True, retain=False)
    assert bump["failed"] == 0 and attempts == 1
    assert bump["lines"] > 3_000_000 and p.bump.lines < 4 * bump["lines"]


def test_c4_full_size_is_sized_by_the_estimator_in_one_attempt(engine):
    s, p = scenes.scene_c4(30_000, 2048)
    p.bump = s.bump_sizes(p.width, p.height)
    rec, bump, attempts = engine.render(s, p, robust=True, retain=False)
    assert bump["failed"] == 0 and attempts == 1


def test_write_image_updates_a_sub_rectangle(engine):
    """WriteImage (recording.go:204-208, wgpu.go:422-452): rows of the rectangle land at (x, y), nothing else changes."""
    hip, ctx = engine.hip, engine.ctx
    iid = 0x77110001
    w, h = 37, 21
    base = (np.arange(w * h * 4, dtype=np.uint32) % 251).astype(np.uint8).reshape(h, w, 4)
    hip.jh_image_upload.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64]
    hip.jh_image_free.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    assert hip.jh_image_upload(ctx, iid, w, h, 0, base.ctypes.data, base.nbytes) == 0
    patch = np.full((5, 9, 4), 200, np.uint8)
    patch[..., 1] = np.arange(9, dtype=np.uint8)[None, :]
    assert hip.jh_image_write(ctx, iid, 11, 3, 9, 5, patch.ctypes.data, patch.nbytes) == 0
    out = np.zeros_like(base)
    assert hip.jh_image_download(ctx, iid, out.ctypes.data, out.nbytes) == 0
    want = base.copy()
    want[3:8, 11:20] = patch
    assert np.array_equal(out, want)
    # outside the image, or too little data: error codes, image untouched
    assert hip.jh_image_write(ctx, iid, 30, 3, 9, 5, patch.ctypes.data, patch.nbytes) < 0
    assert hip.jh_image_write(ctx, iid, 0, 0, 9, 5, patch.ctypes.data, 10) < 0
    assert hip.jh_image_write(ctx, 0xdead, 0, 0, 1, 1, patch.ctypes.data, 4) < 0
    assert hip.jh_image_download(ctx, iid, out.ctypes.data, out.nbytes) == 0
    assert np.array_equal(out, want)
    assert hip.jh_image_free(ctx, iid) == 0


def test_more_images_than_the_inline_descriptor_set(engine):
    """The reference binds an array of up to 2048 textures (wgpu.go:278); past the 8 descriptors that travel in the
    kernel arguments fine indexes a device table.  Every one of the 13 images must really be sampled."""
    s, p = scenes.scene_many_images(13)
    r = compare(engine, s, p)
    img = r["image"].view(np.float16).astype(np.float32)
    for k in range(13):
        cx, cy = 16 + 36 * (k % 5), 16 + 36 * (k // 5)
        assert img[cy, cx, 3] > 0.9, k  # opaque texels of image k reached the target


def test_nested_profile_groups(engine):
    """Profiler.Start / Nest / Compute / Collect (profiler.go): the tree has the caller's group, RunRecording below it,
    one query per dispatch below that, GPU intervals ordered and nested."""
    s, p = scenes.scene_c1()
    rec = jello_amd.Host().record(s, p)
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    engine.profile(True)
    try:
        with engine.profile_group("frame 1"):
            engine.run(rec, RUN_DISPATCHES)
        with engine.profile_group("frame 2"):
            engine.run(rec, RUN_DISPATCHES)
        nodes = engine.profile_collect_tree()
    finally:
        engine.profile(False)
        engine.release(rec)
    tops = [i for i, n in enumerate(nodes) if n["parent"] == -1]
    assert [nodes[i]["label"] for i in tops] == ["frame 1", "frame 2"]
    n_dispatch = sum(1 for c in rec.commands() if c["kind"] in (jello_amd.CMD.DISPATCH, jello_amd.CMD.DISPATCH_INDIRECT))
    for t in tops:
        kids = [i for i, n in enumerate(nodes) if n["parent"] == t]
        assert len(kids) == 1 and nodes[kids[0]]["label"] == "RunRecording" and nodes[kids[0]]["kind"] == "group"
        qs = [n for n in nodes if n["parent"] == kids[0]]
        assert len(qs) == n_dispatch and all(q["kind"] == "query" for q in qs)
        assert qs[0]["label"] == "pathtag_reduce" and qs[-1]["label"] == "fine_area"
        for a, b in zip(qs, qs[1:]):
            assert a["gpu_start_ms"] <= a["gpu_end_ms"] <= b["gpu_start_ms"] + 1e-3
        g = nodes[t]
        assert g["gpu_start_ms"] <= qs[0]["gpu_start_ms"] and g["gpu_end_ms"] >= qs[-1]["gpu_end_ms"]
        assert g["cpu_end_ms"] >= g["cpu_start_ms"]
    assert nodes[tops[1]]["gpu_start_ms"] >= nodes[tops[0]]["gpu_end_ms"] - 1e-3
    assert engine.profile_collect_tree() == []  # collected once


def test_stale_graph_is_refused(engine):
    """A captured frame holds raw device pointers: after one of its buffers went back to the pool the replay must be
    refused (JH_ERR_INVALID), not run into freed memory."""
    s, p = scenes.scene_c1()
    rec = jello_amd.Host().record(s, p)
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    g = engine.capture(rec)
    engine.replay(g)
    engine.sync()
    t = rec.target
    a = engine.download_image(t["id"], t["width"], t["height"]).copy()
    engine.replay(g)  # still valid: nothing was freed
    engine.sync()
    assert np.array_equal(a, engine.download_image(t["id"], t["width"], t["height"]))
    engine.release(rec)  # frees every buffer of the frame
    assert engine.hip.jh_graph_launch(engine.ctx, g) < 0
    assert b"stale" in engine.hip.jh_last_error(engine.ctx)
    engine.graph_destroy(g)


def test_graph_is_stale_after_a_different_config_uniform(engine):
    """The launchers choose kernel instantiations (clip / no-clip fine and coarse) and grids from the host shadow of the
    uploaded ConfigUniform at capture time: uploading a DIFFERENT uniform under the same buffer id must make the captured
    frame stale (JH_ERR_INVALID), re-uploading the SAME bytes must not."""
    from jello_amd.engine import CMD
    s, p = scenes.scene_c1()
    rec = jello_amd.Host().record(s, p)
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    g = engine.capture(rec)
    engine.replay(g)
    engine.sync()
    cfg = [c for c in rec.commands() if c["kind"] == CMD.UPLOAD_UNIFORM][0]
    same = np.frombuffer(cfg["data"], dtype=np.uint8).copy()
    hip = engine.hip
    hip.jh_upload.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    assert hip.jh_upload(engine.ctx, cfg["buf_id"], same.ctypes.data, same.nbytes) == 0
    engine.replay(g)  # identical uniform: still valid
    engine.sync()
    other = same.copy()
    other.view(np.uint32)[10] = 3  # ConfigUniform.n_clip: the clip instantiations would be needed
    assert hip.jh_upload(engine.ctx, cfg["buf_id"], other.ctypes.data, other.nbytes) == 0
    assert hip.jh_graph_launch(engine.ctx, g) < 0
    assert b"stale" in hip.jh_last_error(engine.ctx)
    engine.graph_destroy(g)
    engine.release(rec)


def test_graph_replays_and_eager_frames_interleave(engine):
    """A captured frame contains no fill launches for the counters its kernels clean themselves (flatten's list counters,
    backdrop's wide-row counter): replays, and eager frames of another scene in between, must leave them clean."""
    def frame(rec):
        t = rec.target
        return (engine.download_image(t["id"], t["width"], t["height"]).copy(),
                engine.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8].copy())
    host = jello_amd.Host()
    sa, pa = scenes.scene_large_shapes()        # wide rows: the backdrop list route
    sb, pb = scenes.scene_c3(3000, 512)
    pb.bump = BumpSizes(ptcl=1 << 22)
    ra, rb = host.record(sa, pa), host.record(sb, pb)
    engine.run(ra, RUN_UPLOADS | RUN_DISPATCHES)
    engine.sync()
    img_a, bump_a = frame(ra)
    assert bump_a[0] == 0
    g = engine.capture(ra)
    engine.run(rb, RUN_UPLOADS | RUN_DISPATCHES)
    engine.sync()
    img_b, bump_b = frame(rb)
    for _ in range(3):
        engine.replay(g)
        engine.sync()
        i, b = frame(ra)
        assert np.array_equal(i, img_a) and np.array_equal(b, bump_a)
        engine.run(rb, RUN_DISPATCHES)
        engine.sync()
        i, b = frame(rb)
        assert np.array_equal(i, img_b) and np.array_equal(b, bump_b)
    engine.graph_destroy(g)
    engine.release(ra)
    engine.release(rb)


def test_graph_replay_on_dirty_counters_cleans_them_first(engine):
    """VERDICT r03 #7 / ADVICE r03: a captured frame has no fill for the counters its kernels leave clean, so a replay right
    after a frame that did NOT leave them clean (here: scratch poisoned with 0xA5, the state after a stage that failed half-way
    or a fresh allocation) used to run the look-back scan and the fused pathtag reduce on garbage.  jh_graph_launch now
    notices the lowered flags, zeroes those counters and replays; the frame is the eager frame."""
    s, p = scenes.scene_c3(40000, 1024)   # large scan path: the fused pathtag counter, the look-back descriptors, flatten's lists
    p.bump = s.bump_sizes(1024, 1024)
    rec = jello_amd.Host().record(s, p)
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    engine.sync()
    t = rec.target
    img = engine.download_image(t["id"], t["width"], t["height"]).copy()
    bump = engine.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8].copy()
    g = engine.capture(rec)
    assert engine.graph_node_counts(g)[1] == 0        # captured on clean counters: no fill inside
    before = engine.graph_self_cleans()
    engine.replay(g)
    engine.sync()
    assert engine.graph_self_cleans() == before        # a clean context replays as is
    for _ in range(3):
        engine.debug_poison_scratch(0xA5)
        engine.replay(g)
        engine.sync()
        assert np.array_equal(engine.download_image(t["id"], t["width"], t["height"]), img)
        assert np.array_equal(engine.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8], bump)
    assert engine.graph_self_cleans() == before + 3
    # a capture taken while the counters are dirty carries its own fills, and leaves the host's picture of the device as it was
    engine.debug_poison_scratch(0x5A)
    g2 = engine.capture(rec)
    assert engine.graph_node_counts(g2)[1] > 0
    engine.run(rec, RUN_DISPATCHES)                    # an eager frame straight after that capture must still fill
    engine.sync()
    assert np.array_equal(engine.download_image(t["id"], t["width"], t["height"]), img)
    engine.debug_poison_scratch(0x33)
    engine.replay(g2)
    engine.sync()
    assert np.array_equal(engine.download_image(t["id"], t["width"], t["height"]), img)
    engine.graph_destroy(g)
    engine.graph_destroy(g2)
    engine.release(rec)


def _layers_scene(depth, size):
    """`depth` nested blend layers (alternating mix modes, alpha 0.75), something drawn at every level."""
    s = jello_amd.Scene()
    mixes = [jello_amd.Mix.Multiply, jello_amd.Mix.Normal, jello_amd.Mix.Screen, jello_amd.Mix.Difference, jello_amd.Mix.Overlay, jello_amd.Mix.Darken]
    for k in range(depth):
        m = 8.0 * (k + 1)
        s.push_layer(mixes[k % len(mixes)], jello_amd.Compose.SrcOver, 0.75, None, jello_amd.Path.circle(size / 2, size / 2, size / 2 - m))
        s.fill(jello_amd.Fill.NonZero, None, jello_amd.Brush.solid((0.2 + 0.1 * k, 0.9 - 0.1 * k, 0.5, 0.6)), None,
               jello_amd.Path.rect(m, size / 3, size - m, 2 * size / 3))
    for _ in range(depth):
        s.pop_layer()
    return s, jello_amd.RenderParams(size, size, base_color=(0.1, 0.2, 0.3, 1.0))


@pytest.mark.parametrize("depth", [1, 2, 3, 4, 6])
def test_blend_stack_scratch_follows_the_clip_depth(depth):
    """fine keeps blend-stack levels 1..3 in a per-tile scratch slice (level 0 in LDS, levels >= 4 in blend_spill as in the WGSL).
    Round 3 reserved all three for every tile of any scene with a clip (12 KiB per tile); the engine now passes the nesting depth
    it counts off the draw tags (jh_set_clip_depth_hint) and fine reserves depth - 1 levels.  Image and buffers still equal the
    oracle's at every depth, on a context of its own so that the scratch array's size can be read."""
    size = 2048
    eng = jello_amd.Engine(0)
    try:
        s, p = _layers_scene(depth, size)
        # (small line buffers keep flatten's share of the same scratch slot small; depth 6 spills two levels per tile)
        p.bump = BumpSizes(lines=1 << 16, seg_counts=1 << 18, segments=1 << 18, blend_spill=1 << 24)
        compare(eng, s, p)
        tiles = (size // 16) ** 2
        levels = min(max(depth - 1, 0), 3)
        cap = eng.hip.jh_debug_scratch_bytes(eng.ctx, 4)  # JH_SCR_D (shared with flatten's piece records: a few MiB here)
        assert cap >= tiles * levels * 4096
        if levels < 3:
            assert cap < tiles * 3 * 4096, (cap, tiles * 3 * 4096)
        n = ctypes.c_uint32(99)
        assert eng.hip.jh_debug_clip_hint_overflows(eng.ctx, ctypes.byref(n), 0) == 0 and n.value == 0  # a correct hint drops no save
    finally:
        eng.close()


def test_a_clip_depth_hint_that_is_too_small_is_detectable():
    """jh_set_clip_depth_hint is sticky context state and a hint below the scene's real nesting depth drops blend-stack
    saves (wrong colours, never a fault).  A direct user of the ABI can tell: jh_debug_clip_hint_overflows counts the dropped
    saves.  Here the fine stage of a 3-deep scene is dispatched by hand with the hint forced to 1."""
    from jello_amd.engine import CMD

    class Binding(ctypes.Structure):
        _fields_ = [("kind", ctypes.c_uint32), ("count", ctypes.c_uint32), ("id", ctypes.c_uint64), ("ids", ctypes.POINTER(ctypes.c_uint64))]
    eng = jello_amd.Engine(0)
    try:
        hip, ctx = eng.hip, eng.ctx
        hip.jh_dispatch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(Binding), ctypes.c_int]
        s, p = _layers_scene(3, 512)
        p.bump = BumpSizes(lines=1 << 16, seg_counts=1 << 18, segments=1 << 18, blend_spill=1 << 20)
        rec = jello_amd.Host().record(s, p)
        eng.run(rec, RUN_UPLOADS | RUN_DISPATCHES)  # (every buffer of the frame exists afterwards; fine is then dispatched again by hand)
        fine = [c for c in rec.commands() if c["kind"] == CMD.DISPATCH and c["shader"] == 19][0]  # JH_FINE_AREA
        keep = []

        def to_binding(b):
            if b["kind"] == 3:
                arr = (ctypes.c_uint64 * max(1, len(b["ids"])))(*b["ids"])
                keep.append(arr)
                return Binding(3, len(b["ids"]), 0, ctypes.cast(arr, ctypes.POINTER(ctypes.c_uint64)))
            return Binding(b["kind"], 0, b["id"], None)
        bs = (Binding * len(fine["bindings"]))(*[to_binding(b) for b in fine["bindings"]])
        n = ctypes.c_uint32(0)
        for hint, want_drops in ((3, False), (0, False), (1, True)):
            assert hip.jh_set_clip_depth_hint(ctx, hint) == 0
            assert hip.jh_dispatch(ctx, 19, fine["wg"][0], fine["wg"][1], fine["wg"][2], bs, len(bs)) == 0, hip.jh_last_error(ctx)
            assert hip.jh_debug_clip_hint_overflows(ctx, ctypes.byref(n), 1) == 0
            assert (n.value > 0) == want_drops, (hint, n.value)
        hip.jh_set_clip_depth_hint(ctx, 0)
        eng.release(rec)
    finally:
        eng.close()


def test_launches_per_frame_of_a_large_scene(engine):
    """The launch diet, counted on the captured graph: a scene on the three-level pathtag path (more than 256 tag workgroups)
    is 25 kernel launches (27 until round 5: k_pc_paths is folded into k_pc_count / k_pc_emit, the last pathtag scan rides in
    flatten's classification kernel) and no fill -- the held-back commands (the last pathtag scan, bbox_clear, Clear(bump), both
    setup dispatches, pathtag_reduce2) and the single-launch scans are all in effect; pathtag_reduce and pathtag_scan1 (+ reduce2 in passing)
    of the large scan path are launches of their own again since round 4 (as ONE launch with a release / acquire hand-off
    they were slower) -- and the replay reproduces the eager frame."""
    s, p = scenes.scene_c3(40000, 1024)
    p.bump = s.bump_sizes(1024, 1024)
    rec = jello_amd.Host().record(s, p)
    assert rec.config["pathdata_base"] - rec.config["pathtag_base"] > 256 * 256  # tag words: the large scan path
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    engine.sync()
    t = rec.target
    img = engine.download_image(t["id"], t["width"], t["height"]).copy()
    bump = engine.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8].copy()
    assert bump[0] == 0
    g = engine.capture(rec)
    kernels, others = engine.graph_node_counts(g)
    assert kernels == 25 and others == 0, (kernels, others)
    for _ in range(3):
        engine.replay(g)
        engine.sync()
        assert np.array_equal(engine.download_image(t["id"], t["width"], t["height"]), img)
        assert np.array_equal(engine.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8], bump)
    engine.graph_destroy(g)
    engine.release(rec)


def test_two_graphs_for_two_output_buffers(engine):
    """bench.py's N > 1 path renders into two caller-owned images in turn (the RCCL gather of frame i reads one while
    frame i + 1 is rendered into the other): one captured graph per output, replayed alternately.  Re-importing the
    target under another pointer must not invalidate the graph captured for the first one."""
    hip, ctx = engine.hip, engine.ctx
    s, p = scenes.scene_c3(2000, 256)
    rec = jello_amd.Host().record(s, p)
    t = rec.target
    nbytes = t["width"] * t["height"] * 8
    ids = (0x6601, 0x6602)
    ptrs = []
    for i in ids:
        assert hip.jh_buffer_create(ctx, i, nbytes) == 0
        ptrs.append(hip.jh_buffer_device_ptr(ctx, i))
        assert ptrs[-1]
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES, ptrs[0])
    engine.sync()
    want = engine.download(ids[0], nbytes).copy()
    assert want.any()
    graphs = [engine.capture(rec, q) for q in ptrs]
    for i in ids:
        engine.clear(i)
    for k in range(6):
        engine.replay(graphs[k & 1])
    engine.sync()
    for i in ids:
        assert np.array_equal(engine.download(i, nbytes), want)
    for g in graphs:
        engine.graph_destroy(g)
    engine.release(rec)
    for i in ids:
        assert hip.jh_free(ctx, i) == 0


def test_two_contexts_in_flight_render_the_same_frame(engine):
    """bench.py --in-flight 2 (the default): two engine contexts on ONE device -- each with a stream, buffers, scratch and a captured
    graph of its own -- take the frames in turn, so that their kernels run side by side on the device.  Every replay of either
    context must leave exactly the frame a lone eager render produces (image, bump allocators, PTCL), however the two interleave."""
    import torch
    dev = torch.device("cuda", 0)
    s, p = scenes.scene_c4(1500, 512)
    p.bump = s.bump_sizes(512, 512)
    rec = jello_amd.Host().record(s, p)
    t = rec.target
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    engine.sync()
    want_img = engine.download_image(t["id"], t["width"], t["height"]).copy()
    want_bump = engine.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8].copy()
    want_ptcl = engine.download(rec.buffer("ptclBuf")[0], dtype=np.uint32).copy()
    from parity import ptcl_walk
    live = ptcl_walk(want_ptcl, rec.config)  # (words no command stream reaches are never written: whatever the allocation held)
    assert want_bump[0] == 0 and want_img.any() and live.sum() > 1024
    engine.release(rec)
    ctxs = []
    try:
        for k in range(2):
            e = jello_amd.Engine(0)
            st = torch.cuda.Stream(dev)
            e.set_stream(st.cuda_stream)
            out = torch.zeros((t["height"], t["width"], 4), dtype=torch.float16, device=dev)
            e.run(rec, RUN_UPLOADS | RUN_DISPATCHES, out.data_ptr())
            e.sync()
            ctxs.append((e, st, out, e.capture(rec, out.data_ptr())))
        for rounds in range(3):
            for e, st, out, g in ctxs:
                out.zero_()
            torch.cuda.synchronize(dev)
            for i in range(40):  # 40 frames, dealt round robin, nothing waits in between
                e, st, out, g = ctxs[i & 1]
                e.replay(g)
            torch.cuda.synchronize(dev)
            for e, st, out, g in ctxs:
                assert np.array_equal(out.cpu().numpy().view(np.uint16), want_img.view(np.uint16).reshape(out.shape))
                assert np.array_equal(e.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8], want_bump)
                assert np.array_equal(e.download(rec.buffer("ptclBuf")[0], dtype=np.uint32)[live], want_ptcl[live])
    finally:
        for e, st, out, g in ctxs:
            e.graph_destroy(g)
            e.release(rec)
            e.close()


def test_gather_pipeline_orders_each_buffer_on_its_own_stream():
    """The N > 1 step loop of bench.py with two frames in flight, on ONE GPU with a stand-in for RCCL: `gather` copies the frame on a
    side stream that waits for the stream it was called on, `work.wait()` makes the calling stream wait for that copy -- the stream
    semantics of an asynchronous NCCL collective.  Every gathered frame must be the frame of ITS step (a render that overwrote a buffer
    before its gather had read it, or a gather that did not wait for its render, shows up as a wrong value), in all three modes,
    although nothing here ever synchronises the two per-buffer streams with each other."""
    import torch
    from jello_amd import sharding
    dev = torch.device("cuda", 0)
    n = 1 << 22  # 16 MB frames: the copies take long enough to be overtaken if nothing orders them

    class Work:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream(dev).wait_event(self.ev)

    class FakeDist:  # rank 0 of a "world" of 2; rank 1's frame is a copy of ours
        def __init__(self):
            self.side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
            self.sink = torch.empty(n, dtype=torch.int32, device=dev)

        def gather(self, local, out, dst=0, async_op=False, group=None):
            side = self.side[group]
            side.wait_stream(torch.cuda.current_stream(dev))  # the collective starts behind what the caller's stream holds
            with torch.cuda.stream(side):
                torch.cuda._sleep(2_000_000)  # ~1 ms before the frame is read: a caller that does not wait is caught (checked once
                if out is not None:           # with wait() as a no-op: the values below come out wrong)
                    for o in out:
                        o.copy_(local, non_blocking=True)
                else:
                    self.sink.copy_(local, non_blocking=True)  # a send: the frame is read all the same
                ev = torch.cuda.Event()
                ev.record(side)
            return Work(ev)

    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    outs = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
    gathered = [[torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)] for _ in range(2)]
    pipe = sharding.GatherPipeline(FakeDist(), 0, 2, outs, gathered, [0, 1], streams=streams, alternate=True)
    for mode, base in ((None, 1000), ("0", 2000), ("rotate", 3000)):
        seen = {}
        step_of = []

        def render(k, _s=step_of):
            i = len(_s)
            _s.append(k)
            assert torch.cuda.current_stream(dev) == streams[k]
            outs[k].fill_(base + i)  # (on streams[k]: the pipeline made it the current stream)

        def on_gathered(step, bufs, _seen=seen):
            # (enqueued on the buffer's stream behind the wait for its gather: reads what the gather delivered)
            _seen[step] = [b[::4099].clone() for b in bufs]
        for i in range(12):
            pipe.step(i, render, mode, on_gathered)
        pipe.drain(on_gathered)
        torch.cuda.synchronize(dev)
        assert step_of == [0, 1] * 6
        want_steps = {None: [], "0": list(range(12)), "rotate": [0, 2, 4, 6, 8, 10]}[mode]
        assert sorted(seen) == want_steps
        for step, bufs in seen.items():
            for b in bufs:
                assert int(b.min()) == base + step and int(b.max()) == base + step, (mode, step, int(b.min()), int(b.max()))


def test_too_small_buffers_are_refused_not_read(engine):
    hip, ctx = engine.hip, engine.ctx
    hip.jh_dispatch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(Binding), ctypes.c_int]
    small, big = 0x5511, 0x5512
    z = np.zeros(4096, np.uint8)
    assert hip.jh_upload(ctx, small, z.ctypes.data, 16) == 0      # 16 bytes: cannot hold a ConfigUniform or BumpAllocators
    assert hip.jh_upload(ctx, big, z.ctypes.data, 4096) == 0
    # bbox_clear: [config, path_bboxes] with a 16-byte "config"
    b = (Binding * 2)(Binding(1, 0, small, None), Binding(1, 0, big, None))
    assert hip.jh_dispatch(ctx, 5, 1, 1, 1, b, 2) < 0
    # path_count_setup: [bump, indirect] with a 16-byte bump buffer
    b = (Binding * 2)(Binding(1, 0, small, None), Binding(1, 0, big, None))
    assert hip.jh_dispatch(ctx, 14, 1, 1, 1, b, 2) < 0
    # an imported (caller-owned) buffer cannot be grown by an upload (the "caller's" memory here is another buffer of the
    # context: importing torch just for 64 bytes can take minutes on a cold box)
    assert hip.jh_buffer_create(ctx, 0x5514, 4096) == 0
    mem_ptr = hip.jh_buffer_device_ptr(ctx, 0x5514)
    assert mem_ptr
    assert hip.jh_buffer_import(ctx, 0x5513, mem_ptr, 64) == 0
    assert hip.jh_upload(ctx, 0x5513, z.ctypes.data, 4096) < 0
    assert hip.jh_upload(ctx, 0x5513, z.ctypes.data, 64) == 0
    engine.sync()
    for i in (small, big, 0x5513, 0x5514):
        assert hip.jh_free(ctx, i) == 0
    s, p = scenes.scene_c1()
    rec, bump, attempts = engine.render(s, p, retain=False)  # the context is still usable
    assert bump["failed"] == 0


def test_scratch_of_the_headline_frame_and_trim():
    """The internal scratch of a context rendering C3 (100 k paths, 4096^2) with buffers sized for the frame -- the count / offset
    arrays of the deterministic allocators and flatten's temporary (8 B per line + 64 B per piece record, capacity = the line
    buffer's) -- stays below 0.4 GB (round 4: 1.4 GB, the temporary alone 116 B x 11.7 M slots).  The arrays only grow: after the
    first attempt with the estimator's sizes (three times the lines) jh_scratch_trim gives them back, and the frame after it is
    the same frame."""
    import hashlib
    eng = jello_amd.Engine(0)
    try:
        s, p = scenes.scene_c3(100_000, 4096)
        p.bump = s.bump_sizes(p.width, p.height)
        rec0, bump, attempts = eng.render(s, p, robust=True)
        assert bump["failed"] == 0 and attempts == 1
        img0 = hashlib.sha256(eng.download_image(rec0.target["id"], 4096, 4096).tobytes()).hexdigest()
        generous = eng.scratch_bytes()
        eng.release(rec0)
        eng.trim_scratch()
        assert eng.scratch_bytes() == 0
        margin = lambda x: int(x * 1.1) + 4096
        p.bump = BumpSizes(lines=margin(bump["lines"]), seg_counts=margin(bump["seg_counts"]), segments=margin(bump["segments"]),
                           tiles=margin(bump["tile"]), ptcl=margin(bump["ptcl"] + 256 * 256 * 64), bin_data=margin(bump["binning"] + 200_000))
        rec1, bump1, attempts1 = eng.render(s, p, robust=True)
        assert bump1 == bump and attempts1 == 1
        assert hashlib.sha256(eng.download_image(rec1.target["id"], 4096, 4096).tobytes()).hexdigest() == img0
        sized = eng.scratch_bytes()
        assert sized <= 400_000_000, [eng.scratch_bytes(k) for k in range(14)]
        assert generous > sized
        eng.release(rec1)
    finally:
        eng.close()


@pytest.mark.parametrize("flags", [3, 7])
@pytest.mark.parametrize("which", ["c3", "c2", "polygon", "c4"])
def test_flatten_regions_fill_up_and_are_left_behind(which, flags):
    """flatten's temporary (a slot per line, a record per piece or direct line) is cut into up to eight regions with a cursor each
    (kernels_flatten.hip, FlTemp).  A frame only fills a region up -- and leaves it for the next one, the slots at its end marked
    empty -- when it comes close to its line buffer's capacity.  Here every frame does: eight regions whatever the size, every
    wave starting in region 0 (jh_debug_flatten_regions), and a line buffer one line larger than the frame; curves (c3),
    round joins and caps (c2: arcs of many lines), 170 k straight segments (polygon) and clip scenes (c4); with flag 4 every
    batch of more than 48 lines also allocates its slots job by job (the product: above 51 200 lines, which no test scene
    reaches).  Everything against the oracle, which knows nothing of any of this."""
    s, p = {"c3": lambda: scenes.scene_c3(20000, 1024), "c2": lambda: scenes.scene_c2(4000, 1024),
            "polygon": lambda: scenes.scene_dense_polygon(170000, 512), "c4": lambda: scenes.scene_c4(6000, 1024)}[which]()
    eng = jello_amd.Engine(0)
    try:
        p.bump = s.bump_sizes(p.width, p.height)
        _, bump, _ = eng.render(s, p, robust=True)
        assert bump["failed"] == 0 and bump["lines"] > 8 * 51200 // 7  # (more lines than region 0 holds)
        p.bump.lines = bump["lines"] + 1
        assert eng.hip.jh_debug_flatten_regions(eng.ctx, flags) == 0
        r = compare(eng, s, p)
        assert r["bump"]["failed"] == 0 and r["bump"]["lines"] == bump["lines"]
    finally:
        eng.close()


@pytest.mark.parametrize("n,size", [(300, 256), (40000, 1024)])
def test_held_back_commands_and_commands_as_recorded_give_the_same_frame(engine, n, size):
    """The engine holds back the last pathtag scan, bbox_clear, Clear(bump), both setup dispatches and pathtag_reduce2 and lets the
    stage behind them do their work in passing (jello_hip.cpp, Deferred); with the profiler on every command is launched as
    recorded (every query must time its own stage).  Both routes against the oracle, and against each other word for word:
    the small scene takes pathtag_scan_small, the large one the three-level path with pathtag_scan_large."""
    s, p = scenes.scene_c3(n, size)
    p.bump = s.bump_sizes(size, size)
    names = ["tagmonoidBuf", "pathBboxBuf", "linesBuf", "bumpBuf", "segCountsBuf", "ptclBuf"]
    compare(engine, s, p)  # the held-back route against the oracle
    rec = jello_amd.Host().record(s, p)
    large = rec.config["pathdata_base"] - rec.config["pathtag_base"] > 256 * 256
    assert large == (n == 40000)
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    engine.sync()
    held = {nm: engine.download(rec.buffer(nm)[0], dtype=np.uint32).copy() for nm in names}
    n_lines = int(held["bumpBuf"][7])
    engine.profile(True)
    try:
        with engine.profile_group("as recorded"):
            engine.run(rec, RUN_DISPATCHES)
        nodes = engine.profile_collect_tree()
    finally:
        engine.profile(False)
    labels = [nd["label"] for nd in nodes if nd["kind"] == "query"]
    assert ("pathtag_scan_large" if large else "pathtag_scan_small") in labels and "bbox_clear" in labels
    engine.sync()
    for nm in names:
        got = engine.download(rec.buffer(nm)[0], dtype=np.uint32)
        lim = {"linesBuf": 6 * n_lines, "bumpBuf": 8}.get(nm, got.size)
        if nm in ("segCountsBuf", "ptclBuf"):
            continue  # (compared by compare() through the command streams; their dead words keep what the buffer held)
        assert np.array_equal(got[:lim], held[nm][:lim]), nm
    engine.release(rec)
    compare(engine, s, p)  # and the held-back route again, after a frame that ran as recorded
