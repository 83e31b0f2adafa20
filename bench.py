#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: "Mpixels/sec fine-raster + paths/sec, 100k-path
4096^2 scene, 1/2/4/8 MI355X".

A step = one full pass of the hot path (pathtag scan -> flatten -> draw/clip scans -> binning ->
tile_alloc -> path_count -> backdrop -> coarse -> path_tiling -> fine) over one synthetic scene that is
already resident in HBM (scene bytes + config uploaded before the timed region).

`--gpus N` with N > 1 (and no torchrun environment) STARTS the N ranks itself, as child processes and
before this process has imported torch or touched the GPU (`jello_amd.sharding.launch_ranks` = the
driver's own `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`),
and exits with the launcher's code.  Under torchrun it is one rank: every rank renders its own
independent scene (C5: weak scaling, no data-path collective) and, as C5 is written, the finished
RGBA16F images are gathered on rank 0 over RCCL, double-buffered so that the transfer of frame i
overlaps the render of frame i+1.  For N > 1 ONE run times three modes (K steps each, each bracketed by barrier +
synchronize): no gather (frames stay where they were rendered), gather to rank 0 (C5 as written: this is `value`), and
gather to rank (step mod N) -- all three are printed under `modes` with the wire's ceiling for each
(`sharding.gather_model`) and the speed-up over one of this run's own ranks that each implies.  `--no-gather` /
`--gather-dst 0|rotate` restrict the run to one mode.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~4.8 TB/s is what a device copy reaches
BUMP_NAMES = ["failed", "binning", "ptcl", "tile", "seg_counts", "segments", "blend", "lines"]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", choices=["c3", "c4", "c4n", "c2", "c1"], default="c3",
                    help="c3 = the headline scene (BASELINE.json configs[2]); c4 = configs[3]: nested clips + radial gradients + blends "
                         "(clip circles at independent positions: nearly every paint is clipped away); c4n = the same with concentric "
                         "clips and the paths inside them (visible paints); c2 = configs[1]'s SUBSTITUTE (the Ghostscript tiger is not "
                         "available here): 300 filled / stroked blobs at 1024x1024; c1 = configs[0]: one filled rectangle + one stroked "
                         "cubic at 512x512 (a launch-bound frame)")
    ap.add_argument("--paths", type=int, default=0, help="default: 100000 (c3) / 30000 (c4)")
    ap.add_argument("--size", type=int, default=0, help="default: 4096 (c3) / 2048 (c4)")
    ap.add_argument("--aa", choices=["area", "msaa8", "msaa16"], default="area", help="coverage mode of the fine stage (the headline is area)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: time only the independent renders, no image gather")
    ap.add_argument("--gather-dst", choices=["all", "0", "rotate"], default="all",
                    help="N > 1: where the finished frames are collected: on rank 0 (C5 as written: one compositor GPU; its inbound "
                         "links bound the job) or on rank (step mod N) -- consecutive gathers then use disjoint inbound links and, "
                         "being double-buffered on two communicators, overlap each other as well as the next render; "
                         "all (default) = time both, after the gather-free loop, in one run")
    ap.add_argument("--blocks", type=int, default=5,
                    help="timed blocks of --steps steps each; the line reports the median block.  More blocks are added (up to 101) until "
                         "the timed region of a mode covers >= 1 s, so that a short --steps still keeps the GPU busy long enough to be seen")
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="blocks are added until a mode's timed region covers this long (0: exactly --blocks blocks; the counter passes "
                         "of tools/pmc*.sh use that -- every extra frame is rows in their CSVs)")
    ap.add_argument("--bands", action="store_true",
                    help="N > 1: ONE scene, every rank runs the element stages on it and coarse+fine for its band of bin rows "
                         "(strong scaling of one frame; SURVEY 8e) instead of one independent scene per rank")
    ap.add_argument("--emulate-band-of", type=int, default=0, metavar="N",
                    help="with --bands on ONE GPU: time what rank N/2 of an N-GPU band job would do (its band of the frame)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--in-flight", type=int, choices=[1, 2], default=2,
                    help="frames in flight per GPU: 2 (default) = two engine contexts (own stream, buffers, scratch, graph) take the steps "
                         "in turn, so the kernels of frame i + 1 fill the launch gaps and tails of frame i; 1 = one context, one "
                         "frame after the other (every round before round 4; also timed beside the default for comparison)")
    args = ap.parse_args(argv)
    if args.paths <= 0:
        args.paths = {"c3": 100_000, "c2": 300, "c1": 2}.get(args.scene, 30_000)
    if args.size <= 0:
        args.size = {"c3": 4096, "c2": 1024, "c1": 512}.get(args.scene, 2048)
    return args


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # Not under torchrun: become the launcher.  Nothing above imported torch or opened the device.
        from jello_amd import sharding
        rc = sharding.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus)
        if rc != 0:
            sys.stderr.write("bench.py: the %d-rank job failed (launcher exit code %d)\n" % (args.gpus, rc))
        sys.exit(rc)
    world = int(env_world or "1")
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    run_rank(args, world)


def run_rank(args, world):
    import numpy as np
    import torch
    import jello_amd
    from jello_amd import BumpSizes, scenes, sharding
    from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if local_rank >= torch.cuda.device_count():
            raise RuntimeError("rank %d: this node has %d GPU(s), --gpus %d needs one per rank" % (rank, torch.cuda.device_count(), world))
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    W = H = args.size
    # independent scene per rank (same generator, different seed)
    seed_off = 0 if args.bands else rank
    if args.scene == "c3":
        scene, params = scenes.scene_c3(args.paths, args.size, seed=sharding.scene_seed_for_rank(seed_off))
        what = "C3: %d random stroked+filled cubic Beziers" % args.paths
    elif args.scene == "c1":
        scene, params = scenes.scene_c1()
        W = H = args.size = 512
        args.paths = 2
        what = "C1: one filled rectangle + one stroked cubic"
    elif args.scene == "c2":
        scene, params = scenes.scene_c2(args.paths, args.size, seed=scenes.SEED + 2 + seed_off)
        what = "C2 SUBSTITUTE (no Ghostscript tiger in this image): %d blobs of 3-6 cubics, filled (some even-odd) and stroked" % args.paths
    elif args.scene == "c4":
        scene, params = scenes.scene_c4(args.paths, args.size, seed=scenes.SEED + 4 + seed_off)
        what = "C4: %d paths in groups of 10 under 3-deep clip layers (16 mix modes), every 3rd brush a radial gradient" % args.paths
    else:
        scene, params = scenes.scene_c4_nested(args.paths, args.size, seed=scenes.SEED + 14 + seed_off)
        what = ("C4 (nested variant): %d paths in groups of 10 inside 3 concentric clip layers (16 mix modes), every 3rd brush a "
                "radial gradient" % args.paths)
    params.aa = {"area": jello_amd.Aa.Area, "msaa8": jello_amd.Aa.Msaa8, "msaa16": jello_amd.Aa.Msaa16}[args.aa]
    fine_stage = {"area": "fine_area", "msaa8": "fine_msaa8", "msaa16": "fine_msaa16"}[args.aa]
    eng = jello_amd.Engine(dev.index)
    host = jello_amd.Host()
    n_ctx = 1 if (args.no_graph or args.bands) else args.in_flight  # (eager launches and band mode keep the one-context loop)
    if n_ctx == 1:
        streams = [torch.cuda.current_stream(dev)]
        engs = [eng]
    else:  # one non-blocking stream per context (the legacy default stream would order the two against each other)
        streams = [torch.cuda.Stream(dev) for _ in range(n_ctx)]
        engs = [eng] + [jello_amd.Engine(dev.index) for _ in range(n_ctx - 1)]
    for e, st in zip(engs, streams):
        e.set_stream(st.cuda_stream)
    stream = streams[0]

    # ---- size the bump buffers once, outside the timed region: estimator first, regrow loop as the safety net ----
    params.bump = scene.bump_sizes(W, H)
    rec0, bump, attempts = eng.render(scene, params, robust=True)
    if bump["failed"]:
        raise RuntimeError("bump allocation still failing after regrow: %s" % bump)
    cfg0 = rec0.config
    margin = lambda x: int(x * 1.1) + 4096
    params.bump = BumpSizes(lines=margin(bump["lines"]), seg_counts=margin(bump["seg_counts"]), segments=margin(bump["segments"]),
                            tiles=margin(bump["tile"]), ptcl=margin(bump["ptcl"] + cfg0["width_in_tiles"] * cfg0["height_in_tiles"] * 64),
                            bin_data=margin(bump["binning"] + cfg0["bin_data_start"]), blend_spill=max(4096, margin(bump["blend"])))
    del rec0
    # (the sizing render ran with the estimator's sizes -- three times the lines this frame has -- and the context's internal
    # scratch arrays only grow: give them back, the timed frames allocate what their own sizes need)
    eng.trim_scratch()
    rec = host.record(scene, params)
    cfg = rec.config

    gather_modes = []  # which gathers are timed behind the gather-free loop
    if world > 1 and not args.no_gather and not args.bands:
        gather_modes = ["0", "rotate"] if args.gather_dst == "all" else [args.gather_dst]
    gather = bool(gather_modes)
    # output images: torch owns the device memory (double-buffered for the overlapped gather)
    outs = [torch.empty((H, W, 4), dtype=torch.float16, device=dev) for _ in range(2 if (gather or n_ctx == 2) else 1)]
    eng_of = [engs[k % n_ctx] for k in range(len(outs))]  # buffer k is rendered by context k (one context: both by the same)
    gathered = None
    gather_groups = [None, None]
    mode_errors = {}  # gather mode -> why it could not be timed (the gather-free mode and the line survive it)
    if gather:
        try:
            # one communicator per buffer of the double buffer: two gathers on ONE communicator run one after the other
            gather_groups = [dist.new_group(ranks=list(range(world))) for _ in range(2)]
            if rank == 0 or "rotate" in gather_modes:
                gathered = [[torch.empty((H, W, 4), dtype=torch.float16, device=dev) for _ in range(world)] for _ in range(2)]
        except Exception as e:  # noqa: BLE001 - RCCL / allocation trouble must not cost the gather-free measurement
            for m in gather_modes:
                mode_errors[m] = "gather set-up failed on rank %d: %s: %s" % (rank, type(e).__name__, e)
            gather_groups, gathered = [None, None], None

    if args.bands:  # (buffers were sized by the unsharded render above; from here on this rank owns its band only)
        hb = (cfg["height_in_tiles"] + 15) // 16
        if world == 1 and args.emulate_band_of > 1:
            eng.set_band(*sharding.band_for_rank(hb, args.emulate_band_of, args.emulate_band_of // 2))
        else:
            eng.set_band(*sharding.band_for_rank(hb, world, rank))
    for k in range(n_ctx):  # uploads scene/config; allocates every buffer (of every context)
        engs[k].run(rec, RUN_UPLOADS | RUN_DISPATCHES, outs[k].data_ptr())
    torch.cuda.synchronize(dev)

    def frame_digest(k):
        """SHA-256 of the bump allocators and of the finished image: what a frame IS, for the replay check below."""
        import hashlib
        h = hashlib.sha256()
        h.update(eng_of[k].download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8].tobytes())
        h.update(outs[k].cpu().numpy().tobytes())
        return h.hexdigest()
    eager_digest = frame_digest(0)  # an eagerly launched frame; every timed frame must reproduce it

    # One frame is ~45 short launches, so the dispatch-only replay of the recording is captured once into a hipGraph
    # per output buffer and the timed steps replay it.
    use_graph = not args.no_graph
    graphs = [eng_of[k].capture(rec, o.data_ptr()) for k, o in enumerate(outs)] if use_graph else []
    pipe = sharding.GatherPipeline(dist, rank, world, outs, gathered, gather_groups, streams=streams if n_ctx == 2 else None,
                                   alternate=n_ctx == 2)
    pipe_one = sharding.GatherPipeline(dist, rank, world, outs[:1], None, None)  # one context, one frame after the other

    def render(k):
        if use_graph:
            eng_of[k].replay(graphs[k])
        else:
            eng_of[k].run(rec, RUN_DISPATCHES, outs[k].data_ptr())

    def timed(mode, pipe=pipe, n_blocks=None):
        """W warmup steps, then --blocks blocks of exactly K steps, each between barrier + synchronize on both sides and
        each the max over ranks; returns the block times (seconds), sorted.  mode: None / "0" / "rotate" (GatherPipeline)."""
        for i in range(args.warmup):
            pipe.step(i, render, mode)
        blocks = []
        fixed = n_blocks is not None
        n_blocks = max(1, args.blocks if n_blocks is None else n_blocks)
        b = 0
        while b < n_blocks:
            b += 1
            pipe.drain()
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(args.steps):
                pipe.step(i, render, mode)
            pipe.drain()
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            blocks.append(el)
            if b == 1 and el > 0.0 and args.min_seconds > 0.0 and not fixed:  # (el is the max over ranks: every rank computes the same count)
                need = int(args.min_seconds / el) + 1
                n_blocks = max(n_blocks, min(101, need | 1))
        return sorted(blocks)

    blocks_plain = timed(None)
    # (two frames in flight: the one-context loop of the earlier rounds timed beside it, three blocks)
    blocks_one = timed(None, pipe_one, 3) if n_ctx == 2 else None
    # Each gather mode on its own: one that raises (the first contact of this path with RCCL on real hardware has not
    # happened yet) is reported as {"error": ...} under `modes`; the gather-free measurement above and the line survive,
    # and the process exits non-zero.  The ranks agree on a mode's fate through the default group (gloo-free: a MIN
    # all-reduce of one flag); if even that fails the remaining modes are skipped.
    def timed_mode(m):
        fresh = sharding.GatherPipeline(dist, rank, world, outs, gathered, gather_groups, streams=streams if n_ctx == 2 else None,
                                        alternate=n_ctx == 2)  # (a pipeline of its own: nothing pending from a mode that died)
        return timed(m, fresh)
    blocks_by_mode, mode_errors = sharding.time_modes_surviving_failures(dist, rank, gather_modes, timed_mode, dev, mode_errors) if gather_modes else ({}, mode_errors)
    good_modes = [m for m in gather_modes if m in blocks_by_mode]
    gather = bool(good_modes)
    # Which mode is the line's `value` (round 6): a gather to ONE compositor GPU is capped by that GPU's inbound links (sharding.gather_model:
    # 4.15 x at 8 GPUs at this render time, whatever the renderer does), so with `--gather-dst all` the headline is the gather whose
    # destination rotates when it ran, else the gather-free loop; gather-to-rank-0 stays under `modes` with its ceiling.  An explicit
    # `--gather-dst 0` / `rotate` is reported as asked.
    head_mode = sharding.headline_mode(args.gather_dst, good_modes) if gather else None
    gather = head_mode is not None
    blocks = blocks_by_mode[head_mode] if gather else blocks_plain
    elapsed_plain = blocks_plain[len(blocks_plain) // 2]
    elapsed = blocks[len(blocks) // 2]  # the median block
    # The frames that were timed are the frame that was checked: replayed (or re-run) frames must be bit-identical to the
    # eager one -- image and bump allocators -- or the number describes something else.
    for k, o in enumerate(outs):
        d = frame_digest(k)
        if d != eager_digest:
            raise RuntimeError("rank %d: the timed frame in output buffer %d differs from the eagerly launched frame (SHA-256 %s vs %s)"
                               % (rank, k, d[:16], eager_digest[:16]))

    # Per-stage device times: the same K steps once more, eagerly, with a hipEvent pair around every stage on the
    # launch stream (events cannot be read back from inside a replayed graph).
    eng.profile(True)
    for i in range(args.steps):
        eng.run(rec, RUN_DISPATCHES, outs[0].data_ptr())
    torch.cuda.synchronize(dev)
    prof = eng.profile_collect(1 << 16)
    eng.profile(False)
    stage_ms = {}
    for name, ms in prof:
        stage_ms[name] = stage_ms.get(name, 0.0) + ms
    stage_ms = {k: v / args.steps for k, v in stage_ms.items()}
    fine_ms = stage_ms.get(fine_stage, float("nan"))

    result = None
    if rank == 0:
        # ---- algorithmic bytes of the fine stage, from the PTCL this run produced ----
        ptcl_id, _ = rec.buffer("ptclBuf")
        ptcl = eng.download(ptcl_id, dtype=np.uint32)
        st = (ctypes.c_uint64 * 8)()
        if eng._L.jl_ptcl_stats(ptcl.ctypes.data, ptcl.size, cfg["width_in_tiles"], cfg["height_in_tiles"], st) != 0:
            raise RuntimeError("malformed PTCL")
        words, segs, info_words, texels, spill_px = st[0], st[1], st[2], st[3], st[4]
        b_fine = 4 * words + 24 * segs + 4 * info_words + 8 * texels + 32 * spill_px + 8 * W * H
        achieved = b_fine / (fine_ms * 1e-3) / 1e9
        bump_now = dict(zip(BUMP_NAMES, [int(v) for v in eng.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8]]))
        copy_gbs = measured_copy_gbs(torch, dev)
        kname = "k_fine_area" if args.aa == "area" else "k_fine_area<%s>" % args.aa
        roofline = {"kernel": kname, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes": int(b_fine), "avg_ms": round(fine_ms, 4),
                    "segment_pixel_evals": int(segs) * 256,
                    "peak_measured_copy": None if copy_gbs is None else round(copy_gbs, 1)}
        roofline.update(committed_counters(args, fine_ms))
        if args.bands and (world > 1 or args.emulate_band_of > 1):  # one band only: the whole-target byte count does not apply
            roofline = None
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(host, args)
        units = 1 if args.bands else world  # frames finished per step by the whole job
        per = elapsed / args.steps
        mode = ("bin-row bands of one scene x%d" % world) if args.bands else \
            "scene-per-gpu x%d%s" % (world, (" + RCCL image gather to %s (overlapped, double-buffered)" %
                                             ("rank 0" if head_mode == "0" else "rank (step mod N)")) if gather else "")
        if n_ctx == 2:
            mode += ", 2 frames in flight per GPU (two engine contexts take the steps in turn)"
        headline = args.scene == "c3" and args.paths == 100_000 and args.size == 4096
        metric = "Mpixels/sec fine-raster + paths/sec, 100k-path 4096^2 scene" if headline else \
            "Mpixels/sec fine-raster + paths/sec, %s scene, %d paths, %d^2 (NOT the headline configuration)" % (args.scene, args.paths, args.size)
        result = {
            "metric": metric, "value": round(W * H * units / per / 1e6, 2), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(per * 1e3, 4), "higher_is_better": True,
            "blocks": len(blocks), "block_ms_per_step": [round(b / args.steps * 1e3, 4) for b in blocks],
            "block_spread": round((blocks[-1] - blocks[0]) / elapsed, 4),
            "timed_frames_verified": "SHA-256 of image + bump allocators equals an eagerly launched frame (%s...)" % eager_digest[:16],
            "scaling": "strong" if args.bands else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %dx%d RGBA16F target, %s AA, %s" % (what, W, H, args.aa,
                                   "one scene split into bin-row bands" if args.bands else "one independent scene per GPU"),
                       "paths": args.paths, "draw_objects": cfg["n_drawobj"], "width": W, "height": H, "parallelism": mode},
            "value_definition": ("finished frames per second of the whole job x width x height: K steps between barrier + synchronize, "
                                 "%d frame(s) in flight per GPU%s" % (n_ctx, "" if world == 1 else ", %d GPUs, one scene each" % world)),
            "paths_per_s": round(args.paths * units / per, 1),
            "paths_per_s_definition": "a RATE (user paths of the frames finished per second), not SURVEY 8(d)'s scene-resident -> image-complete "
                                      "time of one frame: that is paths_per_s_latency",
            "paths_per_s_latency": None if blocks_one is None else round(args.paths / (blocks_one[len(blocks_one) // 2] / args.steps), 1),
            "fine_mpixels_per_s": round(W * H / (fine_ms * 1e-3) / 1e6, 2),
            "frames_in_flight": n_ctx,
            "one_frame_at_a_time": None if blocks_one is None else {
                "ms_per_step": round(blocks_one[len(blocks_one) // 2] / args.steps * 1e3, 4),
                "value": round(W * H * units / (blocks_one[len(blocks_one) // 2] / args.steps) / 1e6, 2),
                "note": "the gather-free loop on ONE context (no overlap between consecutive frames), as every round before round 4 timed it"},
            "launch": "hipGraph replay" if use_graph else "eager",
            "launches_per_frame": None if not use_graph else dict(zip(("kernels", "fills_and_copies"), eng.graph_node_counts(graphs[0]))),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_source": "eager replay of the same %d steps with a hipEvent pair per stage" % args.steps,
            "stage_ms_regime": "one context, eager (a stage's duration is its own: the sum is a frame alone, not ms_per_step with %d frames in flight)" % n_ctx,
            "bump": bump_now, "scratch_bytes_per_context": eng.scratch_bytes(), "sizing_attempts": attempts, "bump_estimate_clamped": scene.bump_sizes_clamped(W, H),
            "stage_roofline": stage_roofline(cfg, bump_now, stage_ms, rec),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "device": eng.device_info()["name"],
        }
        if world == 1:
            result["value_mode"] = "%d frame(s) in flight on one GPU%s" % (n_ctx, "" if n_ctx == 1 else " (the one-context figure of rounds 1-3 is one_frame_at_a_time.value)")
        if world > 1 and not args.bands:
            # Every mode of this run side by side.  `render` = the gather-free step of THIS run (every rank finishes one frame
            # per step), so N * render / step is the speed-up over one of this run's own ranks that a mode delivers.
            pp = elapsed_plain / args.steps
            frame_bytes = W * H * 8

            def mode_entry(name, blk, model):
                per_m = blk[len(blk) // 2] / args.steps
                e = {"value": round(W * H * world / per_m / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(per_m * 1e3, 4),
                     "block_ms_per_step": [round(b / args.steps * 1e3, 4) for b in blk],
                     "speedup_over_one_rank_of_this_run": round(world * pp / per_m, 2)}
                if model is not None:
                    e["wire_model"] = model
                    e["ceiling_speedup"] = model["ceiling_speedup"]
                else:
                    e["ceiling_speedup"] = float(world)
                return e
            result["modes"] = {"no_gather": mode_entry("no_gather", blocks_plain, None)}
            for m in gather_modes:
                key = "gather_to_rank0" if m == "0" else "gather_to_rank_step_mod_n"
                model = sharding.gather_model(world, pp * 1e3, frame_bytes, m)
                if m in blocks_by_mode:
                    result["modes"][key] = mode_entry(m, blocks_by_mode[m], model)
                else:
                    result["modes"][key] = {"error": mode_errors.get(m, "not timed"), "wire_model": model, "ceiling_speedup": model["ceiling_speedup"]}
            # north_star asks for >= 6 x at 8 GPUs: which mode can deliver that, from the wire alone, at this run's render time
            for key, m in (("no_gather", None), ("gather_to_rank0", "0"), ("gather_to_rank_step_mod_n", "rotate")):
                if key in result["modes"]:
                    c8 = 8.0 if m is None else sharding.gather_model(8, pp * 1e3, frame_bytes, m)["ceiling_speedup"]
                    result["modes"][key]["ceiling_speedup_at_8_gpus"] = c8
                    result["modes"][key]["can_meet_6x_at_8_gpus"] = bool(c8 >= 6.0)
            result["mode_errors"] = mode_errors or None
            result["value_mode"] = "no_gather" if not gather else ("gather_to_rank0" if head_mode == "0" else "gather_to_rank_step_mod_n")
            result["value_mode_rule"] = ("--gather-dst all: the gather with a rotating destination when it ran, else no gather; gather-to-rank-0 (C5 as written) is "
                                         "under modes.gather_to_rank0, wire-capped at modes.gather_to_rank0.ceiling_speedup_at_8_gpus x")
            result["value_no_gather"] = result["modes"]["no_gather"]["value"]
            result["ms_per_step_no_gather"] = round(pp * 1e3, 4)
            if gather:
                result["value_with_gather"] = result["value"]
                result["gather"] = sharding.gather_model(world, pp * 1e3, frame_bytes, head_mode)
    for k, g in enumerate(graphs):
        eng_of[k].graph_destroy(g)
    for e in engs:
        e.release(rec)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))
    if mode_errors:  # the line is out; a gather mode that failed still fails the run
        sys.exit(3)


def measured_copy_gbs(torch, dev):
    """Achievable HBM bandwidth on this device: a 1 GiB device-to-device copy (read + write bytes), reported next to
    the 8 TB/s vendor figure that `peak` uses (SURVEY 8d asks for both)."""
    try:
        ca = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        cb = torch.empty_like(ca)
        cb.copy_(ca)
        ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ce0.record()
        for _ in range(5):
            cb.copy_(ca)
        ce1.record()
        torch.cuda.synchronize(dev)
        return 5 * 2 * (1 << 30) / (ce0.elapsed_time(ce1) * 1e-3) / 1e9
    except Exception:  # noqa: BLE001 - the measurement is optional
        return None


def fine_source_sha256():
    """What the fine kernel is built from: the counters below are only meaningful for exactly these bytes."""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels_fine.hip", "kcommon.h", "dmath.h"):
        h.update(open(os.path.join(ROOT, "jello_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def committed_counters(args, fine_ms):
    """Counter-derived figures of the fine kernel.  Hardware counters cannot be read from inside this process, so they
    come from the committed rocprofv3 --pmc summary of THIS command (profiles/fine_counters.json, regenerated with
    tools/pmc_fine.sh) and carry the commit and workload they were measured on; null when that file does not describe
    this workload OR was measured on a different kernels_fine.hip (the summary records the SHA-256 of the kernel's
    sources; a kernel that has changed since has different traffic)."""
    out = {"traffic": None, "traffic_source": None, "valu_pipe_bound_ms": None}
    pm = os.path.join(ROOT, "profiles", "fine_counters.json" if args.scene == "c3" else "fine_counters_%s.json" % args.scene)
    if not os.path.exists(pm):
        return out
    try:
        c = json.load(open(pm))
    except Exception:  # noqa: BLE001
        return out
    if c.get("scene") != args.scene or c.get("paths") != args.paths or c.get("size") != args.size or c.get("aa") != args.aa:
        return out
    if c.get("kernel_source_sha256") != fine_source_sha256():
        out["traffic_source"] = ("profiles/%s was measured on other kernel sources (commit %s): not reported; regenerate it with "
                                 "tools/pmc_fine.sh" % (os.path.basename(pm), c.get("commit", "?")))
        return out
    out["traffic"] = c.get("hbm_bytes_per_launch")
    out["traffic_source"] = "profiles/%s: rocprofv3 --pmc passes at commit %s on these kernel sources (not re-measured in this run)" % (
        os.path.basename(pm), c.get("commit", "?"))
    # Pipe bounds from MEASURED issue costs (tools/ubench, profiles/r03_ubench_issue_rates.txt): a SIMD of gfx950 issues a wave64
    # v_add/mul/sub_f32, v_and/or/xor, v_add_u32 or v_mov in 2.2 cycles, almost everything else (min/max, fma, compares, selects,
    # conversions, DPP, packed f32, anything with an SGPR source) in 4.2, lane reads with a scalar lane select and transcendentals in
    # 8.1, and one scalar instruction per 4.08 cycles whatever the number of waves.  The vector figure prices the kernel's STATIC
    # instruction mix (tools/isa_price.py; dynamic trip counts are not known to the counters).
    if c.get("valu_insts_per_launch") and c.get("simds") and c.get("clock_ghz"):
        per = c.get("valu_cycles_per_inst_static_mix") or 4.2
        ib = c["valu_insts_per_launch"] * per / (c["simds"] * c["clock_ghz"] * 1e9) * 1e3
        out["valu_pipe_bound_ms"] = round(ib, 4)
        out["valu_pipe_bound_note"] = ("%.3g VALU wave-instructions per launch x %.2f cycles (static mix, measured issue costs) / (%d SIMDs x %.2f GHz); "
                                       "bound / measured kernel time = %.2f" % (c["valu_insts_per_launch"], per, c["simds"], c["clock_ghz"],
                                                                                  ib / fine_ms if fine_ms > 0 else float("nan")))
        if c.get("salu_insts_per_launch"):
            sb = c["salu_insts_per_launch"] * 4.08 / (c["simds"] * c["clock_ghz"] * 1e9) * 1e3
            out["salu_pipe_bound_ms"] = round(sb, 4)
        out["insts_per_tile"] = {k: round(c[v] / c["tiles"], 1) for k, v in (("valu", "valu_insts_per_launch"), ("salu", "salu_insts_per_launch"),
                                                                             ("lds", "lds_insts_per_launch")) if c.get(v) and c.get("tiles")}
    return out


def stage_roofline(cfg, bump, stage_ms, rec):
    """Algorithmic bytes per stage / measured stage time, against the 8 TB/s HBM peak.  These stages are
    irregular integer / f32 work; the figure says how far each is from being bandwidth-limited, nothing more."""
    n_tagw = cfg["pathdata_base"] - cfg["pathtag_base"]  # tag words (4 tag bytes each)
    n_draw, lines, cross, segs = cfg["n_drawobj"], bump["lines"], bump["seg_counts"], bump["segments"]
    try:
        scene_bytes = rec.buffer("scene")[1]
    except Exception:  # noqa: BLE001
        scene_bytes = 0
    per = {
        "pathtag (4 stages)": (24 * n_tagw, sum(stage_ms.get(k, 0.0) for k in ("pathtag_reduce", "pathtag_reduce2", "pathtag_scan1",
                                                                                "pathtag_scan_small", "pathtag_scan_large"))),
        "flatten": (scene_bytes + 20 * n_tagw + 24 * lines, stage_ms.get("flatten", 0.0)),
        "draw_reduce+leaf": (20 * n_draw + 4 * n_draw, stage_ms.get("draw_reduce", 0.0) + stage_ms.get("draw_leaf", 0.0)),
        "path_count": (24 * lines + 8 * cross, stage_ms.get("path_count", 0.0)),
        "path_tiling": ((8 + 24) * cross + 24 * segs, stage_ms.get("path_tiling", 0.0)),
        "coarse": (4 * bump["binning"] + 40 * bump["tile"] + 4 * bump["ptcl"], stage_ms.get("coarse", 0.0)),
    }
    out = {}
    for k, (b, ms) in per.items():
        if ms > 0:
            gbs = b / (ms * 1e-3) / 1e9
            out[k] = {"algorithmic_bytes": int(b), "ms": round(ms, 4), "achieved_gbs": round(gbs, 1), "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4)}
    return out


def cpu_baseline(host, args):
    """The CPU restatement of the Jello/Vello pipeline (oracle/) timed on the host cores of this box on a bounded
    sample of the same workload: whole frames of the scene the GPU just rendered, on ALL cores the process may use
    (BASELINE.md 3): fine over tile rows, path_tiling over crossings, and the three stages that allocate in canonical
    order -- flatten, path_count, coarse -- as count -> scan -> write over chunks (oracle.cpp, oracle_set_parallel_alloc;
    tests/test_oracle_parallel.py: identical buffers).  Median of up to 5 frames within ~25 s, the best thread count of a
    short sweep; plus one single-thread frame of the serial forms."""
    import numpy as np
    from jello_amd import BumpSizes, scenes
    from oracle import oracle_engine
    from oracle.oracle_engine import OracleEngine
    n, size = args.paths, args.size
    if args.scene == "c3":
        scene, params = scenes.scene_c3(n, size)
    elif args.scene == "c1":
        scene, params = scenes.scene_c1()
    elif args.scene == "c2":
        scene, params = scenes.scene_c2(n, size)
    elif args.scene == "c4":
        scene, params = scenes.scene_c4(n, size)
    else:
        scene, params = scenes.scene_c4_nested(n, size)
    import jello_amd
    params.aa = {"area": jello_amd.Aa.Area, "msaa8": jello_amd.Aa.Msaa8, "msaa16": jello_amd.Aa.Msaa16}[args.aa]
    params.bump = scene.bump_sizes(size, size)  # the estimator's sizes, as in the timed path
    rec = host.record(scene, params)
    L = oracle_engine.lib()
    cores = len(os.sched_getaffinity(0))

    def one(nt, parallel_alloc):
        L.oracle_set_threads(nt)
        L.oracle_set_parallel_alloc(1 if parallel_alloc else 0)
        orc = OracleEngine()
        t0 = time.perf_counter()
        orc.run(rec)
        dt = time.perf_counter() - t0
        if orc.get(rec, "bumpBuf", np.uint32)[0] != 0:
            raise RuntimeError("oracle bump failure in cpu_baseline")
        return dt, {k: round(v, 4) for k, v in orc.stage_seconds.items()}

    try:
        # every core the job may use is the default; a short sweep guards against a thread count that is slower on this host
        # (memory-bound stages on a many-socket box), one frame each
        candidates = sorted({cores, max(1, cores // 2), max(1, cores // 4), min(cores, 16)}, reverse=True)
        sweep = {nt: one(nt, True)[0] for nt in candidates}
        threads = min(sweep, key=sweep.get)
        runs = []
        budget = time.perf_counter() + 20.0
        while len(runs) < 5 and (not runs or time.perf_counter() + runs[-1][0] < budget):
            runs.append(one(threads, True))
        runs.sort(key=lambda r: r[0])
        dt, stages = runs[len(runs) // 2]
        one_thread = one(1, False)[0] if dt * 20 < 60 else None
    finally:
        L.oracle_set_threads(1)
        L.oracle_set_parallel_alloc(0)
    return {"value": round(size * size / dt / 1e6, 3), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "sample": "CPU restatement of the Jello/Vello pipeline (oracle/), every stage on %d threads (OpenMP: fine over tile rows, "
                      "path_tiling over crossings; flatten, path_count and coarse as count -> scan -> write over chunks of their canonical "
                      "order, buffers identical to the serial forms), %s scene with %d paths at %dx%d = the workload of this line, whole "
                      "frames incl. fine, median of %d" % (threads, args.scene.upper(), n, size, size, len(runs)),
            "seconds": round(dt, 3), "paths_per_s": round(n / dt, 1), "runs": len(runs),
            "stage_seconds": stages, "thread_sweep_seconds": {str(k): round(v, 3) for k, v in sweep.items()},
            "value_1thread": None if one_thread is None else round(size * size / one_thread / 1e6, 3),
            "seconds_1thread": None if one_thread is None else round(one_thread, 3),
            "host_cpus": os.cpu_count(), "cpus_available_to_this_process": cores}


if __name__ == "__main__":
    main()
