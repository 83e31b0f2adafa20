#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: "Mpixels/sec fine-raster + paths/sec, 100k-path
4096^2 scene, 1/2/4/8 MI355X".

A step = one full pass of the hot path (pathtag scan -> flatten -> draw/clip scans -> binning ->
tile_alloc -> path_count -> backdrop -> coarse -> path_tiling -> fine) over one synthetic scene that is
already resident in HBM (scene bytes + config uploaded before the timed region).  At N > 1 every rank
renders its own independent scene (weak scaling, no data-path collective); the finished RGBA16F image
stays in that rank's HBM, as the reference leaves it in a texture of the device that rendered it.
`--gather` additionally collects the images on rank 0 over RCCL, overlapped with the next frame (a
compositor on one GPU; 7 x 128 MiB per frame into one device is then the bound, not the renderer).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--paths", type=int, default=100_000)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--aa", choices=["area", "msaa8", "msaa16"], default="area", help="coverage mode of the fine stage (the headline is area)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true", help="N > 1: also gather every rank's image to rank 0 (RCCL, asynchronous)")
    ap.add_argument("--bands", action="store_true",
                    help="N > 1: ONE scene, every rank runs the element stages on it and coarse+fine for its band of bin rows "
                         "(strong scaling of one frame; SURVEY 8e) instead of one independent scene per rank")
    ap.add_argument("--emulate-band-of", type=int, default=0, metavar="N",
                    help="with --bands on ONE GPU: time what rank N/2 of an N-GPU band job would do (its band of the frame)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph")
    args = ap.parse_args()

    import torch
    import jello_amd
    from jello_amd import BumpSizes, scenes
    from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    W = H = args.size
    # independent scene per rank (same generator, different seed)
    scene, params = scenes.scene_c3(args.paths, args.size, seed=scenes.SEED + (0 if args.bands else rank))
    params.aa = {"area": jello_amd.Aa.Area, "msaa8": jello_amd.Aa.Msaa8, "msaa16": jello_amd.Aa.Msaa16}[args.aa]
    fine_stage = {"area": "fine_area", "msaa8": "fine_msaa8", "msaa16": "fine_msaa16"}[args.aa]
    eng = jello_amd.Engine(dev.index)
    host = jello_amd.Host()
    stream = torch.cuda.current_stream(dev)
    eng.set_stream(stream.cuda_stream)

    # ---- size the bump buffers once (regrow loop), outside the timed region ----
    params.bump = BumpSizes(lines=1 << 22, seg_counts=1 << 23, segments=1 << 23, tiles=1 << 22, ptcl=1 << 26, bin_data=1 << 21)
    rec0, bump, attempts = eng.render(scene, params, robust=True)
    if bump["failed"]:
        raise RuntimeError("bump allocation still failing after regrow: %s" % bump)
    cfg0 = rec0.config
    margin = lambda x: int(x * 1.1) + 4096
    params.bump = BumpSizes(lines=margin(bump["lines"]), seg_counts=margin(bump["seg_counts"]), segments=margin(bump["segments"]),
                            tiles=margin(bump["tile"]), ptcl=margin(bump["ptcl"] + cfg0["width_in_tiles"] * cfg0["height_in_tiles"] * 64),
                            bin_data=margin(bump["binning"] + cfg0["bin_data_start"]), blend_spill=max(4096, margin(bump["blend"])))
    del rec0
    rec = host.record(scene, params)
    cfg = rec.config

    # output images: torch owns the device memory (double-buffered for the overlapped gather)
    outs = [torch.empty((H, W, 4), dtype=torch.float16, device=dev) for _ in range(2)]
    gathered = None
    if world > 1 and rank == 0 and args.gather:
        gathered = [[torch.empty((H, W, 4), dtype=torch.float16, device=dev) for _ in range(world)] for _ in range(2)]

    if args.bands:  # (buffers were sized by the unsharded render above; from here on this rank owns its band only)
        from jello_amd import sharding
        if world == 1 and args.emulate_band_of > 1:
            eng.set_band(*sharding.band_for_rank((cfg["height_in_tiles"] + 15) // 16, args.emulate_band_of, args.emulate_band_of // 2))
        else:
            eng.set_band(*sharding.band_for_rank((cfg["height_in_tiles"] + 15) // 16, world, rank))
    eng.run(rec, RUN_UPLOADS | RUN_DISPATCHES, outs[0].data_ptr())  # uploads scene/config; allocates every buffer
    torch.cuda.synchronize(dev)

    pending = [None, None]
    # One frame is ~60 short launches (launch-bound on the host), so the dispatch-only replay of the
    # recording is captured once into a hipGraph per output buffer and the timed steps replay it.
    graphs = [None, None]
    use_graph = not args.no_graph
    if use_graph:
        for k in range(2 if world > 1 and args.gather else 1):
            graphs[k] = eng.capture(rec, outs[k].data_ptr())
        if graphs[1] is None:
            graphs[1] = graphs[0]
            outs[1] = outs[0]

    def step(i):
        k = i & 1
        if pending[k] is not None:
            pending[k].wait()
            pending[k] = None
        if use_graph:
            eng.replay(graphs[k])
        else:
            eng.run(rec, RUN_DISPATCHES, outs[k].data_ptr())
        if world > 1 and args.gather:
            pending[k] = dist.gather(outs[k], gathered[k] if rank == 0 else None, dst=0, async_op=True)

    def drain():
        for j in range(2):
            if pending[j] is not None:
                pending[j].wait()
                pending[j] = None

    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)

    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()

    # Per-stage device times: the same K steps once more, eagerly, with a hipEvent pair around every
    # stage on the launch stream (events cannot be read back from inside a replayed graph).
    eng.profile(True)
    for i in range(args.steps):
        eng.run(rec, RUN_DISPATCHES, outs[0].data_ptr())
    torch.cuda.synchronize(dev)
    prof = eng.profile_collect(1 << 16)
    eng.profile(False)

    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed * 1e3 / args.steps

    # ---- per-stage device times (hipEvent pairs recorded on the launch stream during the timed region) ----
    stage_ms = {}
    for name, ms in prof:
        stage_ms[name] = stage_ms.get(name, 0.0) + ms
    stage_ms = {k: v / args.steps for k, v in stage_ms.items()}
    fine_ms = stage_ms.get(fine_stage, float("nan"))

    result = None
    if rank == 0:
        # ---- algorithmic bytes of the fine stage, from the PTCL this run produced ----
        ptcl_id, ptcl_size = rec.buffer("ptclBuf")
        ptcl = eng.download(ptcl_id, dtype=np.uint32)
        st = (ctypes.c_uint64 * 8)()
        rc = eng._L.jl_ptcl_stats(ptcl.ctypes.data, ptcl.size, cfg["width_in_tiles"], cfg["height_in_tiles"], st)
        if rc != 0:
            raise RuntimeError("malformed PTCL")
        words, segs, info_words, texels, spill_px = st[0], st[1], st[2], st[3], st[4]
        b_fine = 4 * words + 24 * segs + 4 * info_words + 8 * texels + 32 * spill_px + 8 * W * H
        achieved = b_fine / (fine_ms * 1e-3) / 1e9
        bump_now = eng.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8]
        # achievable HBM bandwidth on this device: a 1 GiB device-to-device copy (read + write bytes), reported
        # next to the 8 TB/s vendor figure that `peak` uses (SURVEY 8d asks for both)
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
            copy_gbs = 5 * 2 * (1 << 30) / (ce0.elapsed_time(ce1) * 1e-3) / 1e9
            del ca, cb
        except Exception:  # noqa: BLE001 - the measurement is optional
            copy_gbs = None
        roofline = {"kernel": "k_fine_area" if args.aa == "area" else "k_fine_area<%s>" % args.aa, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes": int(b_fine), "avg_ms": round(fine_ms, 4),
                    "segment_pixel_evals": int(segs) * 256,
                    "peak_measured_copy": None if copy_gbs is None else round(copy_gbs, 1)}
        pm = os.path.join(ROOT, "profiles", "fine_traffic.json")
        if os.path.exists(pm):
            try:
                roofline["traffic"] = json.load(open(pm)).get("hbm_bytes_per_launch")
            except Exception:
                pass
        if args.bands and (world > 1 or args.emulate_band_of > 1):  # one band only: the whole-target byte count does not apply
            roofline = None
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(host)
        units = 1 if args.bands else world  # frames finished per step by the whole job
        mpix = W * H * units / (elapsed / args.steps) / 1e6
        result = {
            "metric": "Mpixels/sec fine-raster + paths/sec, 100k-path 4096^2 scene", "value": round(mpix, 2), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if args.bands else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3: %d random stroked+filled cubic Beziers, %dx%d RGBA16F target, %s AA, one independent scene per GPU"
                                   % (args.paths, W, H, args.aa),
                       "paths": args.paths, "draw_objects": cfg["n_drawobj"], "width": W, "height": H,
                       "parallelism": ("bin-row bands of one scene x%d" % world) if args.bands else
                                      "scene-per-gpu x%d%s" % (world, "" if world == 1 or not args.gather else " + RCCL image gather")},
            "paths_per_s": round(args.paths * units / (elapsed / args.steps), 1),
            "fine_mpixels_per_s": round(W * H / (fine_ms * 1e-3) / 1e6, 2),
            "launch": "hipGraph replay" if use_graph else "eager",
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_source": "eager replay of the same %d steps with a hipEvent pair per stage" % args.steps,
            "bump": {k: int(v) for k, v in zip(["failed", "binning", "ptcl", "tile", "seg_counts", "segments", "blend", "lines"], bump_now)},
            "stage_roofline": stage_roofline(cfg, dict(zip(["failed", "binning", "ptcl", "tile", "seg_counts", "segments", "blend", "lines"],
                                                           [int(v) for v in bump_now])), stage_ms, rec),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "device": eng.device_info()["name"],
        }
    for g in set(id(x) for x in graphs if x is not None):
        pass
    if graphs[0] is not None:
        eng.graph_destroy(graphs[0])
        if graphs[1] is not graphs[0]:
            eng.graph_destroy(graphs[1])
    eng.release(rec)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


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


def cpu_baseline(host):
    """The CPU restatement of the Jello/Vello pipeline (oracle/, single thread) timed on a bounded
    sample of the same workload: one frame of the full headline scene (about 5-10 s of CPU work)."""
    from jello_amd import BumpSizes, scenes
    from oracle.oracle_engine import OracleEngine
    n, size = 100_000, 4096
    scene, params = scenes.scene_c3(n, size)
    params.bump = BumpSizes(lines=1 << 22, seg_counts=1 << 23, segments=1 << 23, tiles=1 << 21, ptcl=1 << 25, bin_data=1 << 20)
    rec = host.record(scene, params)
    from oracle import oracle_engine
    # fine (79 % of the single-thread time) runs tile rows on `threads` host threads; the element stages stay serial
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    res = {}
    for nt in ([1, threads] if threads > 1 else [1]):
        oracle_engine.lib().oracle_set_threads(nt)
        orc = OracleEngine()
        t0 = time.perf_counter()
        orc.run(rec)
        dt = time.perf_counter() - t0
        bump = orc.get(rec, "bumpBuf", np.uint32)[:8]
        if bump[0] != 0:
            raise RuntimeError("oracle bump failure in cpu_baseline")
        res[nt] = (dt, {k: round(v, 4) for k, v in orc.stage_seconds.items()})
    oracle_engine.lib().oracle_set_threads(1)
    dt, stages = res[threads]
    return {"value": round(size * size / dt / 1e6, 3), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "sample": "CPU restatement of the Jello/Vello pipeline (oracle/; fine on %d threads over tile rows, element stages serial), "
                      "C3 generator with %d paths at %dx%d = the full headline workload, one frame, all stages incl. fine" % (threads, n, size, size),
            "seconds": round(dt, 3), "paths_per_s": round(n / dt, 1),
            "stage_seconds": stages,
            "value_1thread": round(size * size / res[1][0] / 1e6, 3), "seconds_1thread": round(res[1][0], 3),
            "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    main()
