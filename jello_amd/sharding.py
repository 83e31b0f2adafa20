"""Multi-GPU sharding of the render path.

Two modes, neither needs a collective on the data path:
  * scenes: independent scenes, one per rank (weak scaling) -- `scene_seed_for_rank`;
  * bands:  disjoint bin-row bands of ONE target -- every rank runs the cheap element stages on the
            whole scene (replicated, so allocation scans and hence PTCL addresses are identical on
            every rank) and coarse's write pass + fine only for its band -- `band_for_rank` + `Engine.set_band`.
The only (optional: `bench.py --gather`) exchange is the final image gather to rank 0 (`gather_images`): RCCL on GPUs ("nccl"
backend), gloo in the CPU tests.
"""
import contextlib
import os
import socket
import subprocess
import sys

from . import scenes


def scene_seed_for_rank(rank, base=scenes.SEED):
    return base + int(rank)


def band_for_rank(height_in_bins, world, rank):
    """Contiguous range [y0, y1) of 256-px bin rows owned by `rank`; earlier ranks take the remainder."""
    base, rem = divmod(int(height_in_bins), int(world))
    y0 = rank * base + min(rank, rem)
    y1 = y0 + base + (1 if rank < rem else 0)
    return y0, y1


def gather_dst_for_step(step, world, mode="0"):
    """Destination rank of the image gather of frame `step`: "0" = always rank 0 (one compositor GPU, C5 as written),
    "rotate" = rank (step mod world): consecutive gathers then arrive over disjoint inbound xGMI links."""
    return int(step) % int(world) if mode == "rotate" else 0


def gather_images(dist, local, rank, world, dst=0, async_op=False, out=None, group=None):
    """Gather every rank's finished image tensor to `dst`.  Returns (list of tensors on dst | None, work)."""
    import torch
    if world == 1:
        return [local], None
    if rank == dst and out is None:
        out = [torch.empty_like(local) for _ in range(world)]
    work = dist.gather(local, out if rank == dst else None, dst=dst, async_op=async_op, group=group)
    return (out if rank == dst else None), work


class GatherPipeline:
    """The step loop of bench.py (and of the gloo tests): render frame i into buffer i & 1, then hand it to an
    asynchronous gather whose destination depends on the mode -- None: no gather (the frame stays on the GPU that rendered
    it), "0": rank 0, "rotate": rank (i mod N).  Each buffer of the double buffer has its own communicator (two gathers on
    ONE communicator run one after the other), so the gather of frame i overlaps the render of frame i + 1.  One object
    runs all three modes one after the other (`drain()` in between).

    Two frames in flight (round 4): with `streams` = one HIP stream per buffer (torch.cuda.Stream objects; the caller renders
    buffer k with an engine of its own that launches on streams[k]) everything that concerns buffer k -- waiting for its previous
    gather, the render, the next gather -- is ordered on streams[k] alone, so frame i + 1 does not wait for frame i at all and
    the two contexts' kernels fill each other's launch gaps and tails.  `alternate` makes the gather-free mode use both buffers
    too (without it mode None renders into buffer 0 only, as before)."""

    def __init__(self, dist, rank, world, outs, gathered=None, groups=None, streams=None, alternate=False):
        self.dist, self.rank, self.world, self.outs = dist, rank, world, outs
        self.gathered = gathered      # [buffer][source rank] on every rank that can be a destination, else None
        self.groups = groups or [None, None]
        self.pending = [None, None]   # (work, step, destination) of the gather that last read outs[k]
        self.streams = streams
        self.alternate = bool(alternate) and len(outs) >= 2

    def _on(self, k):
        if self.streams is None:
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(self.streams[k])

    def _finish(self, k, on_gathered):
        if self.pending[k] is None:
            return
        work, step, dst = self.pending[k]
        work.wait()  # (stream-ordered on RCCL: the host does not block; the CURRENT stream waits -- streams[k] under _on(k))
        if on_gathered is not None and dst == self.rank:
            on_gathered(step, self.gathered[k])
        self.pending[k] = None

    def step(self, i, render, mode, on_gathered=None):
        """render(k) must leave frame i in outs[k].  With a gather mode the buffer is not reused before its gather is done."""
        k = (i & 1) if (mode is not None or self.alternate) else 0
        with self._on(k):
            self._finish(k, on_gathered)
            render(k)
            if mode is None or self.world == 1:
                return
            dst = gather_dst_for_step(i, self.world, mode)
            if dst == self.rank and self.gathered is None:
                raise RuntimeError("rank %d is the destination of step %d but has no receive buffers" % (self.rank, i))
            _, work = gather_images(self.dist, self.outs[k], self.rank, self.world, dst=dst, async_op=True,
                                    out=self.gathered[k] if dst == self.rank else None, group=self.groups[k])
            self.pending[k] = (work, i, dst)

    def drain(self, on_gathered=None):
        for k in range(2):
            with self._on(k):
                self._finish(k, on_gathered)


def time_modes_surviving_failures(dist, rank, modes, timed, device, already_failed=None):
    """Time each gather mode on its own (`timed(mode)` -> block times).  A mode that raises on ANY rank is dropped on EVERY
    rank -- the ranks agree through a MIN all-reduce of one flag on the default group -- and reported in the returned error
    dict; the modes before and behind it are still timed.  If the agreement itself fails the process group is taken to be
    unusable and the remaining modes are skipped.  Returns (blocks_by_mode, errors)."""
    import torch
    blocks, errors = {}, dict(already_failed or {})
    comm_ok = True
    for m in modes:
        if m in errors:
            continue
        if not comm_ok:
            errors[m] = "skipped: an earlier mode left the process group unusable"
            continue
        ok = 1
        try:
            blocks[m] = timed(m)
        except Exception as e:  # noqa: BLE001 - whatever RCCL or the pipeline raises must not cost the other modes
            ok = 0
            errors[m] = "rank %d: %s: %s" % (rank, type(e).__name__, str(e)[:300])
        try:
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                blocks.pop(m, None)
                errors.setdefault(m, "failed on another rank")
        except Exception as e:  # noqa: BLE001
            comm_ok = False
            blocks.pop(m, None)
            errors.setdefault(m, "rank %d: agreement all-reduce failed: %s" % (rank, type(e).__name__))
    return blocks, errors


XGMI_ONE_WAY_GBS = 76.8  # one xGMI link of an MI355X: 153.6 GB/s bidirectional; every GPU pair of a node has its own link


def gather_model(world, render_ms, frame_bytes, mode="0", in_flight=2):
    """What the wire allows for the image gather, as numbers (the job cannot be faster than this whatever the renderer does).

    Every source GPU reaches the destination over its own point-to-point link, so one gather takes
    frame_bytes / 76.8 GB/s however many sources there are.  dst = 0: the gathers of consecutive frames all end on rank 0's
    inbound links and run one after the other: a step takes max(render, link) and the N-GPU job is at best
    N * render / max(render, link) times one GPU.  dst = rotate: consecutive frames go to different destinations over
    disjoint links, so the `in_flight` (= the double buffer's two) gathers that are under way at a time proceed together:
    a step takes max(render, link / in_flight)."""
    link_ms = frame_bytes / (XGMI_ONE_WAY_GBS * 1e9) * 1e3
    eff = link_ms / (in_flight if mode == "rotate" else 1)
    step_ms = max(render_ms, eff)
    return {"bytes_per_rank_and_frame": int(frame_bytes), "into": "rank (step mod N)" if mode == "rotate" else "rank 0",
            "link_bound_ms": round(link_ms, 3), "link_ms_per_step": round(eff, 3), "render_ms": round(render_ms, 4),
            "ceiling_speedup": round(world * render_ms / step_ms, 2),
            "ceiling_speedup_note": "N * render / max(render, link time per step): what %d GPUs can reach over one GPU with this "
                                    "gather, from the wire alone" % world,
            "note": "each source GPU reaches the destination over its own xGMI link (153.6 GB/s bidirectional = 76.8 GB/s one way)"}


def assemble_bands(images, height_in_bins, world):
    """Stitch per-rank band images (each full-size, only its band valid) into one image (on dst)."""
    import torch
    out = torch.zeros_like(images[0])
    for r, im in enumerate(images):
        y0, y1 = band_for_rank(height_in_bins, world, r)
        out[y0 * 256:y1 * 256] = im[y0 * 256:y1 * 256]
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(script, script_args, n, env=None, timeout=None):
    """Start `n` ranks of `script` on this node (one process per GPU, the driver's own recipe:
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 ...`) as CHILD processes
    and return the launcher's exit code (non-zero when any rank failed).  Must be called before the calling process has
    touched the GPU (nothing here imports torch): a process that has initialised HIP must never exec or fork GPU work."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(script_args)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes needs it on this image)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run(cmd, env=e, timeout=timeout).returncode


def headline_mode(gather_dst, good_modes):
    """Which gather mode a multi-GPU bench line reports as `value`: with `--gather-dst all` the gather whose destination rotates when
    it ran (a gather to ONE rank is capped by that rank's inbound links -- gather_model -- whatever the renderer does), else none (the
    gather-free loop); an explicit `--gather-dst 0` / `rotate` is reported as asked.  None = the gather-free loop."""
    if not good_modes:
        return None
    if gather_dst == "all":
        return "rotate" if "rotate" in good_modes else None
    return good_modes[0]
