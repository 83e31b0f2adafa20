"""CPU, world_size 2, gloo: the N>1 path of the benchmark -- one independent scene per rank, no
data-path collective, final images gathered to rank 0 -- and the tile-row band assignment."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import jello_amd
    from jello_amd import scenes, sharding
    from oracle.oracle_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # each rank renders ITS scene (here on the CPU oracle: there is no GPU in this container)
        s, p = scenes.scene_c3(200, 128, seed=sharding.scene_seed_for_rank(rank))
        rec = jello_amd.Host().record(s, p)
        o = OracleEngine()
        o.run(rec)
        local = torch.from_numpy(o.target(rec).astype(np.int32))   # gloo has no uint16
        imgs, _ = sharding.gather_images(dist, local, rank, world, dst=0)
        if rank == 0:
            q.put([im.numpy().astype(np.uint16) for im in imgs])
        else:
            q.put(local.numpy().astype(np.uint16))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_scene_per_rank_and_gather(built):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gathered = [r for r in results if isinstance(r, list)][0]
    other = [r for r in results if not isinstance(r, list)][0]
    assert len(gathered) == 2
    assert np.array_equal(gathered[1], other)            # rank 1's image arrived intact on rank 0
    assert not np.array_equal(gathered[0], gathered[1])  # the two ranks rendered different scenes


def _rotate_worker(rank, world, port, q):
    """bench.py's N > 1 step loop (sharding.GatherPipeline, the object bench.py drives) on gloo, the THREE modes one after
    the other in one process group as `bench.py --gpus N` runs them: no gather, gather to rank 0, gather to rank
    (step mod world) -- double-buffered async gathers on two communicators.  Every destination must end up with every
    rank's frame of ITS steps, and a buffer must not be rendered into before its gather is done."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from jello_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        groups = [dist.new_group(ranks=list(range(world))) for _ in range(2)]
        outs = [torch.zeros(64, dtype=torch.int32) for _ in range(2)]
        gathered = [[torch.zeros(64, dtype=torch.int32) for _ in range(world)] for _ in range(2)]
        pipe = sharding.GatherPipeline(dist, rank, world, outs, gathered, groups)
        steps = 6
        got = {}
        for mode in (None, "0", "rotate"):
            seen = {}
            rendered = []

            def render(k, _r=rendered):
                i = len(_r)
                _r.append(k)
                outs[k].fill_(1000 * {None: 1, "0": 2, "rotate": 3}[mode] + 100 * i + rank)  # "render" frame i of this rank

            def on_gathered(step, bufs, _s=seen):
                _s[step] = [int(t[0]) for t in bufs]
            for i in range(steps):
                pipe.step(i, render, mode, on_gathered)
            pipe.drain(on_gathered)
            dist.barrier()
            got[str(mode)] = (seen, rendered)
        q.put((rank, got))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_rotating_gather_destination_double_buffered(built):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rotate_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        # no gather: nothing arrives anywhere, every frame goes to buffer 0
        assert results[r]["None"] == ({}, [0] * 6)
        # both gather modes alternate the two buffers
        assert results[r]["0"][1] == [0, 1, 0, 1, 0, 1] and results[r]["rotate"][1] == [0, 1, 0, 1, 0, 1]
    # gather to rank 0: rank 0 collected every step with both ranks' frames of that step, rank 1 nothing
    assert results[0]["0"][0] == {i: [2000 + 100 * i, 2000 + 100 * i + 1] for i in range(6)}
    assert results[1]["0"][0] == {}
    # rotating destination: rank 0 collected the even steps, rank 1 the odd ones
    assert results[0]["rotate"][0] == {i: [3000 + 100 * i, 3000 + 100 * i + 1] for i in (0, 2, 4)}
    assert results[1]["rotate"][0] == {i: [3000 + 100 * i, 3000 + 100 * i + 1] for i in (1, 3, 5)}


def test_two_frames_in_flight_alternate_the_buffers_in_every_mode():
    """bench.py --in-flight 2: the gather-free loop takes the two buffers (= the two engine contexts) in turn as well; without
    `alternate` (one context) it keeps to buffer 0.  (The per-buffer HIP streams are the GPU test's business:
    test_gpu_boundary.py::test_two_contexts_in_flight_render_the_same_frame.)"""
    sys.path.insert(0, ROOT)
    from jello_amd import sharding
    for alternate, want in ((True, [0, 1, 0, 1, 0]), (False, [0, 0, 0, 0, 0])):
        pipe = sharding.GatherPipeline(None, 0, 1, [object(), object()], None, None, streams=None, alternate=alternate)
        seen = []
        for i in range(5):
            pipe.step(i, seen.append, None)
        pipe.drain()
        assert seen == want
    one = sharding.GatherPipeline(None, 0, 1, [object()], None, None, alternate=True)  # a single buffer cannot alternate
    seen = []
    for i in range(3):
        one.step(i, seen.append, None)
    assert seen == [0, 0, 0]


def test_gather_model_states_the_ceiling():
    sys.path.insert(0, ROOT)
    from jello_amd import sharding
    frame = 4096 * 4096 * 8
    g0 = sharding.gather_model(8, 1.19, frame, "0")
    gr = sharding.gather_model(8, 1.19, frame, "rotate")
    assert abs(g0["link_bound_ms"] - 1.748) < 0.01 and g0["ceiling_speedup"] < 6.0   # one compositor GPU: wire-bound below 6x
    assert gr["link_ms_per_step"] < 1.19 and gr["ceiling_speedup"] == 8.0             # rotating destinations: render-bound
    assert [sharding.gather_dst_for_step(i, 4, "rotate") for i in range(6)] == [0, 1, 2, 3, 0, 1]
    assert [sharding.gather_dst_for_step(i, 4, "0") for i in range(3)] == [0, 0, 0]


def test_band_assignment_covers_all_bin_rows():
    sys.path.insert(0, ROOT)
    from jello_amd import sharding
    for h in (1, 2, 7, 8, 16, 17):
        for w in (1, 2, 3, 4, 8):
            rows = []
            for r in range(w):
                y0, y1 = sharding.band_for_rank(h, w, r)
                rows += list(range(y0, y1))
            assert rows == list(range(h))
    assert sharding.band_for_rank(16, 8, 3) == (6, 8)


_RANK_SCRIPT = '''
import os, sys
import torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == int(sys.argv[1]) and os.environ["MASTER_ADDR"] == "127.0.0.1"
open(os.path.join(sys.argv[2], "rank%d" % rank), "w").write("ok")
dist.barrier()
dist.destroy_process_group()
sys.exit(int(sys.argv[3]) if rank == world - 1 else 0)
'''


def test_launcher_starts_one_process_per_rank_and_propagates_failure(tmp_path):
    """`bench.py --gpus N` outside torchrun goes through sharding.launch_ranks: N child ranks, exit code of the job."""
    sys.path.insert(0, ROOT)
    from jello_amd import sharding
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    assert sharding.launch_ranks(str(script), ["2", str(tmp_path), "0"], 2, timeout=300) == 0
    assert (tmp_path / "rank0").exists() and (tmp_path / "rank1").exists()
    assert sharding.launch_ranks(str(script), ["2", str(tmp_path), "3"], 2, timeout=300) != 0


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_bench_gpus_2_without_gpus_fails_loudly():
    """No GPU here (and one on the GPU box): the 2-rank job must exit non-zero, not print a 1-GPU line."""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:  # (counting devices does not initialise the GPU)
        pytest.skip("this node could really run two ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert '"n_gpus"' not in r.stdout


def _failing_mode_worker(rank, world, port, q):
    """bench.py's mode loop (sharding.time_modes_surviving_failures) on gloo: mode "0" raises on rank 1 only (before any
    collective of the mode), mode "rotate" works.  Both ranks must drop "0", keep "rotate", and stay in step."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from jello_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        groups = [dist.new_group(ranks=list(range(world))) for _ in range(2)]
        outs = [torch.zeros(8, dtype=torch.int32) for _ in range(2)]
        gathered = [[torch.zeros(8, dtype=torch.int32) for _ in range(world)] for _ in range(2)]

        def timed(mode):
            if mode == "0" and rank == 1:
                raise RuntimeError("simulated RCCL failure")
            if mode == "0":
                return [1.0]  # (rank 0 believes the mode went fine: the agreement must overrule it)
            pipe = sharding.GatherPipeline(dist, rank, world, outs, gathered, groups)
            for i in range(4):
                pipe.step(i, lambda k: outs[k].fill_(10 * i + rank), mode)
            pipe.drain()
            return [2.0, 3.0]
        blocks, errors = sharding.time_modes_surviving_failures(dist, rank, ["0", "rotate"], timed, torch.device("cpu"))
        q.put((rank, blocks, errors))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_a_failing_gather_mode_is_reported_and_the_others_survive(built):
    """a gather mode that fails on first contact with hardware must not cost the line."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_mode_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (b, e) for r, b, e in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        blocks, errors = res[r]
        assert list(blocks) == ["rotate"] and blocks["rotate"] == [2.0, 3.0]
        assert list(errors) == ["0"]
    assert "simulated RCCL failure" in res[1][1]["0"] and res[0][1]["0"] == "failed on another rank"


def test_headline_mode_of_a_multi_gpu_line():
    """bench.py's `value` for N > 1: the rotating gather when it ran, else the gather-free loop; gather-to-rank-0 only when asked for."""
    from jello_amd import sharding
    assert sharding.headline_mode("all", ["0", "rotate"]) == "rotate"
    assert sharding.headline_mode("all", ["0"]) is None          # (the rotating gather died: the line falls back to no gather)
    assert sharding.headline_mode("all", []) is None
    assert sharding.headline_mode("0", ["0"]) == "0"
    assert sharding.headline_mode("rotate", ["rotate"]) == "rotate"
    m8 = sharding.gather_model(8, 0.81, 4096 * 4096 * 8, "0")
    assert m8["ceiling_speedup"] < 6.0 < sharding.gather_model(8, 0.81, 4096 * 4096 * 8, "rotate")["ceiling_speedup"]
