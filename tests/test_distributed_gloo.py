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
