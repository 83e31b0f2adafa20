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
