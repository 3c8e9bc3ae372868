"""CPU, world_size 2 over gloo: the N > 1 path of bench.py / the fit driver - contiguous frame shards,
no data-path collective, one final gather - gives every rank the full, correctly ordered result."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvsmplfitting_amd.sharding import gather_results, shard_range


def test_shard_range_partitions():
    for B in (1, 7, 32, 33, 1024):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                lo, hi = shard_range(B, world, r)
                assert 0 <= lo <= hi <= B
                covered += list(range(lo, hi))
            assert covered == list(range(B))


def _worker(rank, world, port, B, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = shard_range(B, world, rank)
    # stand-in for the per-rank fit: row p of the result is a function of the global problem index p
    local = torch.stack([torch.arange(5, dtype=torch.float32) + 10.0 * p for p in range(lo, hi)]) \
        if hi > lo else torch.zeros(0, 5)
    full = gather_results(local, B)
    q.put((rank, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_over_gloo():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    for B in (7, 32):                                    # ragged and even shards
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
        for p in procs:
            p.start()
        got = dict(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        expect = np.stack([np.arange(5, dtype=np.float32) + 10.0 * p for p in range(B)])
        for r in range(2):
            assert np.array_equal(got[r], expect)
