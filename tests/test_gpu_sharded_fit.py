"""GPU, world size 2 over gloo on ONE device: the N > 1 path end to end with the REAL fit per rank - contiguous frame
shards (mvsmplfitting_amd.sharding), every rank fits its own frames on the GPU, one final all_gather - and the gathered
result is bit-identical to one process fitting all frames (problems never interact)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
TOTAL = 6


def _inputs():
    from mvsmplfitting_amd import synthetic as syn
    from tests.helpers import GOLD, body_model
    g = dict(np.load(os.path.join(GOLD, 'fit_l2.npz')))
    cams = (g['cam_R'], g['cam_t'], g['cam_f'], g['cam_c'])
    rng = np.random.default_rng(21)
    gt = np.repeat(g['gt_xy'][:1], TOTAL, 0) + rng.normal(0, 3.0, (TOTAL,) + g['gt_xy'].shape[1:]).astype(np.float32)
    conf = np.repeat(g['conf'][:1], TOTAL, 0)
    x0 = np.zeros((TOTAL, 118), np.float32)
    x0[:, 85] = 1.0
    x0[:, :86] += rng.normal(0, 0.02, (TOTAL, 86)).astype(np.float32)
    return body_model(0, 4), cams, gt, conf, x0


def _fit(lo, hi):
    from mvsmplfitting_amd.engine import MvFit, stage_weights
    model, cams, gt, conf, x0 = _inputs()
    eng = MvFit(model)
    eng.set_problems(cams, gt[lo:hi], conf[lo:hi])
    xf, st = eng.fit(x0[lo:hi], stage_weights(1536.0, flags=0))
    out = torch.cat([xf, st['final_loss'][:, None], st['n_closure'][:, None].float()], 1)
    eng.close()
    return out


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from mvsmplfitting_amd.sharding import gather_results, shard_range
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = shard_range(TOTAL, world, rank)
    local = _fit(lo, hi).cpu()                       # gloo gathers host tensors; with backend nccl the rows stay on the GPU
    full = gather_results(local, TOTAL)
    q.put((rank, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_fit_their_shards_and_gather_the_single_rank_result():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    single = _fit(0, TOTAL).cpu().numpy()
    assert got[0].shape == (TOTAL, 120)
    assert np.array_equal(got[0], got[1])
    assert np.array_equal(got[0], single)
