"""Frame sharding across the GPUs of one node.

Problems (subject x frame) are independent: the reference has no coupling between frames when
``is_seq`` is off (code/main.py:32-89 processes them one by one) and never batches persons
(code/utils/non_linear_solver.py:56).  So the path shards with NO data-path collective: rank r fits
the contiguous block ``shard_range(B, world, r)`` on its own GPU, model constants are replicated, and
the only exchange is the final gather of the fitted parameters / losses (RCCL over xGMI when the
backend is "nccl"; the same code runs on "gloo" for the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(num_problems: int, world_size: int, rank: int):
    """Contiguous blocks of ceil(B / world) problems (keeps a subject's frames together); trailing
    ranks may get fewer (or zero) problems."""
    per = -(-num_problems // world_size)
    lo = min(num_problems, rank * per)
    return lo, min(num_problems, lo + per)


def gather_results(local: torch.Tensor, num_problems: int, group=None) -> torch.Tensor:
    """All-gather of per-problem result rows [n_local, ...] -> [num_problems, ...] on every rank.
    Ragged shards are padded to the common block size for the collective and trimmed afterwards."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    per = -(-num_problems // world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat(out, 0)[:num_problems]
