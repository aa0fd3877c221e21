"""Multi-GPU plumbing shared by bench.py and the gloo tests: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X, "gloo" on CPU for tests).

What shards today: *models*.  Every rank owns one listing (one KMC database / sample)
and builds + queries its own KModel; there is no data-path collective, only the timing/count reduction
below.  Queries against ONE model shard by batch (`split_batch`): every rank holds a replica of the
read-only model and answers its slice -- again no collective on the data path.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched bare."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def stream_seeds(rank: int):
    """Per-rank generator seeds: rank 0 is the single-GPU workload (seed_k=1, seed_c=2)."""
    return 1 + 1000003 * rank, 2 + 1000003 * rank


def split_batch(n: int, world: int, rank: int):
    """Contiguous slice [lo, hi) of an n-element batch for `rank`; slices differ by at most one element."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_job(seconds, units, device="cpu"):
    """Whole-job figures from per-rank ones: MAX over ranks of every time, SUM over ranks of every unit count."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(seconds), list(units)
    t = torch.tensor(list(seconds), dtype=torch.float64, device=device)
    u = torch.tensor(list(units), dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return t.tolist(), u.tolist()


def gather_slices(local: torch.Tensor, n_total: int, world: int, rank: int):
    """All-gather the per-rank answers of a batch split with `split_batch` back into batch order (ragged slices)."""
    if world == 1:
        return local
    sizes = [split_batch(n_total, world, r) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(width, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)])
