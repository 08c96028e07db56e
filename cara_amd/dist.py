"""Data-parallel plumbing (build-own: the reference has no distributed code at all -- its only
trace is a commented-out ``DataParallel`` at image_classification/dim_experiment.py:419).

Design for xGMI: samples are independent and the trainable state is tiny (2526*R + 4608 CP
values + the head = 121 924 floats at R=16 / 100 classes), so each rank holds a full replica,
processes its own 64-image shard and the ONLY data-path collective of a step is one all-reduce of
one flat fp32 buffer (487 696 B).  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def flat_views(shapes: Sequence[Tuple[str, torch.Size]], device, dtype=torch.float32):
    """One flat buffer + named views into it, in the given order."""
    sizes = [int(torch.Size(s).numel()) for _, s in shapes]
    flat = torch.zeros(sum(sizes), device=device, dtype=dtype)
    views: Dict[str, torch.Tensor] = {}
    off = 0
    for (n, s), sz in zip(shapes, sizes):
        views[n] = flat[off:off + sz].view(s)
        off += sz
    return flat, views


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def get_rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def allreduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place mean over ranks of the flat gradient buffer: a single collective per step."""
    ws = world_size(group)
    if ws > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / ws)
    return flat


def allreduce_sum_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM over ranks: the collective of a train step whose dlogits was divided by the world size at its source
    (cara_cross_entropy_ex), so that no scaling launch follows the all-reduce."""
    if world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def epoch_shard(n_items: int, epoch: int, rank: int, world: int, per_rank_batch: int, seed: int = 0) -> List[torch.Tensor]:
    """Rank-strided slices of an epoch-seeded permutation, drop_last like vtab.py:84-88:
    every rank draws the same permutation, takes items rank, rank+world, ... and cuts them into
    batches of ``per_rank_batch``; the global batch is world * per_rank_batch."""
    g = torch.Generator().manual_seed(seed * 1_000_003 + epoch)
    perm = torch.randperm(n_items, generator=g)
    mine = perm[rank::world]
    nb = min(len(perm[r::world]) for r in range(world)) // per_rank_batch
    return [mine[i * per_rank_batch:(i + 1) * per_rank_batch] for i in range(nb)]


def broadcast_parameters(params: Sequence[torch.Tensor], src: int = 0, group=None) -> None:
    """Make replicas identical once at start-up (afterwards identical gradients + identical
    optimiser state keep them identical; no per-step parameter traffic)."""
    if world_size(group) > 1:
        for p in params:
            dist.broadcast(p.data, src=src, group=group)
