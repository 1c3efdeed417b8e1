"""Multi-GPU layout of the encode path: one process per GPU, sequences sharded over ranks, weights replicated.

The reference reaches multi-GPU only through Lightning DDP (main.py:388-395: one task per GPU); at inference /
evaluation time its sequences are independent, so the MI355X build shards `(subject, segment)` sequences with no
collective on the data path and offers the two exchange steps the evaluation needs:
all-gather of predictions and all-reduce of the per-voxel Pearson statistics.  Backend "nccl" is RCCL over xGMI on
ROCm; "gloo" runs the same code on CPU for tests.
"""

from __future__ import annotations

import typing as tp

import torch
import torch.distributed as dist


def world() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n_sequences: int, rank: int, world_size: int) -> list[int]:
    """Rank r owns sequences r, r + G, r + 2G, ... ('rank r of G takes sequences r::G')."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, n_sequences, world_size))


def shard_batch(data: dict[str, torch.Tensor], rank: int, world_size: int) -> dict[str, torch.Tensor]:
    """Slice every tensor of a SegmentData.data dict along the batch axis for this rank."""
    n = next(iter(data.values())).shape[0]
    idx = torch.tensor(shard_indices(n, rank, world_size), dtype=torch.long)
    return {k: v.index_select(0, idx.to(v.device)) for k, v in data.items()}


def gather_predictions(pred: torch.Tensor, out: torch.Tensor | None = None, async_op: bool = False,
                       group: tp.Any = None) -> tuple[torch.Tensor, tp.Any]:
    """All-gather equally sized [b, V, T'] shards into [G*b, V, T'] (rank-major).  Returns (buffer, work|None)."""
    rank, ws = world()
    if ws == 1:
        return pred, None
    if out is None:
        out = torch.empty((ws * pred.shape[0],) + tuple(pred.shape[1:]), dtype=pred.dtype, device=pred.device)
    if dist.get_backend(group) == "gloo" and pred.is_cuda:
        # rehearsal path (several ranks on one GPU): gloo has no device-side all_gather_into_tensor
        chunks = list(out.chunk(ws, dim=0))
        work = dist.all_gather(chunks, pred.contiguous(), group=group, async_op=async_op)
        return out, work
    work = dist.all_gather_into_tensor(out, pred.contiguous(), group=group, async_op=async_op)
    return out, work


def unshard_order(n_sequences: int, world_size: int) -> list[int]:
    """Permutation that maps the rank-major gather buffer back to original sequence order (equal shards)."""
    order = [i for r in range(world_size) for i in shard_indices(n_sequences, r, world_size)]
    inv = [0] * n_sequences
    for pos, i in enumerate(order):
        inv[i] = pos
    return inv


def allreduce_stats(stats: torch.Tensor, group: tp.Any = None) -> torch.Tensor:
    """Sum the f64 [G, V, 6] Pearson sufficient statistics over ranks (in place)."""
    _, ws = world()
    if ws > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats
