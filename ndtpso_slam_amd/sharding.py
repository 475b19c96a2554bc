"""Multi-GPU layout of a batch of independent scan pairs.

Alignments are independent, so the batch shards by contiguous index range, one process per GPU,
with no data-path collective; the only exchange is one all_gather of the poses (RCCL over xGMI on
the GPUs, gloo in the CPU tests).  Backend-agnostic: works on whatever process group is initialised.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """[first, last) pair indices of `rank` -- contiguous, sizes differing by at most one."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def gather_poses(local: torch.Tensor, group=None, equal_sizes: bool = False, force: bool = False) -> torch.Tensor:
    """all_gather of per-rank [n_i, 3] pose tensors -> [sum n_i, 3] in pair order (every rank gets it).
    equal_sizes=True skips the size exchange (every rank holds the same number of pairs): ONE collective.
    force=True issues the collective even in a one-rank group (exercises the backend)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return local
    world = dist.get_world_size(group)
    if equal_sizes:
        out = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(out, local.contiguous(), group=group)
        return torch.cat(out, dim=0)
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
    sizes = [int(s.item()) for s in sizes]
    if len(set(sizes)) == 1:
        out = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(out, local.contiguous(), group=group)
        return torch.cat(out, dim=0)
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)
