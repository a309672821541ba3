"""Ensemble-member sharding: one process per GPU, members split over ranks, ONE all-gather at the end.

The reference's eval has this shape -- compute locally, then a single ``dist.all_gather`` of the
predictions (/root/reference/src/gwen/models_gnn.py:471; the rank-id all-gather at :470 and the
sort at :475-480 only recover an order that ``all_gather_into_tensor`` already guarantees, and the
barriers at :469,:482,:487,:491 are redundant).  Members are independent (replicated graph and
weights, no gradient exchange), so there is no collective on the data path: the only exchange is the
final gather of ``[members_local, N, C_out]`` fp32, issued on RCCL over xGMI when the process group
backend is "nccl" and on gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def member_range(num_members: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block of members owned by ``rank``: sizes differ by at most one, rank order =
    member order, so a rank-ordered all-gather returns members in their global order."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    if num_members < 0:
        raise ValueError("num_members must be >= 0")
    base, extra = divmod(num_members, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def member_counts(num_members: int, world_size: int) -> List[int]:
    return [member_range(num_members, r, world_size)[1] - member_range(num_members, r, world_size)[0]
            for r in range(world_size)]


def gather_members(local: Tensor, num_members: int, group: Optional[dist.ProcessGroup] = None) -> Tensor:
    """All-gather ``local`` ``[members_local, ...]`` into ``[num_members, ...]`` on every rank.

    One collective.  Equal shards use ``all_gather_into_tensor`` directly into the result; ragged
    shards (num_members not divisible by the world size) pad to the largest shard, gather once and
    drop the padding.
    """
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        if local.size(0) != num_members:
            raise ValueError("single process must hold every member")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = member_counts(num_members, world)
    if local.size(0) != counts[rank]:
        raise ValueError(f"rank {rank} holds {local.size(0)} members, expected {counts[rank]}")
    local = local.contiguous()
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the N > 1 path without RCCL (bench.py --rehearse-gloo: several ranks on one GPU): gloo
        # gathers host tensors, so the shards go through the host -- never the measured configuration
        return gather_members(local.cpu(), num_members, group).to(local.device)
    tail = tuple(local.shape[1:])
    if len(set(counts)) == 1:
        out = torch.empty((num_members,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    cmax = max(counts)
    padded = torch.zeros((cmax,) + tail, dtype=local.dtype, device=local.device)
    padded[: local.size(0)] = local
    buf = torch.empty((world * cmax,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    buf = buf.view((world, cmax) + tail)
    return torch.cat([buf[r, : counts[r]] for r in range(world)], 0)


def ensemble_rollout(step: Callable[[Tensor], Tensor], x_members: Tensor, num_steps: int,
                     num_members: int, group: Optional[dist.ProcessGroup] = None) -> Tensor:
    """Apply ``step`` ``num_steps`` times to this rank's members ``[members_local, N, C]`` (each output
    feeds the next step, so C_out must equal C_in for num_steps > 1), then gather every rank's final
    state once.  Returns ``[num_members, N, C_out]`` on every rank."""
    state = x_members
    for _ in range(num_steps):
        state = step(state)
    return gather_members(state, num_members, group)
