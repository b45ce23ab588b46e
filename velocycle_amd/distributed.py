"""Host-side helpers for the cell-sharded fit: one process per GPU, `torch.distributed`
(backend nccl = RCCL on ROCm; gloo in the CPU / one-device tests).  The data path has exactly one exchange
per SVI step (the all-reduce of `grad[0 : header + n_global]` in `SVIRunner`); everything here runs once
per fit: agreeing on the seed, and gathering the per-cell results (`ϕxy_locs`, per-cell posterior sites,
the columns of the ElogS / ElogU summaries) so that every rank ends with the full-`Nc` attributes the
reference's `fit()` leaves behind (velocity_inference_model.py:153-187, phase_inference_model.py:187-201).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def dist_context(process_group=None) -> Tuple[int, int, Optional[object]]:
    """(rank, world_size, group).  (0, 1, None) when torch.distributed is not initialised."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1, None
    return dist.get_rank(process_group), dist.get_world_size(process_group), process_group


def _comm_device(group, fallback: torch.device) -> torch.device:
    import torch.distributed as dist
    return fallback if dist.get_backend(group) == "nccl" else torch.device("cpu")


def broadcast_int(value: int, group=None, device: Optional[torch.device] = None, src: int = 0) -> int:
    """rank `src`'s value on every rank (seeds, draw bases)."""
    import torch.distributed as dist
    rank, world, group = dist_context(group)
    if world == 1:
        return int(value)
    dev = _comm_device(group, device or torch.device("cpu"))
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return int(t.item())


def gather_cells(local: torch.Tensor, dim: int, sizes: List[int], group=None) -> torch.Tensor:
    """Concatenate every rank's block of cells along `dim` (block r has sizes[r] cells); returns a CPU tensor on
    every rank.  Blocks are padded to the largest shard so that one all_gather of equal shapes suffices."""
    import torch.distributed as dist
    rank, world, group = dist_context(group)
    if world == 1:
        return local.detach().cpu()
    dev = _comm_device(group, local.device)
    x = local.detach().movedim(dim, 0).contiguous().to(dev)
    assert x.shape[0] == sizes[rank], (x.shape, sizes, rank)
    nmax = max(sizes)
    if x.shape[0] < nmax:
        x = torch.cat([x, x.new_zeros((nmax - x.shape[0],) + tuple(x.shape[1:]))])
    parts = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(parts, x, group=group)
    out = torch.cat([p[: sizes[r]] for r, p in enumerate(parts)]).movedim(0, dim)
    return out.cpu()
