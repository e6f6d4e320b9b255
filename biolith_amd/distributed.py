"""Chain sharding across the GPUs of one node + the posterior gather.

The reference's only multi-device strategy is chain-parallel sampling (``chain_method="parallel"``,
biolith/utils/fit.py:109-113: one chain per local device under ``pmap``, draws gathered implicitly
by ``mcmc.get_samples()``, fit.py:132).  Here: one process per GPU, rank ``r`` runs chains
``[r*c, (r+1)*c)`` (their RNG streams are selected by ``chain_offset``), no communication while
sampling, and ONE collective at the end: an all-gather of the draws over RCCL/xGMI
(``torch.distributed`` backend ``"nccl"``; ``"gloo"`` on CPU in the tests).
"""
from __future__ import annotations

from typing import Tuple


def shard_chains(num_chains: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(chains on this rank, global id of its first chain); chains are dealt in contiguous blocks."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad world_size/rank")
    base, extra = divmod(num_chains, world_size)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return count, offset


def gather_draws(local, group=None):
    """All-gather per-rank draws ``(c_local, S, D)`` into ``(sum c_local, S, D)`` on every rank.

    ``local`` is a torch tensor (CUDA for RCCL, CPU for gloo).  Ranks may hold different chain
    counts; shorter shards are padded to the longest for the collective and trimmed afterwards.
    """
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
    counts = [int(c.item()) for c in counts]
    cmax = max(counts)
    if local.shape[0] < cmax:
        pad = torch.zeros((cmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * cmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    parts = [out[r * cmax: r * cmax + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0)
