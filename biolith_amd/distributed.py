"""Chain sharding across the GPUs of one node + the posterior gather.

The reference's only multi-device strategy is chain-parallel sampling (``chain_method="parallel"``,
biolith/utils/fit.py:109-113: one chain per local device under ``pmap``, draws gathered implicitly
by ``mcmc.get_samples()``, fit.py:132).  Here rank ``r`` runs chains ``[first_r, first_r + count_r)``
(their RNG streams are selected by ``chain_offset``), nothing is exchanged while sampling, and ONE
collective follows: an all-gather of every rank's result block over RCCL/xGMI, behind the C-ABI
(``bl_gather_draws`` in ``include/biolith_hip.h``, on librccl directly -- no torch on the data path).

Two ways to get communicators (both end in the same ``bl_gather_draws``):

* one process per GPU: :func:`comm_from_env` -- rank 0 makes the 128-byte RCCL unique id, the others read
  it from a side channel (``torch.distributed``'s store when a process group exists, else a TCP store on
  ``MASTER_ADDR:MASTER_PORT``), every rank calls ``bl_comm_init_rank``;
* one process, several GPUs: :func:`comms_for_devices` -- ``bl_comm_init_all`` (what ``fit(devices=[...])`` uses).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi


def shard_chains(num_chains: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(chains on this rank, global id of its first chain); chains are dealt in contiguous blocks."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad world_size/rank")
    base, extra = divmod(num_chains, world_size)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return count, offset


def rccl_version() -> int:
    v = C.c_int(0)
    _ffi.check(_ffi.load().bl_comm_rccl_version(C.byref(v)))
    return int(v.value)


class RcclComm:
    """One rank's RCCL communicator (``bl_comm``)."""

    def __init__(self, handle: C.c_void_p):
        self._lib = _ffi.load()
        self._h = handle
        w, r, d, ms = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        _ffi.check(self._lib.bl_comm_info(self._h, C.byref(w), C.byref(r), C.byref(d), C.byref(ms)))
        self.world, self.rank, self.device, self.init_ms = w.value, r.value, d.value, ms.value

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * _ffi.COMM_ID_BYTES)()
        _ffi.check(_ffi.load().bl_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def init_rank(cls, unique_id: bytes, world: int, rank: int, device: int) -> "RcclComm":
        if len(unique_id) != _ffi.COMM_ID_BYTES:
            raise ValueError("unique_id must be 128 bytes")
        buf = (C.c_uint8 * _ffi.COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _ffi.check(_ffi.load().bl_comm_init_rank(buf, int(world), int(rank), int(device), C.byref(h)))
        return cls(h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bl_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comms_for_devices(devices: Sequence[int]) -> List[RcclComm]:
    """``ncclCommInitAll`` over distinct local GPUs: one communicator per device, all in this process."""
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    hs = (C.c_void_p * len(devices))()
    _ffi.check(_ffi.load().bl_comm_init_all(len(devices), devs, hs))
    return [RcclComm(C.c_void_p(h)) for h in hs]


def comm_from_env(device: int, rank: Optional[int] = None, world: Optional[int] = None) -> RcclComm:
    """Process-per-GPU communicator.  The unique id travels through torch.distributed's store when a process
    group is up, else through a TCP store on MASTER_ADDR:MASTER_PORT+1 (same env contract as torchrun)."""
    import os

    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    if world == 1:
        return RcclComm.init_rank(RcclComm.unique_id(), 1, 0, device)
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        box = [RcclComm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    else:
        from datetime import timedelta

        store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]) + 1, world,
                              is_master=(rank == 0), timeout=timedelta(seconds=120))
        if rank == 0:
            store.set("biolith_rccl_id", RcclComm.unique_id())
        uid = bytes(store.get("biolith_rccl_id"))
    return RcclComm.init_rank(uid, world, rank, device)


def gather_draws(comms: Sequence[RcclComm], datasets, chains_per_rank: Sequence[int], want_result: bool = True):
    """``bl_gather_draws``: all-gather of the last finished launch of every local dataset; returns the
    :class:`~biolith_amd.engine.NutsResult` of ALL chains in rank order (or None with ``want_result=False``).

    ``comms`` / ``datasets``: the ranks this process drives (one for process-per-GPU).  ``chains_per_rank``:
    chains of every rank of the world (``shard_chains`` gives them), so block sizes are never negotiated.
    """
    lib = _ffi.load()
    n = len(comms)
    if n == 0 or n != len(datasets):
        raise ValueError("one dataset per communicator")
    cs = (C.c_void_p * n)(*[c._h for c in comms])
    ds = (C.c_void_p * n)(*[d._h for d in datasets])
    counts = (C.c_int32 * len(chains_per_rank))(*[int(c) for c in chains_per_rank])
    if len(chains_per_rank) != comms[0].world:
        raise ValueError("chains_per_rank must have one entry per rank")
    if not want_result:
        _ffi.check(lib.bl_gather_draws(cs, ds, n, counts, None))
        return None
    d0 = datasets[0]
    a, out = d0._output(int(sum(chains_per_rank)), d0._shape[1])
    _ffi.check(lib.bl_gather_draws(cs, ds, n, counts, C.byref(out)))
    res = d0._result(a)
    return res


def gather_host_arrays(local, group=None):
    """Control-plane helper for hosts without RCCL (the ``gloo`` tests): all-gather of a per-rank torch tensor
    ``(c_local, ...)`` with possibly different ``c_local`` into ``(sum c_local, ...)``.  Not used by ``fit`` or ``bench.py``
    on a GPU -- there the gather is ``bl_gather_draws``."""
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
