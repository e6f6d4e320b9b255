"""The host-only half of the posterior gather (biolith/utils/fit.py:109-113, 132 -> bl_gather_draws): where every rank's
result block lies in the gathered buffer and how the blocks unpack into chains in rank order -- for even and uneven chain
counts and world sizes 2, 3 and 8, against a NumPy concatenation.  No GPU, no RCCL: this is the part of the N > 1 path
that CAN run here; the collective itself (ncclAllGather / grouped ncclBroadcast over world > 1) runs on multi-GPU nodes only."""
import ctypes as C

import numpy as np
import pytest

from biolith_amd import _ffi

FIELDS = [("draws", np.float32), ("diverging", np.uint8), ("num_steps", np.int32), ("accept_prob", np.float32),
          ("potential_energy", np.float32), ("step_size", np.float32), ("inv_mass", np.float32), ("n_leapfrog", np.int64)]


def _layout(lib, chains, S, D):
    off = (C.c_uint64 * 9)()
    _ffi.check(lib.bl_result_block_layout(chains, S, D, off))
    return [int(x) for x in off]


def _rank_arrays(rng, chains, S, D):
    return {"draws": rng.normal(size=(chains, S, D)).astype(np.float32), "diverging": rng.integers(0, 2, size=(chains, S)).astype(np.uint8),
            "num_steps": rng.integers(1, 1024, size=(chains, S)).astype(np.int32), "accept_prob": rng.uniform(size=(chains, S)).astype(np.float32),
            "potential_energy": rng.normal(size=(chains, S)).astype(np.float32), "step_size": rng.uniform(size=(chains,)).astype(np.float32),
            "inv_mass": rng.uniform(size=(chains, D)).astype(np.float32), "n_leapfrog": rng.integers(0, 1 << 40, size=(chains, 2)).astype(np.int64)}


@pytest.mark.parametrize("counts", [[4, 4], [2, 1], [1, 1, 1], [3, 2, 2], [4] * 8, [2, 2, 2, 1, 1, 1, 1, 1], [1, 5, 1, 2, 7, 1, 3, 1]])
@pytest.mark.parametrize("S,D", [(1000, 8), (7, 13), (3, 61)])
def test_unpack_equals_numpy_concatenation(counts, S, D):
    lib = _ffi.load()
    rng = np.random.default_rng(len(counts) * 1000 + S + D)
    world = len(counts)
    per_rank, blocks = [], []
    for c in counts:
        a = _rank_arrays(rng, c, S, D)
        off = _layout(lib, c, S, D)
        assert all(o % 256 == 0 for o in off) and off == sorted(off)
        blk = np.frombuffer(rng.bytes(off[8]), dtype=np.uint8).copy()      # padding between fields: arbitrary bytes
        for (name, _), o in zip(FIELDS, off):
            raw = a[name].tobytes()
            assert o + len(raw) <= off[8]
            blk[o:o + len(raw)] = np.frombuffer(raw, dtype=np.uint8)
        per_rank.append(a)
        blocks.append(blk)
    gathered = np.concatenate(blocks)
    total = sum(counts)
    out_arrays = {"draws": np.zeros((total, S, D), np.float32), "diverging": np.zeros((total, S), np.uint8), "num_steps": np.zeros((total, S), np.int32),
                  "accept_prob": np.zeros((total, S), np.float32), "potential_energy": np.zeros((total, S), np.float32),
                  "step_size": np.zeros(total, np.float32), "inv_mass": np.zeros((total, D), np.float32), "n_leapfrog": np.zeros((total, 2), np.int64)}
    out = _ffi.bl_nuts_output()
    for (name, _), (fname, ftype) in zip(FIELDS, _ffi.bl_nuts_output._fields_):
        assert name == fname
        setattr(out, fname, out_arrays[name].ctypes.data_as(ftype))
    cnt = (C.c_int32 * world)(*counts)
    _ffi.check(lib.bl_gather_unpack(gathered.ctypes.data_as(C.c_void_p), gathered.nbytes, world, cnt, S, D, C.byref(out)))
    for name, _ in FIELDS:
        want = np.concatenate([a[name] for a in per_rank], axis=0)
        assert np.array_equal(out_arrays[name], want), name


def test_unpack_rejects_a_buffer_of_the_wrong_size_and_empty_ranks():
    lib = _ffi.load()
    off = _layout(lib, 2, 10, 8)
    buf = np.zeros(2 * off[8] + 256, np.uint8)
    out = _ffi.bl_nuts_output()
    with pytest.raises(ValueError, match="bytes given"):
        _ffi.check(lib.bl_gather_unpack(buf.ctypes.data_as(C.c_void_p), buf.nbytes, 2, (C.c_int32 * 2)(2, 2), 10, 8, C.byref(out)))
    with pytest.raises(ValueError, match="no chains"):
        _ffi.check(lib.bl_gather_unpack(buf.ctypes.data_as(C.c_void_p), buf.nbytes, 2, (C.c_int32 * 2)(2, 0), 10, 8, C.byref(out)))


def test_block_layout_is_the_launch_carve():
    """draws first, every field 256-byte aligned, sizes as bl_nuts_fetch returns them (num_samples = 0 keeps one slot)."""
    lib = _ffi.load()
    off = _layout(lib, 4, 1000, 8)
    assert off[0] == 0 and off[1] == 4 * 1000 * 8 * 4 and off[8] >= off[7] + 4 * 16
    assert _layout(lib, 4, 0, 8)[8] == _layout(lib, 4, 1, 8)[8]
