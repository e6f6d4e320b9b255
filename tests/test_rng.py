"""xoshiro128++ streams: the engine's host-side generator (C-ABI bl_rng_streams) against the
oracle's, plus structural properties of next()/jump()."""
import ctypes as C

import numpy as np

import oracle
from biolith_amd.engine import rng_streams


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def test_engine_streams_equal_oracle_streams():
    for seed, chain in [(0, 0), (0, 3), (12345, 1), (2**63 + 5, 2)]:
        assert np.array_equal(rng_streams(seed, chain), oracle.rng_streams(seed, chain))
    # chain c's streams continue where chain c-1's 64 streams end
    a = oracle.rng_streams(9, 0, 128)
    assert np.array_equal(a[64:], oracle.rng_streams(9, 1, 64))
    assert len({tuple(r) for r in a}) == 128


def test_jump_commutes_with_next():
    L = oracle.lib()
    s = oracle.rng_streams(42, 0, 1)[0].copy()
    a, b = s.copy(), s.copy()
    for _ in range(17):
        L.orc_rng_next(_u32p(a))
    L.orc_rng_jump(_u32p(a))
    L.orc_rng_jump(_u32p(b))
    for _ in range(17):
        L.orc_rng_next(_u32p(b))
    assert np.array_equal(a, b)


def test_next_matches_python_restatement():
    def rotl(x, k):
        return ((x << k) | (x >> (32 - k))) & 0xFFFFFFFF

    s = [1, 2, 3, 4]
    st = np.array(s, dtype=np.uint32)
    L = oracle.lib()
    for _ in range(50):
        res = (rotl((s[0] + s[3]) & 0xFFFFFFFF, 7) + s[0]) & 0xFFFFFFFF
        t = (s[1] << 9) & 0xFFFFFFFF
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 11)
        assert L.orc_rng_next(_u32p(st)) == res
    assert list(st) == s


def test_uniform_and_normal_moments():
    L = oracle.lib()
    st = oracle.rng_streams(1, 0, 1)[0].copy()
    u = np.array([L.orc_rng_uniform(_u32p(st)) for _ in range(20000)])
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    assert np.all(u.astype(np.float32).astype(np.float64) == u)  # exactly representable in float32
    z = np.array([L.orc_rng_normal(_u32p(st)) for _ in range(20000)])
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03
