"""Random-effects occupancy model (biolith/models/occu.py:170-173, 191-196, 215-218) -- oracle side, no GPU:
the C oracle's potential against the literal NumPy model statement and against finite differences for the
three option combinations, and its NUTS (vectors on the heap, one RNG stream per coordinate) on a small case."""
import numpy as np
import pytest

import oracle
from conftest import load_golden


@pytest.mark.parametrize("name,site,obs,scales", [("small_3x3", True, False, (1.0, 1.0)), ("small_3x3", False, True, (1.0, 0.5)),
                                                  ("missing", True, True, (0.7, 2.0))])
def test_re_potential_equals_literal_model_and_fd(name, site, obs, scales):
    g = load_golden(name)
    N, T, J = g["obs"].shape[-3:]
    kw = dict(site_random_effects=site, obs_random_effects=obs, prior_site_re_sd=scales[0], prior_obs_re_sd=scales[1])
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_re", **kw)
    G0 = g["site_covs"].shape[1] + g["obs_covs"].shape[3] + 2
    assert od.D == G0 + site + obs + (2 * N if site else 0) + (N * T * J if obs else 0)
    rng = np.random.default_rng(5)
    for _ in range(2):
        th = rng.uniform(-1.2, 1.2, size=od.D)
        U, G = od.potential_grad(th)
        lit = oracle.literal_log_joint_re(th, g["site_covs"], g["obs_covs"], g["obs"], **kw)
        assert U == pytest.approx(-lit, rel=1e-12, abs=1e-9)
        h = 1e-6
        for k in list(range(G0 + site + obs)) + list(rng.integers(G0, od.D, size=12)):   # all globals, a sample of the effects
            e = np.zeros(od.D); e[k] = h
            fd = (od.potential_grad(th + e)[0] - od.potential_grad(th - e)[0]) / (2 * h)
            assert abs(fd - G[k]) <= 2e-6 * max(1.0, np.max(np.abs(G))), (k, fd, G[k])


def test_re_effects_at_zero_reduce_to_the_plain_model():
    g = load_golden("small_3x3")
    N = g["obs"].shape[-3]
    od0 = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"])
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_re", site_random_effects=True)
    th = np.random.default_rng(0).uniform(-1, 1, size=od0.D)
    phi = 0.3
    U, G = od.potential_grad(np.concatenate([th, [phi], np.zeros(2 * N)]))
    U0, G0 = od0.potential_grad(th)
    sd = np.exp(phi)
    prior = 0.5 * np.log(2 / np.pi) - 0.5 * sd ** 2 + phi + 2 * N * (-phi - 0.5 * np.log(2 * np.pi))
    assert U == pytest.approx(U0 - prior, rel=1e-12)
    assert np.allclose(G[: od0.D], G0, rtol=1e-12, atol=1e-12)


def test_re_streams_do_not_collide_across_chains():
    """D + 2 streams per chain (one per coordinate, then the two scalar streams): chain c starts c * stride jumps into
    the one sequence of jumped states, so chain 1's streams are exactly the continuation of chain 0's."""
    import ctypes as C

    L = oracle.oracle.lib()
    stride = 2112
    both = np.zeros((2 * stride, 4), dtype=np.uint32)
    L.orc_rng_streams_strided(7, 0, stride, 2 * stride, both.ctypes.data_as(C.POINTER(C.c_uint32)))
    second = np.zeros((stride, 4), dtype=np.uint32)
    L.orc_rng_streams_strided(7, 1, stride, stride, second.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert np.array_equal(both[stride:], second)
    assert len({tuple(r) for r in both}) == 2 * stride
    small = np.zeros((64, 4), dtype=np.uint32)   # the small models' layout is the stride-64 case
    L.orc_rng_streams(7, 1, 64, small.ctypes.data_as(C.POINTER(C.c_uint32)))
    strided = np.zeros((64, 4), dtype=np.uint32)
    L.orc_rng_streams_strided(7, 1, 64, 64, strided.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert np.array_equal(small, strided)


def test_re_oracle_nuts_recovers_a_sensible_posterior():
    g = load_golden("small_3x3")
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_re", site_random_effects=True)
    r = oracle.nuts_run(od, 300, 300, num_chains=2, seed=0)
    G0 = od.Ks + od.Ko + 2
    sd = np.exp(r["draws"][:, :, G0])
    assert np.all(np.isfinite(r["draws"])) and r["diverging"].mean() < 0.1
    assert 0.05 < sd.mean() < 3.0
    # fixed effects stay in the neighbourhood of the plain model's posterior mean
    o0 = oracle.nuts_run(oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"]), 300, 300, num_chains=2, seed=0)
    assert np.max(np.abs(r["draws"][:, :, :G0].mean((0, 1)) - o0["draws"].mean((0, 1)))) < 1.0


def test_oracle_sampler_is_pinned_by_its_own_first_draws():
    """The oracle's sampler output for fixed seeds, captured when the GPU kernels reproduced its trees transition by
    transition: any change of stream layout, adaptation or tree logic in the restatement shows up here, on the CPU."""
    import json
    import os

    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_first_draws.json")))
    g = load_golden("small_3x3")
    r = oracle.nuts_run(oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"]), 20, 3, num_chains=3, seed=3)
    assert np.array_equal(r["num_steps"], np.array(ref["occu"]["num_steps"]))
    assert np.allclose(r["draws"], np.array(ref["occu"]["draws"]), rtol=0, atol=1e-9)
    assert np.allclose(r["step_size"], np.array(ref["occu"]["step_size"]), rtol=1e-10)
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_re", site_random_effects=True)
    r = oracle.nuts_run(od, 10, 2, num_chains=2, seed=3)
    assert np.array_equal(r["num_steps"], np.array(ref["occu_re_site"]["num_steps"]))
    assert np.allclose(r["draws"][:, :, :12], np.array(ref["occu_re_site"]["draws"]), rtol=0, atol=1e-9)


def test_re_several_species_potential_is_the_sum_with_shared_sds():
    """Random effects inside the species plate, their sds outside it (occu.py:170-173, 182-196): the joint potential over
    [species' beta, alpha | log sds | site_re_occ [S][N] | site_re_det [S][N] | obs_re [S][N][T][J]] equals the sum of the species'
    one-species potentials with the sds' own prior + Jacobian counted once; gradient against central differences."""
    import math
    rng = np.random.default_rng(0)
    N, T, J, Ks, Ko, S = 12, 2, 3, 2, 1, 3
    X = rng.normal(size=(N, Ks)).astype(np.float32)
    W = rng.normal(size=(N, T, J, Ko)).astype(np.float32)
    Y = (rng.uniform(size=(S, N, T, J)) < 0.4).astype(np.float32)
    Y[1, 3, 0, 1] = np.nan
    W[5, 1, 2, 0] = np.nan
    kw = dict(model="occu_re", site_random_effects=True, obs_random_effects=True, prior_site_re_sd=0.7, prior_obs_re_sd=1.3)
    od = oracle.OracleData(X, W, Y, **kw)
    V, Dp = T * J, Ks + Ko + 2
    assert od.D == S * Dp + 2 + S * (2 * N + N * V) and od.n_species == S
    th = rng.uniform(-1, 1, size=(1, od.D))
    U, G = od.potential_grad(th)
    num = np.zeros(od.D)
    for k in range(od.D):
        a, b = th.copy(), th.copy()
        a[0, k] += 1e-6
        b[0, k] -= 1e-6
        num[k] = (od.potential_grad(a)[0][0] - od.potential_grad(b)[0][0]) / 2e-6
    assert np.max(np.abs(num - G[0])) <= 1e-6 * np.max(np.abs(G[0]))
    o_sd = S * Dp
    o_u, o_v, o_e = o_sd + 2, o_sd + 2 + S * N, o_sd + 2 + 2 * S * N
    tot = 0.0
    for s in range(S):
        one = oracle.OracleData(X, W, Y[s], **kw)
        t1 = np.concatenate([th[0, s * Dp:(s + 1) * Dp], th[0, o_sd:o_sd + 2], th[0, o_u + s * N:o_u + (s + 1) * N],
                             th[0, o_v + s * N:o_v + (s + 1) * N], th[0, o_e + s * N * V:o_e + (s + 1) * N * V]])[None]
        tot += one.potential_grad(t1)[0][0]
    own = 0.0   # -log HalfNormal(sd; s0) - log |d sd / d phi| of the two sds
    for phi, s0 in ((th[0, o_sd], 0.7), (th[0, o_sd + 1], 1.3)):
        sd = math.exp(phi)
        own += -(0.5 * math.log(2 / math.pi) - math.log(s0) - 0.5 * sd * sd / (s0 * s0) + phi)
    assert U[0] == pytest.approx(tot - (S - 1) * own, rel=1e-12)


@pytest.mark.parametrize("mode", ["constant", "unoccupied"])
@pytest.mark.parametrize("site,obs", [(True, False), (False, True), (True, True)])
def test_re_with_false_positives_equals_literal_model_and_finite_differences(mode, site, obs):
    """occu(site_random_effects / obs_random_effects, false_positives_* ) together (occu.py:146-157 with :170-173, 191-196):
    theta = [beta, alpha, phi = logit(rate), (log sds), (effects)]; the C closed form against the literal statement (z summed by
    brute force, numpyro's clamps) and its analytic gradient against central differences."""
    rng = np.random.default_rng(7)
    N, T, J, Ks, Ko = 9, 2, 4, 2, 2
    X, W = rng.normal(size=(N, Ks)), rng.normal(size=(N, T, J, Ko))
    Y = (rng.uniform(size=(N, T, J)) < 0.4) * 1.0
    Y[1, 1, 1] = np.nan
    W[3, 1, 0, 1] = np.nan
    od = oracle.OracleData(X, W, Y, model="occu_re", site_random_effects=site, obs_random_effects=obs, prior_site_re_sd=0.7,
                           prior_obs_re_sd=1.3, re_fp_mode=mode, prior_fp=(2.0, 6.0))
    assert od.D == Ks + Ko + 2 + 1 + site + obs + (2 * N if site else 0) + (N * T * J if obs else 0)
    th = rng.uniform(-1, 1, size=od.D)
    U, G = od.potential_grad(th)
    lit = oracle.literal_log_joint_re(th, X, W, Y, site, obs, 0.7, 1.3, re_fp_mode=mode, prior_fp=(2.0, 6.0))
    assert abs(U + lit) < 1e-10 * max(1.0, abs(U))
    h = 1e-6
    fd = np.array([(od.potential_grad(th + h * e)[0] - od.potential_grad(th - h * e)[0]) / (2 * h) for e in np.eye(od.D)])
    assert np.max(np.abs(fd - G)) < 1e-6 * max(1.0, np.max(np.abs(G)))
