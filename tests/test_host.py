"""Host-side mirror of the reference interface: prepare_data / rename_samples
(biolith/utils/data.py), occu()'s validation (biolith/models/occu.py:103-133), fit()'s argument
handling (biolith/utils/fit.py:85-104), diagnostics (biolith/evaluation/diagnostics.py)."""
import numpy as np
import pandas as pd
import pytest

import oracle
from biolith_amd.distributions import Beta, HalfNormal, Laplace, Normal
from biolith_amd.evaluation import diagnostics, effective_sample_size, split_gelman_rubin, summary
from biolith_amd.models import occu
from biolith_amd.regression import AbstractRegression, LinearRegression
from biolith_amd.utils import fit
from biolith_amd.utils.data import prepare_data, rename_samples
from biolith_amd.utils.mcmc import HipMCMC, LazySamples
from conftest import load_golden


def test_prepare_data_ndarray_shapes_and_names():
    g = load_golden("default")
    sc, oc, ob, sd, sn, on = prepare_data(g["site_covs"], g["obs_covs"], g["obs"])
    assert sc.dtype == oc.dtype == ob.dtype == np.float32 and sd is None
    assert sc.shape == (100, 1) and oc.shape == (100, 1, 52, 1) and ob.shape == (1, 100, 1, 52)
    assert sn == ["0", "1"] and on == ["0", "1"]  # data.py:130-133
    # season-less inputs get the period axis (data.py:113-128)
    _, oc2, ob2, sd2, _, _ = prepare_data(None, np.zeros((5, 3)), np.zeros((5, 3)), np.ones((5, 3)))
    assert oc2.shape == (5, 1, 3, 1) and ob2.shape == (5, 1, 3) and sd2.shape == (5, 1, 3)
    _, oc3, _, _, _, on3 = prepare_data(None, np.zeros((5, 3, 2)))
    assert oc3.shape == (5, 1, 3, 2) and on3 == ["0", "1", "2"]


def test_prepare_data_dataframes_align_and_name():
    sites = [f"s{i}" for i in range(4)]
    site_df = pd.DataFrame({"elev": [1.0, 2, 3, 4], "forest": [0.1, 0.2, 0.3, 0.4]}, index=sites)
    cols = pd.MultiIndex.from_product([["effort", "temp"], [0, 1, 2]])
    obs_cov_df = pd.DataFrame(np.arange(24.0).reshape(4, 6), index=sites, columns=cols)
    obs_df = pd.DataFrame(np.eye(4, 3), index=sites[::-1])  # reversed order: must not win
    sc, oc, ob, _, sn, on = prepare_data(site_df, obs_cov_df, obs_df)
    assert sn == ["intercept", "elev", "forest"] and on == ["intercept", "effort", "temp"]
    assert oc.shape == (4, 1, 3, 2) and ob.shape == (4, 1, 3)
    # obs is the reference frame (first DataFrame among obs, site_covs, obs_covs): others follow ITS order
    assert np.array_equal(sc[:, 0], [4, 3, 2, 1])
    assert np.array_equal(oc[0, 0, :, 0], obs_cov_df.loc["s3"].to_numpy()[:3])
    assert np.array_equal(oc[0, 0, :, 1], obs_cov_df.loc["s3"].to_numpy()[3:])
    cols3 = pd.MultiIndex.from_product([["effort"], [0, 1], [0, 1, 2]])
    oc3 = prepare_data(None, pd.DataFrame(np.arange(24.0).reshape(4, 6), columns=cols3))[1]
    assert oc3.shape == (4, 2, 3, 1) and oc3[1, 1, 2, 0] == 11.0
    with pytest.raises(ValueError, match="MultiIndex"):
        prepare_data(None, pd.DataFrame(np.zeros((4, 6))))


def test_rename_samples():
    s = dict(beta=np.arange(12.0).reshape(6, 1, 2), alpha=np.arange(18.0).reshape(6, 1, 3), psi=np.zeros(3))
    r = rename_samples(s, ["0", "1"], ["intercept", "a", "b"])
    assert set(r) == {"cov_state_0", "cov_state_1", "cov_det_intercept", "cov_det_a", "cov_det_b", "psi"}
    assert r["cov_state_1"].shape == (6, 1) and np.array_equal(r["cov_det_b"][:, 0], s["alpha"][:, 0, 2])
    assert "beta" in s  # input not mutated


def test_occu_validates_like_reference():
    g = load_golden("seed7_2x1")
    spec = occu(g["site_covs"], g["obs_covs"], obs=g["obs"], coords=None, ell=0.0)
    assert spec.shape == dict(S=1, N=64, T=1, J=8, Ks=2, Ko=1)
    assert spec.prior_beta == (0.0, 1.0)
    with pytest.raises(AssertionError, match="obs must be None or of shape"):
        occu(g["site_covs"], g["obs_covs"], obs=g["obs"][0])
    with pytest.raises(AssertionError, match="same number of sites"):
        occu(g["site_covs"][:10], g["obs_covs"], obs=g["obs"])
    with pytest.raises(AssertionError, match="cannot both be True"):
        occu(g["site_covs"], g["obs_covs"], obs=g["obs"], false_positives_constant=True, false_positives_unoccupied=True)
    for kw in (dict(coords=np.zeros((64, 2))), dict(regressor_occ=AbstractRegression)):
        with pytest.raises(NotImplementedError):
            occu(g["site_covs"], g["obs_covs"], obs=g["obs"], **kw)
    # random effects (occu.py:170-173, 191-196, 215-218): one species, HalfNormal priors on the sds, not with false positives
    re = occu(g["site_covs"], g["obs_covs"], obs=g["obs"], site_random_effects=True, prior_site_re_sd=HalfNormal(0.5))
    assert re.model == "occu_re" and re.extras == dict(site_random_effects=True, obs_random_effects=False,
                                                        prior_site_re_sd=0.5, prior_obs_re_sd=1.0)
    assert occu(g["site_covs"], g["obs_covs"], obs=g["obs"], obs_random_effects=True).extras["obs_random_effects"]
    with pytest.raises(NotImplementedError, match="HalfNormal"):
        occu(g["site_covs"], g["obs_covs"], obs=g["obs"], obs_random_effects=True, prior_obs_re_sd=Normal())
    # several species share the sds (occu.py:170-173): accepted, fit() samples them under one chain
    two = occu(g["site_covs"], g["obs_covs"], obs=np.concatenate([g["obs"], g["obs"]]), site_random_effects=True)
    assert two.model == "occu_re" and two.obs.shape[0] == 2
    # random effects together with false positives (occu.py:146-157 with :170-173): one species
    both = occu(g["site_covs"], g["obs_covs"], obs=g["obs"], site_random_effects=True, false_positives_constant=True)
    assert both.model == "occu_re" and both.extras["re_fp_mode"] == "constant" and both.extras["prior_fp"] == (2.0, 5.0)
    with pytest.raises(NotImplementedError, match="together with false positives for several species"):
        occu(g["site_covs"], g["obs_covs"], obs=np.concatenate([g["obs"], g["obs"]]), site_random_effects=True, false_positives_constant=True)
    # false positives (occu.py:146-157): one species, Beta prior on the rate
    fp = occu(g["site_covs"], g["obs_covs"], obs=g["obs"], false_positives_constant=True)
    assert fp.model == "occu_fp" and fp.extras == dict(fp_mode="constant", prior_fp=(2.0, 5.0))
    fu = occu(g["site_covs"], g["obs_covs"], obs=g["obs"], false_positives_unoccupied=True, prior_prob_fp_unoccupied=Beta(1.0, 9.0))
    assert fu.extras == dict(fp_mode="unoccupied", prior_fp=(1.0, 9.0))
    with pytest.raises(NotImplementedError, match="Beta"):
        occu(g["site_covs"], g["obs_covs"], obs=g["obs"], false_positives_constant=True, prior_prob_fp_constant=Normal())
    # several species share the rate (occu.py:146-157): accepted, fit() samples all species under one chain
    fp2 = occu(g["site_covs"], g["obs_covs"], obs=np.concatenate([g["obs"], g["obs"]]), false_positives_constant=True)
    assert fp2.model == "occu_fp" and fp2.n_species == 2
    two = occu(g["site_covs"], g["obs_covs"], obs=np.concatenate([g["obs"], g["obs"]]))
    assert two.n_species == 2 and two.shape["S"] == 2
    assert occu(g["site_covs"], g["obs_covs"], obs=g["obs"], prior_beta=Normal(0.5, 2.0)).prior_beta == (0.5, 2.0)

    class Cauchy:
        loc, scale = 0.0, 1.0

    with pytest.raises(NotImplementedError, match="Normal"):
        occu(g["site_covs"], g["obs_covs"], obs=g["obs"], prior_alpha=Cauchy())
    # Laplace coefficient priors (the other family grid_search_priors tries, utils/grid_search.py:366-371)
    lap = occu(g["site_covs"], g["obs_covs"], obs=g["obs"], prior_beta=Laplace(0.0, 0.5))
    assert lap.prior_beta == (0.0, 0.5) and lap.prior_beta.family == "laplace" and lap.prior_alpha.family == "normal"
    assert LinearRegression("beta", 2).n_covs == 2


def test_fit_argument_errors():
    g = load_golden("seed7_2x1")
    kw = dict(site_covs=g["site_covs"], obs_covs=g["obs_covs"], obs=g["obs"])
    with pytest.raises(TypeError, match="biolith_amd model"):
        fit(lambda **k: None, **kw)
    for k in ("hmc", "mixed_hmc", "discrete_hmc_gibbs", "hmcecs"):
        with pytest.raises(NotImplementedError):
            fit(occu, kernel=k, **kw)
    with pytest.raises(KeyError):
        fit(occu, kernel="bogus", **kw)  # the reference's dict lookup raises KeyError too (fit.py:92-104)
    with pytest.raises(NotImplementedError, match="init_strategy"):
        fit(occu, init_strategy=lambda: None, **kw)


def test_lazy_samples():
    calls = []
    s = LazySamples(a=1)
    s.set_lazy("big", lambda: calls.append(1) or np.ones(3))
    assert "big" in s and list(s.keys()) == ["a", "big"] and not calls
    import copy
    c = copy.copy(s)
    assert c["big"].sum() == 3 and calls == [1]
    assert s["big"].sum() == 3 and calls == [1, 1] and s["big"] is s["big"]


def test_lazy_samples_survive_dict_fast_paths(tmp_path):
    """dict(s), {**s} and f(**s) must hand out the arrays, never a placeholder (the reference returns plain
    dicts of arrays, fit.py:132-133; np.savez(path, **samples) is the common consumer)."""
    def fresh():
        s = LazySamples(a=np.zeros(2))
        s.set_lazy("psi", lambda: np.ones(3))
        return s

    assert dict(fresh())["psi"] is not None and dict(fresh())["psi"].sum() == 3
    assert {**fresh()}["psi"].sum() == 3
    assert (lambda **kw: kw["psi"].sum())(**fresh()) == 3
    assert [v is not None for v in fresh().values()] == [True, True]
    assert dict(fresh().items())["psi"].sum() == 3
    np.savez(tmp_path / "s.npz", **fresh())
    assert np.load(tmp_path / "s.npz")["psi"].sum() == 3
    s = fresh()
    assert s.get("psi").sum() == 3 and s.get("nope", 7) == 7 and s.pop("psi").sum() == 3 and "psi" not in s and len(s) == 1


class _Res:
    def __init__(self, C, S, D, rng):
        self.draws = rng.normal(size=(C, S, D)).astype(np.float32)
        self.diverging = np.zeros((C, S), bool); self.diverging[0, 3] = True
        self.num_steps = np.full((C, S), 7, np.int32)
        self.accept_prob = np.full((C, S), 0.8, np.float32)
        self.potential_energy = np.zeros((C, S), np.float32)
        self.step_size = np.ones(C, np.float32); self.inv_mass = np.ones((C, D), np.float32)


def test_mcmc_shim_and_diagnostics():
    rng = np.random.default_rng(0)
    res = _Res(3, 200, 4, rng)
    beta = res.draws[:, :, :2].reshape(3, 200, 1, 2)
    alpha = res.draws[:, :, 2:].reshape(3, 200, 1, 2)
    m = HipMCMC(res, dict(beta=beta, alpha=alpha), dict(psi=np.zeros((3, 200, 1, 5, 1)), prob_detection=lambda: np.ones((3, 200, 2, 1, 5, 1))),
                num_warmup=10, spec_shape={})
    assert m.num_chains == 3 and m.num_samples == 200
    s = m.get_samples()
    assert s["beta"].shape == (600, 1, 2) and s["psi"].shape == (600, 1, 5, 1) and s["prob_detection"].shape == (600, 2, 1, 5, 1)
    assert m.get_samples(group_by_chain=True)["alpha"].shape == (3, 200, 1, 2)
    assert m.get_extra_fields()["diverging"].shape == (600,)
    d = diagnostics(m)  # reads _states/_sample_field/_last_state like diagnostics.py:10-13
    assert set(d) == {"mean_r_hat", "mean_frac_eff", "frac_diverging", "mean_beta_sd", "mean_alpha_sd"}
    assert d["frac_diverging"] == pytest.approx(1 / 600) and 0.9 < d["mean_r_hat"] < 1.1 and 0.6 < d["mean_frac_eff"] < 1.5
    assert d["mean_beta_sd"] == pytest.approx(1.0, abs=0.1)
    m.print_summary()


def test_ess_and_rhat_against_oracle_restatement():
    rng = np.random.default_rng(5)
    x = np.zeros((4, 500, 3, 2))
    for t in range(1, 500):
        x[:, t] = 0.7 * x[:, t - 1] + rng.normal(size=(4, 3, 2))
    assert np.allclose(effective_sample_size(x), oracle.effective_sample_size(x), rtol=1e-10)
    assert np.allclose(split_gelman_rubin(x), oracle.split_gelman_rubin(x), rtol=1e-12)
    assert np.allclose(effective_sample_size(x[:1]), oracle.effective_sample_size(x[:1]), rtol=1e-10)
    # AR(1), rho=.7: ESS ~ n (1-rho)/(1+rho)
    assert abs(effective_sample_size(x).mean() / (2000 * 0.3 / 1.7) - 1) < 0.25
    t = summary(dict(x=x))
    assert t["x"]["n_eff"].shape == (3, 2) and "5.0%" in t["x"] and "95.0%" in t["x"]


@pytest.mark.parametrize("rho", [0.0, 0.5, 0.9, -0.3])
def test_ess_against_the_analytic_value_of_an_ar1_process(rho):
    """ESS is the numerator of the metric (reference: evaluation/diagnostics.py:23 -> numpyro.diagnostics.summary); pin the estimator
    to something OUTSIDE the repo: a stationary AR(1) process has integrated autocorrelation time (1 + rho) / (1 - rho), so
    ESS = C n (1 - rho) / (1 + rho) (iid: C n; antithetic rho < 0: more than C n).  Mean over 256 independent series, 4 x 2000."""
    rng = np.random.default_rng(11)
    C, n, m = 4, 2000, 256
    x = np.empty((C, n, m))
    x[:, 0] = rng.normal(size=(C, m)) / np.sqrt(1.0 - rho * rho)          # stationary start
    e = rng.normal(size=(C, n, m))
    for t in range(1, n):
        x[:, t] = rho * x[:, t - 1] + e[:, t]
    ess = effective_sample_size(x)
    want = C * n * (1.0 - rho) / (1.0 + rho)
    assert ess.shape == (m,)
    assert abs(ess.mean() / want - 1.0) < 0.05, (rho, ess.mean(), want)
    assert abs(split_gelman_rubin(x).mean() - 1.0) < (0.02 if rho >= 0.9 else 0.005)


def test_rhat_against_the_analytic_value_of_shifted_chains():
    """Two chains of iid N(0, 1) draws whose means differ by d: var+ / W -> 1 + d^2 / 2 (between-chain variance of the two means
    (ddof 1) is d^2 / 2), split-R-hat -> sqrt(1 + d^2 / 2) for long chains."""
    rng = np.random.default_rng(12)
    n, d = 20000, 1.0
    x = rng.normal(size=(2, n, 8))
    x[1] += d
    # split chains: 4 half-chains, two at 0 and two at d: variance of the four means (ddof 1) = d^2 / 3
    assert np.allclose(split_gelman_rubin(x), np.sqrt(1.0 + d * d / 3.0), rtol=0.02)


def test_grid_search_hands_kernel_and_init_strategy_to_fit_only(monkeypatch):
    """grid_search_priors(kernel='nuts') must not leak `kernel` into predict() / lppd(), whose unknown keywords
    go to the model (reference: grid_search.py:64-96 passes kernel / init_strategy to fit() alone)."""
    from biolith_amd.utils import grid_search as gs

    seen = dict(fit=[], predict=[], lppd=[])

    def fake_fit(model_fn, **kw):
        seen["fit"].append(kw)
        return gs.FitResult({}, object())

    def fake_predict(model_fn, mcmc, **kw):
        seen["predict"].append(kw)
        return {}

    def fake_lppd(model_fn, preds, **kw):
        seen["lppd"].append(kw)
        return -1.0

    monkeypatch.setattr(gs, "fit", fake_fit)
    monkeypatch.setattr(gs, "predict", fake_predict)
    monkeypatch.setattr(gs, "lppd", fake_lppd)
    g = load_golden("default")
    res = gs.grid_search_priors(occu, g["site_covs"], g["obs_covs"], g["obs"], LinearRegression, LinearRegression,
                                prior_types=["normal"], prior_params_occ={"normal": {"loc": [0.0], "scale": [1.0]}},
                                prior_params_det=False, cv_folds=2, kernel="nuts", init_strategy=None, num_samples=5, num_warmup=5, num_chains=1)
    assert res.best_score == -1.0 and len(seen["fit"]) == 3          # two folds + the refit
    assert all(kw.get("kernel") == "nuts" for kw in seen["fit"])
    assert all("kernel" not in kw and "init_strategy" not in kw for kw in seen["predict"] + seen["lppd"])


def test_pinned_pool_hands_out_page_locked_buffers_from_the_second_request_on():
    """engine._PinnedPool (outputs of bl_deterministic from 8 MB on): the first request of a size class gets ordinary pages and has a
    buffer pinned in the background (after the copy: kick()); the next one takes it; it returns to the pool with its array."""
    import ctypes as C
    import gc
    import threading
    import time

    from biolith_amd.engine import _PinnedPool

    libc = C.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes, libc.free.argtypes = C.c_void_p, [C.c_size_t], [C.c_void_p]

    class Lib:   # bl_host_alloc / bl_host_free on malloc
        allocs, frees = [], []

        def bl_host_alloc(self, size, out):
            out._obj.value = libc.malloc(size)
            self.allocs.append(out._obj.value)
            return 0

        def bl_host_free(self, p):
            self.frees.append(p.value)
            libc.free(p)
            return 0

    lib, pool = Lib(), _PinnedPool(cap=1 << 25)
    shape = (3000, 1000)   # 12 MB -> the 16 MB class
    a = pool.empty(lib, shape, np.float32)
    assert a.shape == shape and not lib.allocs     # ordinary pages, nothing pinned yet
    pool.kick(lib)
    for _ in range(200):
        if lib.allocs and not pool.pending:
            break
        time.sleep(0.01)
    assert len(lib.allocs) == 1 and pool.held == 1 << 24
    b = pool.empty(lib, shape, np.float32)
    assert b.ctypes.data == lib.allocs[0] and pool.held == 0
    b[:] = 2.0
    view = b[5:7]
    del b
    gc.collect()
    assert pool.held == 0 and float(view.sum()) == 4000.0   # a view keeps the buffer out of the pool
    del view
    gc.collect()
    assert pool.held == 1 << 24 and not lib.frees
    c = pool.empty(lib, shape, np.float32)
    assert c.ctypes.data == lib.allocs[0]
    # over the cap: a second class is asked for, pinned, and freed again when it comes back
    d = pool.empty(lib, (6000, 1000), np.float32)   # 24 MB -> the 32 MB class: held 0 + 32 MB <= cap, so it is wanted
    pool.kick(lib)
    for _ in range(200):
        if len(lib.allocs) == 2 and not pool.pending:
            break
        time.sleep(0.01)
    del c
    gc.collect()
    assert len(lib.allocs) == 2 and len(lib.frees) == 1 and pool.held == 1 << 25   # 16 + 32 MB > cap: one of them was released
    assert threading.active_count() >= 1 and d.shape == (6000, 1000)


def test_init_strategies_resolve_to_start_positions():
    """fit(init_strategy=...) (fit.py:29, 93): the descriptors of biolith_amd.utils.init and how they become the kernel's init_theta."""
    import functools

    from biolith_amd.distributions import LocScale
    from biolith_amd.utils import init_to_feasible, init_to_mean, init_to_median, init_to_sample, init_to_uniform, init_to_value
    from biolith_amd.utils.init import as_strategy, initial_positions

    kw = dict(D=7, Ks=2, Ko=3, n_species=1, plain=True, prior_beta=LocScale(0.5, 2.0), prior_alpha=LocScale(-1.0, 0.1, "laplace"),
              num_chains=3, first_chain=4, seed=9)
    assert initial_positions(None, **kw) is None and initial_positions(init_to_uniform(), **kw) is None   # the kernel's own draw
    u = initial_positions(init_to_uniform(radius=0.5), **kw)
    assert u.shape == (3, 7) and np.all(np.abs(u) <= 0.5) and len(np.unique(u)) == 21
    # a chain's start depends on its global id only: chains 5, 6 of this launch = chains 0, 1 of a launch that starts at 5
    v = initial_positions(init_to_uniform(radius=0.5), **{**kw, "first_chain": 5, "num_chains": 2})
    assert np.array_equal(v, u[1:])
    assert np.all(initial_positions(init_to_feasible(), **kw) == 0.0)
    m = initial_positions(init_to_mean(), **kw)
    assert np.all(m[:, :3] == 0.5) and np.all(m[:, 3:] == -1.0)
    s = initial_positions(init_to_sample(), **kw)
    assert s.shape == (3, 7) and np.std(s[:, :3]) > 0.3 and np.std(s[:, 3:]) < 0.5
    med = initial_positions(init_to_median(num_samples=401), **kw)
    assert np.all(np.abs(med[:, :3] - 0.5) < 0.5) and np.all(np.abs(med[:, 3:] + 1.0) < 0.05)
    val = initial_positions(init_to_value(values={"alpha": [1.0, 2.0, 3.0, 4.0]}), **kw)
    assert np.all(val[:, 3:] == [1.0, 2.0, 3.0, 4.0]) and np.all(np.abs(val[:, :3]) <= 2.0)   # the rest as init_to_uniform (NumPyro's rule)
    two = initial_positions(init_to_value(values={"beta": np.arange(6.0).reshape(2, 3)}), **{**kw, "D": 15, "n_species": 2, "plain": False})
    assert np.all(two[:, 0:3] == [0, 1, 2]) and np.all(two[:, 7:10] == [3, 4, 5])
    with pytest.raises(NotImplementedError, match="coefficient sites"):
        initial_positions(init_to_value(values={"site_re_sd": 1.0}), **kw)
    with pytest.raises(NotImplementedError, match="all regression coefficients"):
        initial_positions(init_to_median(), **{**kw, "D": 8, "plain": False})
    # NumPyro's own callables are recognised by name (functools.partial forms, as numpyro.infer.init_to_value(values=...) returns)
    def init_to_median(site=None, num_samples=15):  # noqa: F811 -- stands for numpyro.infer.init_to_median
        raise AssertionError("never called")

    st = as_strategy(functools.partial(init_to_median, num_samples=7))
    assert st.kind == "median" and st.num_samples == 7
    assert as_strategy(init_to_feasible).kind == "feasible"
    with pytest.raises(NotImplementedError, match="init_strategy"):
        as_strategy(lambda: None)
