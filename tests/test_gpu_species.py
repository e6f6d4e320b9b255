"""Several species under ONE chain (the species plate of biolith/models/occu.py:182-186 inside one NUTS; a false-positive rate
shared across the plate, occu.py:146-157): theta = [species 0: beta, alpha | species 1: ... | (phi)].  K1 parity and the first
trees against the oracle's joint potential, the reference's own multi-species test, and agreement with the species-by-species
form (same marginals)."""
import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from biolith_amd.evaluation import split_gelman_rubin
from biolith_amd.models import occu, simulate
from biolith_amd.utils import fit

pytestmark = pytest.mark.gpu


def _data(rng, S, N=150, T=2, J=4, Ks=2, Ko=3):
    X = rng.normal(size=(N, Ks)) * 0.6
    W = rng.normal(size=(N, T, J, Ko)) * 0.6
    Y = (rng.uniform(size=(S, N, T, J)) < rng.uniform(0.15, 0.45, size=(S, 1, 1, 1))) * 1.0
    Y[rng.uniform(size=Y.shape) < 0.07] = np.nan
    W[5, 1, 2, 0] = np.nan
    X[9, 1] = np.nan
    return X, W, Y


@pytest.mark.parametrize("S,model,kw", [(2, "occu", {}), (3, "occu", {}), (2, "occu_fp", dict(fp_mode="constant")),
                                        (3, "occu_fp", dict(fp_mode="unoccupied", prior_fp=(2.0, 8.0)))])
def test_joint_species_logp_and_first_trees(S, model, kw):
    rng = np.random.default_rng(10 * S + len(model))
    X, W, Y = _data(rng, S)
    od, ds = oracle.OracleData(X, W, Y, model=model, **kw), OccuDataset(X, W, Y, model=model, **kw)
    Dsp = X.shape[1] + W.shape[3] + 2
    assert ds.D == od.D == S * Dsp + (model == "occu_fp")
    th = rng.uniform(-1.2, 1.2, size=(4, od.D)).astype(np.float32).astype(np.float64)
    Uo, Go = od.potential_grad(th)
    Ug, Gg = ds.logp_grad(th)
    assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= 1e-6, (Ug, Uo)
    assert np.max(np.abs(Gg - Go)) <= 1e-5 * np.max(np.abs(Go))
    # the joint potential is the sum of the species' potentials (shared rate's prior once)
    parts = [OccuDataset(X, W, Y[s:s + 1], model=model, **kw) for s in range(S)]
    if model == "occu":
        U1 = sum(p.logp_grad(th[:, s * Dsp:(s + 1) * Dsp])[0] for s, p in enumerate(parts))
        assert np.allclose(U1, Ug, rtol=1e-6)
    # same streams on both sides: the first transitions build the same trees (no adaptation, as tests/test_gpu_fp.py does for
    # the false-positive model, whose long trees let float32 / float64 rounding part the two sides soon after)
    W_ = 10 if model == "occu" else 0
    init = rng.uniform(-0.5, 0.5, size=(2, od.D))
    o = oracle.nuts_run(od, W_, 5, num_chains=2, seed=3, init=init)
    for k in (0, 1, 3):
        r = ds.nuts(num_warmup=W_, num_samples=5, num_chains=2, seed=3, init_theta=init, wgs_per_chain=k)
        assert np.array_equal(o["num_steps"][:, :3], r.num_steps[:, :3]), (k, o["num_steps"], r.num_steps)
        assert np.allclose(o["draws"][:, 0], r.draws[:, 0], atol=2e-2 if W_ else 3e-3)
        assert np.allclose(o["step_size"], r.step_size, rtol=0.1)


def test_occu_multi_species():  # occu.py:478-492 (the reference's own test: one chain over both species)
    data, true_params = simulate(n_species=2, n_sites=30, simulate_missing=True)
    results = fit(occu, **data, num_chains=1, num_samples=100, num_warmup=100)
    assert results.samples["psi"].shape[-1] == 2
    assert results.samples["cov_state_0"].shape == (100, 2) and results.samples["prob_detection"].shape[-1] == 2
    # ONE sampler for both species: one step size, one tree size per transition
    assert results.mcmc.result.step_size.shape == (1,) and results.mcmc.result.draws.shape == (1, 100, 2 * 4)


def test_joint_chain_and_species_by_species_have_the_same_marginals():
    data, truth = simulate(n_species=3, n_sites=250, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=42, random_seed=5)
    kw = dict(num_chains=4, num_samples=500, num_warmup=500)
    a = fit(occu, **data, **kw)                                   # joint
    b = fit(occu, **data, **kw, joint_species=False, random_seed=7)   # one sampler per species
    assert a.mcmc.result.draws.shape == b.mcmc.result.draws.shape == (4, 500, 3 * 6)
    assert a.mcmc.result.step_size.shape == (4,)
    for k in a.samples:
        if k.startswith("cov_"):
            ma, mb, sa, sb = a.samples[k].mean(0), b.samples[k].mean(0), a.samples[k].std(0), b.samples[k].std(0)
            assert np.all(np.abs(ma - mb) < 5 * np.sqrt(sa ** 2 + sb ** 2) / np.sqrt(400)), k
            assert np.all((sa / sb > 0.8) & (sa / sb < 1.25)), k
    assert split_gelman_rubin(a.mcmc.result.draws).max() < 1.03
    assert abs(float(a.samples["psi"].mean()) - truth["z"].mean()) < 0.1


def test_joint_species_too_large_for_lds_falls_back_to_species_by_species():
    """the joint form needs all species' records in LDS for the chain count asked for, which only the launch can
    tell (16 chains leave a chain 16 workgroups: three species' 10 000-site records do not fit them).  fit() then samples the
    species one by one -- the same result, bit for bit, as asking for that form -- instead of failing."""
    data, truth = simulate(n_species=3, n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7, random_seed=2)
    kw = dict(num_chains=16, num_samples=60, num_warmup=60)
    a = fit(occu, **data, **kw)
    b = fit(occu, **data, **kw, joint_species=False)
    assert a.mcmc.result.draws.shape == (16, 60, 3 * 8)
    assert np.array_equal(a.mcmc.result.draws, b.mcmc.result.draws)
    few = fit(occu, **data, num_chains=2, num_samples=20, num_warmup=20)      # two chains: the joint form fits and is used
    assert few.mcmc.result.step_size.shape == (2,)


def test_false_positive_rate_shared_across_species():
    """occu.py:146-157: prob_fp_constant sits outside the species plate -- one rate for all species, sampled with them."""
    data, truth = simulate(n_species=2, n_sites=300, deployment_days_per_site=84, prob_fp_constant=0.1, random_seed=3)
    res = fit(occu, **data, false_positives_constant=True, num_chains=4, num_samples=400, num_warmup=400)
    assert res.samples["prob_fp_constant"].shape == (1600,)
    assert res.mcmc.result.draws.shape == (4, 400, 2 * 4 + 1)
    assert abs(res.samples["prob_fp_constant"].mean() - 0.1) < 0.05
    assert res.samples["psi"].shape == (1600, 1, 300, 2)
    assert abs(float(res.samples["psi"].mean()) - truth["z"].mean()) < 0.1
    assert res.samples["prob_detection_fp"].shape == (1600, 2, 12, 1, 300, 2)
