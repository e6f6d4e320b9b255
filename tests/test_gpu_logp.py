"""K1 parity (through the C-ABI bl_logp_grad): the HIP occupancy log-density + analytic gradient
against the float64 CPU oracle on the same inputs.

Tolerance: per-term math is float32 on the device, so
|dU|/|U| <= 1e-6 and max|dgrad| / max|grad| <= 1e-5.  Measured: ~1e-8 and ~1e-7."""
import numpy as np
import pytest

import oracle
from biolith_amd.engine import OccuDataset
from conftest import load_golden, quiet_simulate

pytestmark = pytest.mark.gpu
U_RTOL, G_RTOL = 1e-6, 1e-5


def _compare(site_covs, obs_covs, obs, thetas, priors=((0.0, 1.0), (0.0, 1.0))):
    od = oracle.OracleData(site_covs, obs_covs, obs, *priors)
    ds = OccuDataset(site_covs, obs_covs, obs, *priors)
    th = np.asarray(thetas, dtype=np.float32).astype(np.float64)  # float32-representable on both sides
    Uo, Go = od.potential_grad(th)
    for staged in (True, False):
        Ug, Gg = ds.logp_grad(th, staged=staged)
        assert np.all(np.isfinite(Ug)) and np.all(np.isfinite(Gg))
        assert np.max(np.abs(Ug - Uo) / np.abs(Uo)) <= U_RTOL, (staged, Ug, Uo)
        assert np.max(np.abs(Gg - Go)) <= G_RTOL * np.max(np.abs(Go)), (staged, np.max(np.abs(Gg - Go)))
    ds.close()


@pytest.mark.parametrize("name", ["default", "missing", "missing_3periods", "small_3x3", "seed7_2x1"])
def test_golden_datasets(name):
    g = load_golden(name)
    D = g["site_covs"].shape[1] + g["obs_covs"].shape[3] + 2
    th = np.random.default_rng(2).uniform(-2, 2, size=(5, D))
    _compare(g["site_covs"], g["obs_covs"], g["obs"], th)


def test_headline_config2(cfg2_data):
    data, truth = cfg2_data
    th = np.random.default_rng(1).uniform(-2, 2, size=(3, 8))
    th = np.vstack([th, np.concatenate([truth["beta"][0], truth["alpha"][0]])])
    _compare(data["site_covs"], data["obs_covs"], data["obs"], th)


@pytest.mark.parametrize("n_sites", [1, 2, 63, 64, 65, 511, 513, 1025, 4099])
def test_ragged_site_counts(n_sites):
    rng = np.random.default_rng(n_sites)
    X = rng.normal(size=(n_sites, 2)); W = rng.normal(size=(n_sites, 2, 3, 2))
    Y = (rng.uniform(size=(1, n_sites, 2, 3)) < 0.3) * 1.0
    Y[rng.uniform(size=Y.shape) < 0.1] = np.nan
    _compare(X, W, Y, rng.uniform(-1.5, 1.5, size=(2, 6)))


@pytest.mark.parametrize("ks,ko", [(0, 0), (0, 3), (4, 0), (1, 4), (5, 2), (7, 8), (9, 5), (16, 16), (3, 11)])
def test_covariate_counts_and_padding(ks, ko):
    """Ks/Ko outside {1,2,3,4} run on padded kernel capacities (8, 16); K=0 is intercept-only."""
    rng = np.random.default_rng(100 * ks + ko)
    N, T, J = 200, 1, 4
    X = rng.normal(size=(N, ks)) * 0.5; W = rng.normal(size=(N, T, J, ko)) * 0.5
    Y = (rng.uniform(size=(1, N, T, J)) < 0.3) * 1.0
    _compare(X, W, Y, rng.uniform(-1, 1, size=(2, ks + ko + 2)) * 0.7)


def test_many_visits_goes_unstaged_and_still_matches():
    """A slice too large for 160 KB of LDS takes the HBM-row path."""
    data, _, _ = quiet_simulate(n_sites=12800, n_site_covs=2, n_obs_covs=1, deployment_days_per_site=90 * 7, random_seed=49)
    th = np.random.default_rng(7).uniform(-1, 1, size=(2, 5))
    _compare(data["site_covs"], data["obs_covs"], data["obs"], th)


def test_missing_data_edge_cases():
    rng = np.random.default_rng(9)
    N, T, J = 130, 2, 5
    X = rng.normal(size=(N, 3)); W = rng.normal(size=(N, T, J, 2)); Y = (rng.uniform(size=(1, N, T, J)) < 0.35) * 1.0
    X[::7, 1] = np.nan            # whole site masked (occu.py:138)
    W[::5, 1, 2, 0] = np.nan      # single visit masked (occu.py:137)
    Y[0, ::3, 0, :] = np.nan      # a period with no visits at all
    Y[0, 5] = np.nan              # a site with no data
    Y[0, 6] = 1.0                 # detections at every visit
    Y[0, 8] = 0.0                 # never detected
    th = rng.uniform(-2, 2, size=(3, 7))
    _compare(X, W, Y, th)
    _compare(X, W, np.full_like(Y, np.nan), th)  # nothing observed: prior only


def test_extreme_parameters_stay_finite_and_match():
    g = load_golden("small_3x3")
    th = np.array([[8.0, -6, 5, 7, 9, -8, 6, 5], [-9, 3, 3, 3, -12, 4, 4, 4], [0, 0, 0, 0, 30, 0, 0, 0], [-40, 0, 0, 0, -40, 0, 0, 0.0]])
    _compare(g["site_covs"], g["obs_covs"], g["obs"], th)


def test_prior_location_and_scale():
    g = load_golden("seed7_2x1")
    th = np.random.default_rng(3).uniform(-1, 1, size=(2, 5))
    _compare(g["site_covs"], g["obs_covs"], g["obs"], th, priors=((0.5, 2.0), (-0.25, 0.3)))


def test_linearity_in_site_blocks(cfg2_data):
    """Size-independent property at full size: the log-likelihood is additive over sites, so
    U(all) - prior = sum over disjoint site blocks of (U(block) - prior)."""
    data, _ = cfg2_data
    th = np.random.default_rng(5).uniform(-1, 1, size=(1, 8)).astype(np.float32).astype(np.float64)
    prior = 0.5 * np.sum(th ** 2) + 8 * 0.9189385332046727
    full = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"]).logp_grad(th)
    parts_U, parts_G = 0.0, 0.0
    for lo, hi in [(0, 3333), (3333, 7001), (7001, 10000)]:
        U, G = OccuDataset(data["site_covs"][lo:hi], data["obs_covs"][lo:hi], data["obs"][:, lo:hi]).logp_grad(th)
        parts_U += U[0] - prior
        parts_G = parts_G + (G[0] - th[0])
    assert abs((full[0][0] - prior) - parts_U) <= 1e-7 * abs(full[0][0])
    assert np.max(np.abs((full[1][0] - th[0]) - parts_G)) <= 1e-6 * np.max(np.abs(full[1][0]))
