"""``occu_cs`` / ``simulate_cs`` -- host-side mirror of biolith/models/occu_cs.py for the HIP engine.

The continuous-score occupancy model (Rhinehart et al. 2022): every replicate carries a classifier score
``s ~ Normal(mu_f, sigma_f)`` from the false-positive (f = 0) or the true-positive (f = 1) score distribution, with
``f ~ Bernoulli(z p)`` and ``z ~ Bernoulli(psi)`` both summed out (occu_cs.py:196-232).  ``occu_cs`` keeps the
reference's signature (occu_cs.py:17-36), validates like it (occu_cs.py:100-117) and resolves to an ``OccuSpec``;
``simulate_cs`` is the reference's generator (occu_cs.py:222-361) on the shared scaffold, bit-identical (tests/golden).
"""
from __future__ import annotations

from typing import Any

import numpy as np

from ..distributions import Gamma, HalfNormal, Normal, as_gamma, as_normal
from ..regression import LinearRegression
from ._generators import Generator, expit, within
from .occu import OccuSpec


def occu_cs(
    site_covs,
    obs_covs,
    coords=None,
    ell: float = 1.0,
    obs=None,
    n_species: int = 1,
    prior_beta: Any = Normal(),
    prior_alpha: Any = Normal(),
    regressor_occ=LinearRegression,
    regressor_det=LinearRegression,
    prior_mu: Any = Normal(0, 10),
    prior_sigma: Any = Gamma(5, 1),
    prior_gp_sd: Any = HalfNormal(1.0),
    prior_gp_length: Any = HalfNormal(1.0),
    site_random_effects: bool = False,
    obs_random_effects: bool = False,
    prior_site_re_sd: Any = HalfNormal(1.0),
    prior_obs_re_sd: Any = HalfNormal(1.0),
) -> OccuSpec:
    """Continuous-score occupancy model, z and f marginalised, on the HIP engine.

    Same parameters as the reference (biolith/models/occu_cs.py:17-36).  Built: the default option path -- linear regressors,
    Normal / Laplace coefficient priors, ``prior_mu`` Normal (one, or a pair for mu0 and for the base of mu1, which is
    truncated below at mu0, occu_cs.py:146-148), ``prior_sigma`` Gamma (one or a pair), one species (mu and sigma are
    sampled outside the species plate).  Anything else raises ``NotImplementedError``.
    """
    site_covs = np.asarray(site_covs, dtype=np.float32)
    obs_covs = np.asarray(obs_covs, dtype=np.float32)
    obs = None if obs is None else np.asarray(obs, dtype=np.float32)
    # occu_cs.py:100-117
    assert obs is None or obs.ndim == 4, "obs must be None or of shape (n_species, n_sites, n_periods, n_replicates)"
    assert site_covs.ndim == 2, "site_covs must be of shape (n_sites, n_site_covs)"
    assert obs_covs.ndim == 4, "obs_covs must be of shape (n_sites, n_periods, n_replicates, n_obs_covs)"
    n_sites, n_periods, n_replicates = site_covs.shape[0], obs_covs.shape[1], obs_covs.shape[2]
    if obs is not None:
        n_species = obs.shape[0]
    assert n_sites == obs_covs.shape[0], "site_covs and obs_covs must have the same number of sites"
    if obs is not None:
        assert n_sites == obs.shape[1], "obs must have n_sites rows"
        assert n_periods == obs.shape[2], "obs must have n_periods columns"
        assert n_replicates == obs.shape[3], "obs must have n_replicates columns"
    unsupported = []
    if coords is not None:
        unsupported.append("coords (spatial HSGP effect)")
    if site_random_effects or obs_random_effects:
        unsupported.append("random effects")
    if regressor_occ is not LinearRegression or regressor_det is not LinearRegression:
        unsupported.append("non-linear regressors")
    if obs is None:
        unsupported.append("obs=None (prior predictive)")
    if n_species != 1:
        unsupported.append("n_species > 1 (mu0, mu1, sigma0, sigma1 are shared across species, occu_cs.py:143-152)")
    if site_covs.shape[1] > 16 or obs_covs.shape[3] > 16:
        unsupported.append("more than 16 covariates per side")
    if unsupported:
        raise NotImplementedError("biolith_amd.occu_cs: not built: " + "; ".join(unsupported))
    mus = prior_mu if isinstance(prior_mu, tuple) else (prior_mu, prior_mu)
    sigmas = prior_sigma if isinstance(prior_sigma, tuple) else (prior_sigma, prior_sigma)
    spec = OccuSpec(site_covs, obs_covs, obs, n_species, as_normal(prior_beta, "prior_beta"), as_normal(prior_alpha, "prior_alpha"),
                    model="occu_cs")
    prior_mus = tuple(as_normal(p, "prior_mu") for p in mus)
    if any(p.family != "normal" for p in prior_mus):
        raise NotImplementedError("prior_mu: Normal(loc, scale) only")
    spec.extras.update(prior_mu=tuple(tuple(p) for p in prior_mus), prior_sigma=tuple(as_gamma(p, "prior_sigma") for p in sigmas))
    return spec


occu_cs.__biolith_amd_model__ = "occu_cs"


def simulate_cs(
    n_site_covs: int = 1,
    n_obs_covs: int = 1,
    n_sites: int = 100,
    n_periods: int = 1,
    n_species: int = 1,
    deployment_days_per_site: int = 365,
    session_duration: int = 7,
    simulate_missing: bool = False,
    min_occupancy: float = 0.25,
    max_occupancy: float = 0.75,
    random_seed: int = 0,
    spatial: bool = False,
    gp_sd: float = 1.0,
    gp_l: float = 0.2,
):
    """Synthetic dataset for :func:`occu_cs`; returns ``(data, true_params)`` (occu_cs.py:222-361), bit-identical to the
    reference for equal arguments.  ``spatial=True`` is outside the built path."""
    if spatial:
        raise NotImplementedError("simulate_cs(spatial=True): the spatial effect is not built")
    mu0, sigma0, mu1, sigma1 = 0, 10, 10, 5   # occu_cs.py:311-314

    def latent(rng, occ_linear):  # occu_cs.py:283-285
        return rng.binomial(n=1, p=expit(occ_linear)[:, None, :], size=(n_species, n_periods, n_sites))

    def observe(rng, det_linear, z_site, _):  # occu_cs.py:320-330
        shape = (n_species, n_sites, n_periods, n_replicates)
        f = rng.binomial(n=1, p=expit(det_linear) * z_site[..., None], size=shape)
        return rng.normal(loc=np.where(f == 1, mu1, mu0), scale=np.where(f == 1, sigma1, sigma0), size=shape)

    def accept(d):  # occu_cs.py:263
        return within(d.latent.mean(), min_occupancy, max_occupancy)

    n_replicates = round(deployment_days_per_site / session_duration)
    d = Generator(n_species, n_sites, n_periods, n_replicates, n_site_covs, n_obs_covs, latent, observe, accept,
                  simulate_missing=simulate_missing).run(random_seed)
    print(f"True occupancy: {np.mean(d.latent):.4f}")
    return dict(site_covs=d.site_covs, obs_covs=d.obs_covs, obs=d.obs, coords=None, ell=0.0), \
        dict(z=d.latent, beta=d.beta, alpha=d.alpha, mu0=mu0, sigma0=sigma0, mu1=mu1, sigma1=sigma1, w=d.extra["w"], gp_sd=gp_sd, gp_l=gp_l)
