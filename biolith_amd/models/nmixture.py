"""``nmixture`` / ``simulate_nmixture`` -- host-side mirror of biolith/models/nmixture.py for the HIP engine.

N-mixture model for repeated counts (Royle 2004): abundance ``lambda_i = exp(beta0 + x beta)``, latent
``N_it`` enumerated over ``0..max_abundance`` with Poisson weights cut off below the largest count of the
(site, period) (nmixture.py:150-155, 183-196), counts ``y_itj ~ Binomial(N_it, sigmoid(alpha0 + w alpha))``
(nmixture.py:206-220).  The callable keeps the reference's signature (nmixture.py:17-35), validates, and
resolves to an :class:`OccuSpec`; the N-marginalised density, its gradient and the sampler run in gfx950 kernels.
"""
from __future__ import annotations

from typing import Any

import numpy as np

from ..distributions import HalfNormal, Normal, as_half_normal, as_normal
from ..regression import LinearRegression
from ._generators import Generator, expit, observed_mean, within
from .occu import OccuSpec

MAX_ABUNDANCE_LIMIT = 127  # the kernel's lgamma table over N
MAX_NMIX_COVS = 16  # covariates per side (the engine's BL_MAX_COVS; every model is instantiated at every capacity)


def nmixture(
    site_covs,
    obs_covs,
    coords=None,
    ell: float = 1.0,
    max_abundance: int = 100,
    obs=None,
    n_species: int = 1,
    prior_beta: Any = Normal(),
    prior_alpha: Any = Normal(),
    regressor_abu=LinearRegression,
    regressor_det=LinearRegression,
    prior_gp_sd: Any = HalfNormal(1.0),
    prior_gp_length: Any = HalfNormal(1.0),
    site_random_effects: bool = False,
    obs_random_effects: bool = False,
    prior_site_re_sd: Any = HalfNormal(1.0),
    prior_obs_re_sd: Any = HalfNormal(1.0),
) -> OccuSpec:
    """N-mixture model on the HIP engine (parameters: nmixture.py:17-35).

    Built: linear regressors, Normal priors, no spatial effect, ``max_abundance`` <= 127; several species are sampled species by
    species.  ``site_random_effects`` / ``obs_random_effects`` with HalfNormal priors on their sds (nmixture.py:139-141, 166-172,
    199-214) run on the random-effects kernels, one species per fit (the sds are sampled outside the species plate, so several
    species would share them under one chain -- not built for the count model).  Everything else raises ``NotImplementedError``.

    Examples
    --------
    >>> from biolith_amd.models import nmixture, simulate_nmixture
    >>> from biolith_amd.utils import fit
    >>> data, _ = simulate_nmixture()
    >>> results = fit(nmixture, **data)
    >>> print(results.samples['abundance'].mean())
    """
    site_covs = np.asarray(site_covs, dtype=np.float32)
    obs_covs = np.asarray(obs_covs, dtype=np.float32)
    obs = None if obs is None else np.asarray(obs, dtype=np.float32)
    # nmixture.py:85-115
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
        unsupported.append("coords (spatial HSGP effect, nmixture.py:125-133)")
    if (site_random_effects or obs_random_effects) and obs is not None and n_species > 1:
        unsupported.append("random effects with several species (the sds are shared across the species plate, nmixture.py:139-141)")
    if regressor_abu is not LinearRegression or regressor_det is not LinearRegression:
        unsupported.append("non-linear regressors (nmixture.py:160-161)")
    if obs is None:
        unsupported.append("obs=None (prior predictive)")
    if not 1 <= int(max_abundance) <= MAX_ABUNDANCE_LIMIT:
        unsupported.append(f"max_abundance outside 1..{MAX_ABUNDANCE_LIMIT}")
    if site_covs.shape[1] > MAX_NMIX_COVS or obs_covs.shape[3] > MAX_NMIX_COVS:
        unsupported.append(f"more than {MAX_NMIX_COVS} covariates per side")
    if unsupported:
        raise NotImplementedError("biolith_amd.nmixture: not built: " + "; ".join(unsupported))
    spec = OccuSpec(site_covs, obs_covs, obs, n_species, as_normal(prior_beta, "prior_beta"),
                    as_normal(prior_alpha, "prior_alpha"), model="nmixture")
    spec.extras["max_abundance"] = int(max_abundance)
    if site_random_effects or obs_random_effects:
        spec.extras.update(site_random_effects=bool(site_random_effects), obs_random_effects=bool(obs_random_effects),
                           prior_site_re_sd=as_half_normal(prior_site_re_sd, "prior_site_re_sd"),
                           prior_obs_re_sd=as_half_normal(prior_obs_re_sd, "prior_obs_re_sd"))
    return spec


nmixture.__biolith_amd_model__ = "nmixture"


def simulate_nmixture(
    n_site_covs: int = 1,
    n_obs_covs: int = 1,
    n_sites: int = 100,
    n_periods: int = 1,
    n_species: int = 1,
    deployment_days_per_site: int = 365,
    session_duration: int = 7,
    simulate_missing: bool = False,
    min_abundance: float = 0.5,
    max_abundance: float = 6.0,
    min_observation_rate: float = 0.5,
    max_observation_rate: float = 4.0,
    random_seed: int = 0,
    spatial: bool = False,
    gp_sd: float = 1.0,
    gp_l: float = 0.2,
    site_random_effects: bool = False,
    obs_random_effects: bool = False,
    site_re_sd: float = 0.5,
    obs_re_sd: float = 0.3,
):
    """Generator of :func:`nmixture` data, bit-identical to the reference's (nmixture.py:223-369) for
    ``spatial=False`` (random effects included: nmixture.py:285-310): same NumPy PCG64 stream, draw order and rejection loop.

    Examples
    --------
    >>> from biolith_amd.models import simulate_nmixture
    >>> data, params = simulate_nmixture()
    >>> sorted(data.keys())
    ['coords', 'ell', 'obs', 'obs_covs', 'site_covs']
    """
    if spatial:
        raise NotImplementedError("simulate_nmixture(spatial=True) is outside the built path (utils/spatial.py:52-76)")

    def latent(rng, abu_linear):  # nmixture.py:295-296
        return rng.poisson(np.exp(abu_linear)[:, None, :], size=(n_species, n_periods, n_sites))

    def observe(rng, det_linear, N_site, _):  # nmixture.py:322-323
        return rng.binomial(n=N_site[..., None], p=expit(det_linear)).astype(float)

    def accept(d):  # negation of the reference's while-condition (nmixture.py:262-268)
        return (within(np.mean(d.latent), min_abundance, max_abundance)
                and within(observed_mean(d.obs), min_observation_rate, max_observation_rate))

    n_replicates = round(deployment_days_per_site / session_duration)
    d = Generator(n_species, n_sites, n_periods, n_replicates, n_site_covs, n_obs_covs, latent, observe, accept,
                  site_re_sd=site_re_sd if site_random_effects else None, obs_re_sd=obs_re_sd if obs_random_effects else None,
                  simulate_missing=simulate_missing).run(random_seed)
    print(f"True abundance: {np.mean(d.latent):.4f}")
    print(f"Mean count: {np.mean(d.obs[np.isfinite(d.obs)]):.4f}")
    true_params = dict(N_i=d.latent, abundance=np.exp(d.site_linear), beta=d.beta, alpha=d.alpha, w=d.extra["w"],
                       gp_sd=gp_sd, gp_l=gp_l)
    if site_random_effects:   # nmixture.py:348-356
        true_params.update(site_re_abu=d.extra["site_re_a"], site_re_det=d.extra["site_re_b"], site_re_sd=site_re_sd)
    if obs_random_effects:    # nmixture.py:357-363
        true_params.update(obs_re=d.extra["obs_re"], obs_re_sd=obs_re_sd)
    return dict(site_covs=d.site_covs, obs_covs=d.obs_covs, obs=d.obs, coords=None, ell=0.0), true_params
