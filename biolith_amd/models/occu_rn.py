"""``occu_rn`` / ``simulate_rn`` -- host-side mirror of biolith/models/occu_rn.py for the HIP engine.

Royle & Nichols (2003): latent abundance ``N_it ~ RightTruncatedPoisson(exp(beta0 + x beta), max_abundance)``
(enumerated; utils/distributions.py:6-40), detection ``y_itj ~ Bernoulli(1 - (1 - r_itj)^N_it)`` with
``r = sigmoid(alpha0 + w alpha)`` (occu_rn.py:194-222).  As with :func:`occu`, the callable keeps the
reference's signature (occu_rn.py:20-40), validates, and resolves to an :class:`OccuSpec` that
``fit`` lowers onto the C-ABI; the marginal density, gradient and sampler run in gfx950 kernels.
"""
from __future__ import annotations

from typing import Any

import numpy as np

from ..distributions import Beta, HalfNormal, Normal, as_beta, as_half_normal, as_normal
from ..regression import LinearRegression
from ._generators import Generator, expit, observed_mean, within
from .occu import OccuSpec

MAX_ABUNDANCE_LIMIT = 127  # the kernel's per-lane table over N (BL_RN_NB - 1)
MAX_RN_COVS = 16  # covariates per side (the engine's BL_MAX_COVS; every model is instantiated at every capacity)


def occu_rn(
    site_covs,
    obs_covs,
    coords=None,
    ell: float = 1.0,
    false_positives_constant: bool = False,
    max_abundance: int = 100,
    obs=None,
    n_species: int = 1,
    prior_beta: Any = Normal(),
    prior_alpha: Any = Normal(),
    regressor_abu=LinearRegression,
    regressor_det=LinearRegression,
    prior_prob_fp_constant: Any = Beta(2, 5),
    prior_gp_sd: Any = HalfNormal(1.0),
    prior_gp_length: Any = HalfNormal(1.0),
    site_random_effects: bool = False,
    obs_random_effects: bool = False,
    prior_site_re_sd: Any = HalfNormal(1.0),
    prior_obs_re_sd: Any = HalfNormal(1.0),
) -> OccuSpec:
    """Royle-Nichols abundance-occupancy model on the HIP engine (parameters: occu_rn.py:20-40).

    Built: linear regressors, Normal priors, no spatial effect; ``false_positives_constant`` (occu_rn.py:133-138, 214-221; Beta prior),
    ``site_random_effects`` /
    ``obs_random_effects`` with HalfNormal priors on their sds (occu_rn.py:151-154, 172-184, 199-212) run on the random-effects
    kernels, one species per fit (the sds are sampled outside the species plate).  Everything else raises ``NotImplementedError``.

    Examples
    --------
    >>> from biolith_amd.models import occu_rn, simulate_rn
    >>> from biolith_amd.utils import fit
    >>> data, _ = simulate_rn()
    >>> results = fit(occu_rn, **data)
    >>> print(results.samples['abundance'].mean())
    """
    site_covs = np.asarray(site_covs, dtype=np.float32)
    obs_covs = np.asarray(obs_covs, dtype=np.float32)
    obs = None if obs is None else np.asarray(obs, dtype=np.float32)
    # occu_rn.py:101-108
    assert obs is None or obs.ndim == 4, "obs must be None or of shape (n_species, n_sites, n_periods, n_replicates)"
    assert site_covs.ndim == 2, "site_covs must be (n_sites, n_site_covs)"
    assert obs_covs.ndim == 4, "obs_covs must be (n_sites, n_periods, n_replicates, n_obs_covs)"
    if obs is not None:
        n_species = obs.shape[0]
    unsupported = []
    if coords is not None:
        unsupported.append("coords (spatial effect, occu_rn.py:140-148)")
    if false_positives_constant and obs is not None and n_species > 1:
        unsupported.append("false positives with several species (the rate is shared across the species plate, occu_rn.py:133-138)")
    if (site_random_effects or obs_random_effects) and obs is not None and n_species > 1:
        unsupported.append("random effects with several species (the sds are shared across the species plate, occu_rn.py:151-154)")
    if regressor_abu is not LinearRegression or regressor_det is not LinearRegression:
        unsupported.append("non-linear regressors (occu_rn.py:166-167)")
    if obs is None:
        unsupported.append("obs=None (prior predictive)")
    if not 1 <= int(max_abundance) <= MAX_ABUNDANCE_LIMIT:
        unsupported.append(f"max_abundance={max_abundance} (1..{MAX_ABUNDANCE_LIMIT})")
    if site_covs.shape[1] > MAX_RN_COVS or obs_covs.shape[3] > MAX_RN_COVS:
        unsupported.append(f"more than {MAX_RN_COVS} covariates per side for occu_rn")
    if unsupported:
        raise NotImplementedError(
            "biolith_amd.occu_rn runs the default-option Royle-Nichols path on the HIP engine; not built: "
            + "; ".join(unsupported)
        )
    if obs.shape[1:] != obs_covs.shape[:3] or site_covs.shape[0] != obs_covs.shape[0]:
        raise ValueError("site_covs, obs_covs and obs disagree on (n_sites, n_periods, n_replicates)")
    spec = OccuSpec(site_covs, obs_covs, obs, n_species, as_normal(prior_beta, "prior_beta"),
                    as_normal(prior_alpha, "prior_alpha"), model="occu_rn",
                    extras=dict(max_abundance=int(max_abundance)))
    if false_positives_constant:   # occu_rn.py:133-138, 214-221: runs on the random-effects kernels, with or without effects
        spec.extras.update(re_fp_mode="constant", prior_fp=as_beta(prior_prob_fp_constant, "prior_prob_fp_constant"))
    if site_random_effects or obs_random_effects or false_positives_constant:
        spec.extras.update(site_random_effects=bool(site_random_effects), obs_random_effects=bool(obs_random_effects),
                           prior_site_re_sd=as_half_normal(prior_site_re_sd, "prior_site_re_sd"),
                           prior_obs_re_sd=as_half_normal(prior_obs_re_sd, "prior_obs_re_sd"))
    return spec


occu_rn.__biolith_amd_model__ = "occu_rn"


def simulate_rn(
    n_site_covs: int = 1,
    n_obs_covs: int = 1,
    n_sites: int = 100,
    n_periods: int = 1,
    n_species: int = 1,
    deployment_days_per_site: int = 365,
    session_duration: int = 7,
    prob_fp: float = 0.0,
    simulate_missing: bool = False,
    min_occupancy: float = 0.25,
    max_occupancy: float = 0.75,
    min_observation_rate: float = 0.1,
    max_observation_rate: float = 0.5,
    random_seed: int = 0,
    spatial: bool = False,
    gp_sd: float = 1.0,
    gp_l: float = 0.2,
):
    """Synthetic dataset for :func:`occu_rn`; returns ``(data, true_params)`` (occu_rn.py:225-358).

    Bit-identical to the reference for equal arguments: per rejection-loop iteration the PCG64
    calls are beta, alpha, site_covs, Poisson N, obs_covs, binomial obs, [3 missingness masks].
    """
    if spatial:
        raise NotImplementedError("simulate_rn(spatial=True) is outside the built path (utils/spatial.py:52-76)")

    def latent(rng, abu_linear):  # N ~ Poisson(abundance) per period (occu_rn.py:296-303)
        return rng.poisson(np.exp(abu_linear)[:, None, :], size=(n_species, n_periods, n_sites))

    def observe(rng, det_linear, N_site, _):  # occu_rn.py:310-330
        p_it = 1.0 - (1.0 - expit(det_linear)) ** N_site[..., None]
        obs = rng.binomial(n=1, p=1 - (1 - p_it) * (1 - prob_fp), size=(n_species, n_sites, n_periods, n_replicates))
        return (obs >= 1) * 1.0

    def accept(d):  # negation of the reference's while-condition (occu_rn.py:272-278)
        return (within((d.latent > 0).mean(), min_occupancy, max_occupancy)
                and within(observed_mean(d.obs), min_observation_rate, max_observation_rate))

    n_replicates = round(deployment_days_per_site / session_duration)
    d = Generator(n_species, n_sites, n_periods, n_replicates, n_site_covs, n_obs_covs, latent, observe, accept,
                  simulate_missing=simulate_missing).run(random_seed)
    abundance = np.exp(d.site_linear)
    print(f"True occupancy: {np.mean(d.latent > 0):.4f}")
    print(f"True abundance: {np.mean(abundance):.4f}")
    print(f"Proportion of timesteps with observation: {np.mean(d.obs[np.isfinite(d.obs)]):.4f}")
    data = dict(site_covs=d.site_covs, obs_covs=d.obs_covs, obs=d.obs, coords=None, ell=0.0)
    return data, dict(abundance=abundance, beta=d.beta, alpha=d.alpha, w=d.extra["w"], gp_sd=gp_sd, gp_l=gp_l)
