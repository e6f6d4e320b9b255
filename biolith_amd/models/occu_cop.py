"""``occu_cop`` / ``simulate_cop`` -- host-side mirror of biolith/models/occu_cop.py for the HIP engine.

Count occupancy model (Pautrel et al. 2024): ``y_itj ~ Poisson(session_duration_itj * (z_it lambda_itj +
(1 - z_it) rate_fp_unoccupied + rate_fp_constant))`` with ``lambda = exp(alpha0 + w alpha)`` and
``z ~ Bernoulli(sigmoid(beta0 + x beta))`` enumerated (occu_cop.py:206-255).  As with :func:`occu`, the
callable keeps the reference's signature (occu_cop.py:17-39), validates, and resolves to an
:class:`OccuSpec`; the z-marginalised density, its gradient and the sampler run in gfx950 kernels.
"""
from __future__ import annotations

from typing import Any

import numpy as np

from ..distributions import Exponential, HalfNormal, Normal, as_exponential, as_half_normal, as_normal
from ..regression import LinearRegression
from ._generators import Generator, expit, observed_mean, within
from .occu import OccuSpec

MAX_COP_COVS = 16  # covariates per side (the engine's BL_MAX_COVS; every model is instantiated at every capacity)


def occu_cop(
    site_covs,
    obs_covs,
    coords=None,
    ell: float = 1.0,
    session_duration=None,
    false_positives_constant: bool = False,
    false_positives_unoccupied: bool = False,
    obs=None,
    n_species: int = 1,
    prior_beta: Any = Normal(),
    prior_alpha: Any = Normal(),
    regressor_occ=LinearRegression,
    regressor_det=LinearRegression,
    prior_rate_fp_constant: Any = Exponential(),
    prior_rate_fp_unoccupied: Any = Exponential(),
    prior_gp_sd: Any = HalfNormal(1.0),
    prior_gp_length: Any = HalfNormal(1.0),
    site_random_effects: bool = False,
    obs_random_effects: bool = False,
    prior_site_re_sd: Any = HalfNormal(1.0),
    prior_obs_re_sd: Any = HalfNormal(1.0),
) -> OccuSpec:
    """Count occupancy model with a Poisson detection process on the HIP engine (parameters: occu_cop.py:17-39).

    Built: linear regressors, Normal priors, ``false_positives_constant`` / ``false_positives_unoccupied`` with an
    Exponential prior on the rate, no spatial effect; one species when a false-positive rate is sampled (it is shared across species,
    occu_cop.py:158-170); ``site_random_effects`` / ``obs_random_effects`` (occu_cop.py:183-186, 204-210, 229-243) on the random-effects
    kernels, one species, with or without a false-positive rate.  Everything else raises ``NotImplementedError``.

    Examples
    --------
    >>> from biolith_amd.models import occu_cop, simulate_cop
    >>> from biolith_amd.utils import fit
    >>> data, _ = simulate_cop()
    >>> results = fit(occu_cop, **data)
    >>> print(results.samples['psi'].mean())
    """
    site_covs = np.asarray(site_covs, dtype=np.float32)
    obs_covs = np.asarray(obs_covs, dtype=np.float32)
    obs = None if obs is None else np.asarray(obs, dtype=np.float32)
    session_duration = None if session_duration is None else np.asarray(session_duration, dtype=np.float32)
    # occu_cop.py:101-144
    assert obs is None or obs.ndim == 4, "obs must be None or of shape (n_species, n_sites, n_periods, n_replicates)"
    assert site_covs.ndim == 2, "site_covs must be of shape (n_sites, n_site_covs)"
    assert obs_covs.ndim == 4, "obs_covs must be of shape (n_sites, n_periods, n_replicates, n_obs_covs)"
    assert session_duration is None or session_duration.ndim == 3, \
        "session_duration must be None or of shape (n_sites, n_periods, n_replicates)"
    assert not (false_positives_constant and false_positives_unoccupied), \
        "false_positives_constant and false_positives_unoccupied cannot both be True"
    n_sites, n_periods, n_replicates = site_covs.shape[0], obs_covs.shape[1], obs_covs.shape[2]
    if obs is not None:
        n_species = obs.shape[0]
    assert n_sites == obs_covs.shape[0], "site_covs and obs_covs must have the same number of sites"
    if obs is not None:
        assert n_sites == obs.shape[1], "obs must have n_sites rows"
        assert n_periods == obs.shape[2], "obs must have n_periods columns"
        assert n_replicates == obs.shape[3], "obs must have n_replicates columns"
    if session_duration is not None:
        assert n_sites == session_duration.shape[0], "session_duration must have n_sites rows"
        assert n_periods == session_duration.shape[1], "session_duration must have n_periods columns"
        assert n_replicates == session_duration.shape[2], "session_duration must have n_replicates columns"
    else:  # occu_cop.py:146-148: a constant duration of 1
        session_duration = np.ones((n_sites, n_periods, n_replicates), dtype=np.float32)

    fp_mode = "constant" if false_positives_constant else ("unoccupied" if false_positives_unoccupied else None)
    unsupported = []
    if coords is not None:
        unsupported.append("coords (spatial HSGP effect, occu_cop.py:172-180)")
    if (site_random_effects or obs_random_effects) and obs is not None and n_species > 1:
        unsupported.append("random effects with several species (the sds are shared across the species plate, occu_cop.py:183-186)")
    if regressor_occ is not LinearRegression or regressor_det is not LinearRegression:
        unsupported.append("non-linear regressors (occu_cop.py:199-200)")
    if obs is None:
        unsupported.append("obs=None (prior predictive)")
    if fp_mode is not None and n_species != 1:
        unsupported.append("false positives with n_species > 1 (the rate is shared across species, occu_cop.py:158-170)")
    if site_covs.shape[1] > MAX_COP_COVS or obs_covs.shape[3] > MAX_COP_COVS:
        unsupported.append(f"more than {MAX_COP_COVS} covariates per side")
    if unsupported:
        raise NotImplementedError("biolith_amd.occu_cop: not built: " + "; ".join(unsupported))
    spec = OccuSpec(site_covs, obs_covs, obs, n_species, as_normal(prior_beta, "prior_beta"),
                    as_normal(prior_alpha, "prior_alpha"), model="occu_cop")
    spec.extras.update(session_duration=session_duration, fp_mode=fp_mode)
    if site_random_effects or obs_random_effects:   # occu_cop.py:183-186, 204-210, 229-243: on the random-effects kernels
        spec.extras.update(site_random_effects=bool(site_random_effects), obs_random_effects=bool(obs_random_effects),
                           prior_site_re_sd=as_half_normal(prior_site_re_sd, "prior_site_re_sd"),
                           prior_obs_re_sd=as_half_normal(prior_obs_re_sd, "prior_obs_re_sd"))
    if fp_mode is not None:
        prior = prior_rate_fp_constant if fp_mode == "constant" else prior_rate_fp_unoccupied
        spec.extras["prior_fp_rate"] = as_exponential(prior, f"prior_rate_fp_{fp_mode}")
    return spec


occu_cop.__biolith_amd_model__ = "occu_cop"


def simulate_cop(
    n_site_covs: int = 1,
    n_obs_covs: int = 1,
    n_sites: int = 100,
    n_species: int = 1,
    n_periods: int = 1,
    deployment_days_per_site: int = 365,
    session_duration: int = 7,
    simulate_missing: bool = False,
    min_occupancy: float = 0.25,
    max_occupancy: float = 0.75,
    min_observation_rate: float = 0.5,
    max_observation_rate: float = 10.0,
    random_seed: int = 0,
    spatial: bool = False,
    gp_sd: float = 1.0,
    gp_l: float = 0.2,
):
    """Generator of :func:`occu_cop` data, bit-identical to the reference's (occu_cop.py:258-396) for
    ``spatial=False``: same NumPy PCG64 stream, same draw order, same rejection loop.  Returns ``(data, true_params)``.

    Examples
    --------
    >>> from biolith_amd.models import simulate_cop
    >>> data, params = simulate_cop()
    >>> sorted(data.keys())
    ['coords', 'ell', 'false_positives_constant', 'obs', 'obs_covs', 'session_duration', 'site_covs']
    """
    if spatial:
        raise NotImplementedError("simulate_cop(spatial=True): the spatial HSGP effect is not built")

    def leading(rng):  # the false-positive rate is drawn first in every pass (occu_cop.py:305)
        return dict(rate_fp=rng.uniform(0.05, 0.2))

    def latent(rng, occ_linear):  # occu_cop.py:329-331
        return rng.binomial(n=1, p=expit(occ_linear)[:, None, :], size=(n_species, n_periods, n_sites))

    def observe(rng, det_linear, z_site, extra):  # occu_cop.py:343-357
        z_site = z_site[..., None]
        lam = session_duration * (np.exp(det_linear) * z_site + extra["rate_fp"] * (1 - z_site))
        return rng.poisson(lam=lam, size=(n_species, n_sites, n_periods, n_replicates)).astype(float)

    def accept(d):  # negation of the reference's while-condition (occu_cop.py:296-302)
        return (within(d.latent.mean(), min_occupancy, max_occupancy)
                and within(observed_mean(d.obs), min_observation_rate, max_observation_rate))

    n_replicates = round(deployment_days_per_site / session_duration)
    d = Generator(n_species, n_sites, n_periods, n_replicates, n_site_covs, n_obs_covs, latent, observe, accept,
                  leading=leading, simulate_missing=simulate_missing).run(random_seed)
    z, obs = d.latent, d.obs
    print(f"True occupancy: {np.mean(z):.4f}")
    print(f"Fraction of observations with at least one observation: {np.mean(obs[np.isfinite(obs)] >= 1):.4f}")
    print(f"Mean rate: {np.mean(obs[np.isfinite(obs)]):.4f}")
    session_duration_arr = np.full((n_sites, n_periods, n_replicates), session_duration)
    return dict(site_covs=d.site_covs, obs_covs=d.obs_covs, session_duration=session_duration_arr, obs=obs,
                false_positives_constant=True, coords=None, ell=0.0), \
        dict(z=z, beta=d.beta, alpha=d.alpha, w=d.extra["w"], gp_sd=gp_sd, gp_l=gp_l)
