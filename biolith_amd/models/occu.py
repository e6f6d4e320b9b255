"""``occu`` / ``simulate`` -- host-side mirror of biolith/models/occu.py for the HIP engine.

In the reference ``occu`` is a NumPyro program that ``fit`` hands to ``NUTS`` (fit.py:93).  Here the
log-density, its gradient and the sampler live in hand-written gfx950 kernels, so ``occu`` keeps the
reference's *signature* (occu.py:19-40) but acts as a validator + dispatch token: calling it checks
the same shape assertions (occu.py:103-133), rejects options outside the built path loudly, and
returns an :class:`OccuSpec` that ``biolith_amd.utils.fit`` lowers onto the C-ABI.

``simulate`` is a NumPy restatement of the reference generator (occu.py:245-430) that must stay
bit-identical to it: tests/golden holds outputs of the reference itself.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np

from ..distributions import Beta, HalfNormal, Normal, as_beta, as_half_normal, as_normal
from ..regression import AbstractRegression, LinearRegression
from ._generators import Generator, expit, observed_mean, within


@dataclass
class OccuSpec:
    """What ``occu(...)`` resolves to: validated arrays + the Normal priors of beta / alpha."""

    site_covs: np.ndarray            # (N, Ks) float32, NaN = missing
    obs_covs: np.ndarray             # (N, T, J, Ko) float32
    obs: Optional[np.ndarray]        # (S, N, T, J) float32 or None
    n_species: int
    prior_beta: tuple
    prior_alpha: tuple
    model: str = "occu"
    extras: dict = field(default_factory=dict)

    @property
    def shape(self):
        N, Ks = self.site_covs.shape
        _, T, J, Ko = self.obs_covs.shape
        return dict(S=self.n_species, N=N, T=T, J=J, Ks=Ks, Ko=Ko)


def occu(
    site_covs,
    obs_covs,
    coords=None,
    ell: float = 1.0,
    false_positives_constant: bool = False,
    false_positives_unoccupied: bool = False,
    obs=None,
    n_species: int = 1,
    prior_beta: Any = Normal(),
    prior_alpha: Any = Normal(),
    regressor_occ=LinearRegression,
    regressor_det=LinearRegression,
    prior_prob_fp_constant: Any = Beta(2, 5),
    prior_prob_fp_unoccupied: Any = Beta(2, 5),
    prior_gp_sd: Any = HalfNormal(1.0),
    prior_gp_length: Any = HalfNormal(1.0),
    site_random_effects: bool = False,
    obs_random_effects: bool = False,
    prior_site_re_sd: Any = HalfNormal(1.0),
    prior_obs_re_sd: Any = HalfNormal(1.0),
) -> OccuSpec:
    """Bernoulli occupancy model (MacKenzie et al. 2002), z marginalised, on the HIP engine.

    Same parameters as the reference (biolith/models/occu.py:19-40).  Supported here: the default
    option path -- linear regressors on both sides, Normal priors, no spatial effect -- plus
    ``site_random_effects`` / ``obs_random_effects`` with HalfNormal priors on their sds (occu.py:170-173, 191-196,
    215-218; not together with false positives) and
    ``false_positives_constant`` / ``false_positives_unoccupied`` with a Beta prior on the
    rate (occu.py:146-157, 229-241); several species are sampled under one chain as in the reference (the ``species`` plate,
    occu.py:182-186, with the false-positive rate / the random effects' sds shared across it).  Anything else raises ``NotImplementedError`` (the
    engine has no silent fallback).  ``coords=None`` / any ``ell`` are accepted and ignored, as the
    reference does when ``coords`` is None (occu.py:159-167); ``simulate()`` returns both.

    Examples
    --------
    >>> from biolith_amd.models import occu, simulate
    >>> from biolith_amd.utils import fit
    >>> data, _ = simulate()
    >>> results = fit(occu, **data)
    >>> print(results.samples['psi'].mean())
    """
    site_covs = np.asarray(site_covs, dtype=np.float32)
    obs_covs = np.asarray(obs_covs, dtype=np.float32)
    obs = None if obs is None else np.asarray(obs, dtype=np.float32)

    # occu.py:103-133
    assert obs is None or obs.ndim == 4, "obs must be None or of shape (n_species, n_sites, n_periods, n_replicates)"
    assert site_covs.ndim == 2, "site_covs must be of shape (n_sites, n_site_covs)"
    assert obs_covs.ndim == 4, "obs_covs must be of shape (n_sites, n_periods, n_replicates, n_obs_covs)"
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

    unsupported = []
    if coords is not None:
        unsupported.append("coords (spatial HSGP effect, occu.py:159-165)")
    fp_mode = "constant" if false_positives_constant else ("unoccupied" if false_positives_unoccupied else None)
    # (prob_fp_* is sampled outside the species plate, occu.py:146-157: several species share it -- fit() then samples all
    # species under one chain)
    if site_random_effects or obs_random_effects:
        # site_re_sd / obs_re_sd are sampled outside the species plate (occu.py:170-173): several species share them -- fit()
        # then samples all species under one chain (up to 8 species)
        if fp_mode is not None and n_species != 1:
            unsupported.append("random effects together with false positives for several species")
    if regressor_occ is not LinearRegression or regressor_det is not LinearRegression:
        unsupported.append("non-linear regressors (occu.py:185-186)")
    if obs is None:
        unsupported.append("obs=None (prior predictive)")
    if unsupported:
        raise NotImplementedError(
            "biolith_amd.occu runs the default-option occupancy path on the HIP engine; not built: "
            + "; ".join(unsupported)
        )
    spec = OccuSpec(site_covs, obs_covs, obs, n_species, as_normal(prior_beta, "prior_beta"),
                    as_normal(prior_alpha, "prior_alpha"))
    if site_random_effects or obs_random_effects:
        spec.model = "occu_re"
        spec.extras.update(site_random_effects=bool(site_random_effects), obs_random_effects=bool(obs_random_effects),
                           prior_site_re_sd=as_half_normal(prior_site_re_sd, "prior_site_re_sd"),
                           prior_obs_re_sd=as_half_normal(prior_obs_re_sd, "prior_obs_re_sd"))
    if fp_mode is not None:
        prior = prior_prob_fp_constant if fp_mode == "constant" else prior_prob_fp_unoccupied
        if spec.model == "occu_re":   # both: the random-effects kernels with the rate as one more replicated coordinate
            spec.extras.update(re_fp_mode=fp_mode, prior_fp=as_beta(prior, f"prior_prob_fp_{fp_mode}"))
        else:
            spec.model = "occu_fp"
            spec.extras.update(fp_mode=fp_mode, prior_fp=as_beta(prior, f"prior_prob_fp_{fp_mode}"))
    return spec


occu.__biolith_amd_model__ = "occu"


def simulate(
    n_site_covs: int = 1,
    n_obs_covs: int = 1,
    n_sites: int = 100,
    n_species: int = 1,
    n_periods: int = 1,
    deployment_days_per_site: int = 365,
    session_duration: int = 7,
    prob_fp_unoccupied: float = 0.0,
    prob_fp_constant: float = 0.0,
    simulate_missing: bool = False,
    min_occupancy: float = 0.25,
    max_occupancy: float = 0.75,
    min_observation_rate: float = 0.1,
    max_observation_rate: float = 0.5,
    random_seed: int = 0,
    spatial: bool = False,
    gp_sd: float = 1.0,
    gp_l: float = 0.2,
    site_random_effects: bool = False,
    obs_random_effects: bool = False,
    site_re_sd: float = 0.5,
    obs_re_sd: float = 0.3,
):
    """Synthetic dataset for :func:`occu`; returns ``(data, true_params)`` (occu.py:245-430).

    The PCG64 draw order of the reference is preserved call for call (beta, alpha, site_covs,
    [site RE], z, obs_covs, [obs RE], obs, [3 missingness masks]) inside its rejection loop, so for
    equal arguments the arrays are bit-identical to the reference's.  ``spatial=True`` (NNGP
    simulator, utils/spatial.py:52-76) is outside the built path.
    """
    if spatial:
        raise NotImplementedError("simulate(spatial=True) is outside the built path (utils/spatial.py:52-76)")

    def latent(rng, occ_linear):  # z ~ Bernoulli(psi) per period (occu.py:333-337)
        return rng.binomial(n=1, p=expit(occ_linear)[:, None, :], size=(n_species, n_periods, n_sites))

    def observe(rng, det_linear, z_site, _):  # occu.py:360-374
        z_site = z_site[..., None]
        prob_detection_fp = 1 - (1 - (z_site * expit(det_linear))) * (1 - prob_fp_constant) * (
            1 - ((1 - z_site) * prob_fp_unoccupied))
        obs = rng.binomial(n=1, p=prob_detection_fp, size=(n_species, n_sites, n_periods, n_replicates))
        return (obs >= 1) * 1.0

    def accept(d):  # negation of the reference's while-condition (occu.py:290-296)
        return (within(d.latent.mean(), min_occupancy, max_occupancy)
                and within(observed_mean(d.obs), min_observation_rate, max_observation_rate))

    n_replicates = round(deployment_days_per_site / session_duration)
    d = Generator(n_species, n_sites, n_periods, n_replicates, n_site_covs, n_obs_covs, latent, observe, accept,
                  site_re_sd=site_re_sd if site_random_effects else None,
                  obs_re_sd=obs_re_sd if obs_random_effects else None,
                  simulate_missing=simulate_missing).run(random_seed)
    z, obs = d.latent, d.obs

    print(f"True occupancy: {np.mean(z):.4f}")
    print(f"Proportion of timesteps with observation: {np.mean(obs[np.isfinite(obs)]):.4f}")

    true_params = dict(z=z, beta=d.beta, alpha=d.alpha, w=d.extra["w"], gp_sd=gp_sd, gp_l=gp_l)
    if site_random_effects:
        true_params.update(site_re_occ=d.extra["site_re_a"], site_re_det=d.extra["site_re_b"], site_re_sd=site_re_sd)
    if obs_random_effects:
        true_params.update(obs_re=d.extra["obs_re"], obs_re_sd=obs_re_sd)
    data = dict(site_covs=d.site_covs, obs_covs=d.obs_covs, obs=obs, coords=None, ell=0.0)
    return data, true_params
