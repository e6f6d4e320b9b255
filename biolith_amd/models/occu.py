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

from ..distributions import Beta, HalfNormal, Normal, as_beta, as_normal
from ..regression import AbstractRegression, LinearRegression


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
    option path -- linear regressors on both sides, Normal priors, no spatial effect, no random
    effects -- plus ``false_positives_constant`` / ``false_positives_unoccupied`` with a Beta prior on the
    rate (occu.py:146-157, 229-241; one species, at most 4 covariates per side); several species are sampled species by species (their joint density
    factorises over the ``species`` plate, occu.py:182-186).  Anything else raises ``NotImplementedError`` (the
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
    if fp_mode is not None:
        # prob_fp_* is sampled outside the species plate (occu.py:146-157): with several species it couples them
        if n_species != 1:
            unsupported.append("false positives with n_species > 1 (the rate is shared across species, occu.py:146-157)")
        if site_covs.shape[1] > 4 or obs_covs.shape[3] > 4:
            unsupported.append("false positives with more than 4 covariates per side")
    if site_random_effects or obs_random_effects:
        unsupported.append("random effects (occu.py:170-173)")
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
    if fp_mode is not None:
        prior = prior_prob_fp_constant if fp_mode == "constant" else prior_prob_fp_unoccupied
        spec.model = "occu_fp"
        spec.extras.update(fp_mode=fp_mode, prior_fp=as_beta(prior, f"prior_prob_fp_{fp_mode}"))
    return spec


occu.__biolith_amd_model__ = "occu"


def _expit_ref(x):
    # written exactly as the reference evaluates it (1 / (1 + exp(-x))) so the Bernoulli draws agree bitwise
    return 1 / (1 + np.exp(-x))


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
    rng = np.random.default_rng(random_seed)
    coords = None
    n_replicates = round(deployment_days_per_site / session_duration)

    def in_range(z, obs):
        occ = z.mean()
        rate = np.mean(obs[np.isfinite(obs)])
        # negation of the reference's while-condition (occu.py:290-296), NaN behaviour included
        return not (occ < min_occupancy or occ > max_occupancy
                    or rate < min_observation_rate or rate > max_observation_rate)

    while True:
        beta = rng.normal(size=(n_species, n_site_covs + 1))
        alpha = rng.normal(size=(n_species, n_obs_covs + 1))
        site_covs = rng.normal(size=(n_sites, n_site_covs))
        w, ell = np.zeros(n_sites), 0.0
        if site_random_effects:
            site_re_occ = rng.normal(0, site_re_sd, size=(n_species, n_sites))
            site_re_det = rng.normal(0, site_re_sd, size=(n_species, n_sites))
        else:
            site_re_occ = np.zeros((n_species, n_sites))
            site_re_det = np.zeros((n_species, n_sites))

        occ_linear = beta[:, 0][:, None] + np.tensordot(beta[:, 1:], site_covs, axes=([1], [1])) + w[None, :] + site_re_occ
        psi = _expit_ref(occ_linear)
        z = rng.binomial(n=1, p=psi[:, None, :], size=(n_species, n_periods, n_sites))

        obs_covs = rng.normal(size=(n_sites, n_periods, n_replicates, n_obs_covs))
        if obs_random_effects:
            obs_re = rng.normal(0, obs_re_sd, size=(n_species, n_sites, n_periods, n_replicates))
        else:
            obs_re = np.zeros((n_species, n_sites, n_periods, n_replicates))
        det_linear = (alpha[:, 0][:, None, None, None] + np.tensordot(alpha[:, 1:], obs_covs, axes=([1], [3]))
                      + site_re_det[:, :, None, None] + obs_re)
        prob_detection = _expit_ref(det_linear)

        z_site = z.transpose(0, 2, 1)[..., None]
        prob_detection_fp = 1 - (1 - (z_site * prob_detection)) * (1 - prob_fp_constant) * (
            1 - ((1 - z_site) * prob_fp_unoccupied))
        obs = rng.binomial(n=1, p=prob_detection_fp, size=(n_species, n_sites, n_periods, n_replicates))
        obs = (obs >= 1) * 1.0

        if simulate_missing:
            obs[rng.choice([True, False], size=obs.shape, p=[0.2, 0.8])] = np.nan
            obs_covs[rng.choice([True, False], size=obs_covs.shape, p=[0.05, 0.95])] = np.nan
            site_covs[rng.choice([True, False], size=site_covs.shape, p=[0.05, 0.95])] = np.nan
        if in_range(z, obs):
            break

    print(f"True occupancy: {np.mean(z):.4f}")
    print(f"Proportion of timesteps with observation: {np.mean(obs[np.isfinite(obs)]):.4f}")

    true_params = dict(z=z, beta=beta, alpha=alpha, w=w, gp_sd=gp_sd, gp_l=gp_l)
    if site_random_effects:
        true_params.update(site_re_occ=site_re_occ, site_re_det=site_re_det, site_re_sd=site_re_sd)
    if obs_random_effects:
        true_params.update(obs_re=obs_re, obs_re_sd=obs_re_sd)
    data = dict(site_covs=site_covs, obs_covs=obs_covs, obs=obs, coords=coords, ell=ell)
    return data, true_params
