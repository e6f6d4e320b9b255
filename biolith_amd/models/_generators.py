"""Shared scaffold of the four data generators (simulate, simulate_rn, simulate_cop, simulate_nmixture).

The reference's generators (biolith/models/occu.py:245-430, occu_rn.py:225-358, occu_cop.py:258-396,
nmixture.py:223-369) all run the same rejection loop over the same NumPy PCG64 stream:

    [extra leading draws] -> beta -> alpha -> site covariates -> [site random effects] -> latent state
    -> observation covariates -> [observation random effects] -> observations -> [three missingness masks]

and differ only in the latent / observation distributions and in the acceptance test.  ``Generator``
fixes that order once; a model supplies ``latent``, ``observe`` and ``accept`` callbacks.  Because the
order and the array expressions are the reference's, equal arguments give bit-identical arrays
(``tests/golden`` holds fixtures made by importing the reference's own functions).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np


def expit(x):
    # evaluated as the reference does, 1 / (1 + exp(-x)), so that downstream draws agree bitwise
    return 1 / (1 + np.exp(-x))


@dataclass
class Draw:
    """Everything one pass of the rejection loop produced."""
    beta: np.ndarray            # (S, Ks+1)
    alpha: np.ndarray           # (S, Ko+1)
    site_covs: np.ndarray       # (N, Ks)
    obs_covs: np.ndarray        # (N, T, J, Ko)
    site_linear: np.ndarray     # (S, N)        beta0 + x beta (+ site random effect)
    obs_linear: np.ndarray      # (S, N, T, J)  alpha0 + w alpha (+ random effects)
    latent: np.ndarray          # (S, T, N)
    obs: np.ndarray             # (S, N, T, J) float, NaN = knocked out
    extra: dict = field(default_factory=dict)


@dataclass
class Generator:
    n_species: int
    n_sites: int
    n_periods: int
    n_replicates: int
    n_site_covs: int
    n_obs_covs: int
    latent: Callable            # (rng, site_linear) -> (S, T, N)
    observe: Callable           # (rng, obs_linear, latent_by_site (S, N, T), extra) -> (S, N, T, J)
    accept: Callable            # (Draw) -> bool
    leading: Optional[Callable] = None                  # (rng) -> dict, drawn before the coefficients
    site_re_sd: Optional[float] = None                  # site random effects (two draws) when not None
    obs_re_sd: Optional[float] = None                   # observation random effects when not None
    simulate_missing: bool = False

    def run(self, random_seed: int) -> Draw:
        rng = np.random.default_rng(random_seed)
        S, N, T, J = self.n_species, self.n_sites, self.n_periods, self.n_replicates
        while True:
            extra = self.leading(rng) if self.leading is not None else {}
            beta = rng.normal(size=(S, self.n_site_covs + 1))
            alpha = rng.normal(size=(S, self.n_obs_covs + 1))
            site_covs = rng.normal(size=(N, self.n_site_covs))
            w = np.zeros(N)  # spatial effect: not built (the reference draws nothing for it when spatial=False)
            if self.site_re_sd is not None:
                site_re_a = rng.normal(0, self.site_re_sd, size=(S, N))
                site_re_b = rng.normal(0, self.site_re_sd, size=(S, N))
            else:
                site_re_a, site_re_b = np.zeros((S, N)), np.zeros((S, N))
            site_linear = beta[:, 0][:, None] + np.tensordot(beta[:, 1:], site_covs, axes=([1], [1])) + w[None, :] + site_re_a
            latent = self.latent(rng, site_linear)
            obs_covs = rng.normal(size=(N, T, J, self.n_obs_covs))
            if self.obs_re_sd is not None:
                obs_re = rng.normal(0, self.obs_re_sd, size=(S, N, T, J))
            else:
                obs_re = np.zeros((S, N, T, J))
            obs_linear = (alpha[:, 0][:, None, None, None] + np.tensordot(alpha[:, 1:], obs_covs, axes=([1], [3]))
                          + site_re_b[:, :, None, None] + obs_re)
            obs = self.observe(rng, obs_linear, latent.transpose(0, 2, 1), extra)
            if self.simulate_missing:  # 20 % of the observations, 5 % of either covariate array
                obs[rng.choice([True, False], size=obs.shape, p=[0.2, 0.8])] = np.nan
                obs_covs[rng.choice([True, False], size=obs_covs.shape, p=[0.05, 0.95])] = np.nan
                site_covs[rng.choice([True, False], size=site_covs.shape, p=[0.05, 0.95])] = np.nan
            extra.update(w=w, site_re_a=site_re_a, site_re_b=site_re_b, obs_re=obs_re)
            draw = Draw(beta, alpha, site_covs, obs_covs, site_linear, obs_linear, latent, obs, extra)
            if self.accept(draw):
                return draw


def within(value, lo, hi) -> bool:
    """``not (value < lo or value > hi)``: the reference's loop conditions, NaN behaviour included."""
    return not (value < lo or value > hi)


def observed_mean(obs) -> float:
    return np.mean(obs[np.isfinite(obs)])
