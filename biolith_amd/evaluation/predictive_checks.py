"""Deviance, residuals and the posterior predictive check -- the remaining consumers of ``predict()``'s output
(biolith/evaluation/deviance.py:10-117, residuals.py:6-90, posterior_predictive_check.py:17-160).

Host code: array reductions over the predictive sites (``psi``, ``z``, ``prob_detection``, ``y``), nothing here is on
the sampling path.  Layouts are the reference's: predictive sites are (draws, [J,] T, N, S), ``obs`` is (S, N, T, J).
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import numpy as np
from scipy.special import logsumexp

from .predictive_density import _valid_obs, log_likelihood, log_likelihood_manual


def _deviance_of(ll_valid) -> float:
    """-2 log of the posterior-mean likelihood of the whole data set; ``ll_valid`` is (draws, n_valid)."""
    per_draw = ll_valid.sum(axis=1)
    return float(-2.0 * (logsumexp(per_draw) - np.log(per_draw.shape[0])))


def deviance(model_fn: Callable, posterior_samples: Dict[str, np.ndarray], **kwargs) -> float:
    """Deviance as a scoring rule, ``-2 log( mean_q prod_ij p(y_ij | z_i^q, p_ij^q) )`` over the valid
    observations (deviance.py:10-71; spOccupancy, Hooten & Hobbs 2015)."""
    valid = _valid_obs(kwargs["site_covs"], kwargs["obs_covs"], kwargs["obs"])
    ll = log_likelihood(model_fn, posterior_samples, **kwargs)["y"].transpose((0, 4, 3, 2, 1))
    return _deviance_of(ll[:, valid].astype(np.float64))


def deviance_manual(posterior_samples: Dict[str, np.ndarray], data: Dict[str, np.ndarray]) -> float:
    """Deviance from the marginal ``psi * p`` likelihood of the no-false-positive Bernoulli model (deviance.py:74-117)."""
    valid = _valid_obs(data["site_covs"], data["obs_covs"], data["obs"])
    return _deviance_of(log_likelihood_manual(posterior_samples, data)[:, valid])


def residuals(posterior_samples: Dict[str, np.ndarray], obs) -> Tuple[np.ndarray, np.ndarray]:
    """Occupancy and detection residuals of Wright et al. (2019) per predictive draw (residuals.py:6-90):
    ``o = z - psi`` of shape (draws, T, N, S), and ``d = y - p`` where the draw has the site occupied, NaN elsewhere,
    of shape (draws, S, N, T, J)."""
    z = np.asarray(posterior_samples["z"], dtype=np.float64)
    psi = np.asarray(posterior_samples["psi"], dtype=np.float64)
    p = np.asarray(posterior_samples["prob_detection"], dtype=np.float64)      # (draws, J, T, N, S)
    occupancy = z - psi
    y = np.asarray(obs, dtype=np.float64).transpose((3, 2, 1, 0))[None]        # (1, J, T, N, S)
    detection = np.where(z[:, None] == 1, y - p, np.nan)
    return occupancy, detection.transpose((0, 4, 3, 2, 1))


def _freeman_tukey(observed, expected):
    return (np.sqrt(observed) - np.sqrt(expected)) ** 2


def _chi_squared(observed, expected, eps: float = 1e-10):
    return (observed - expected) ** 2 / (expected + eps)


def posterior_predictive_check(posterior_samples: Dict[str, np.ndarray], obs, group_by: str = "site",
                               statistic: str = "freeman-tukey") -> float:
    """Bayesian p-value ``P(T(y_rep, theta) > T(y, theta) | y)`` with detections grouped by site or by revisit
    and the Freeman-Tukey or chi-squared discrepancy against ``E = psi * p`` (posterior_predictive_check.py:17-160;
    valid without false positives, as the reference notes)."""
    stats = {"freeman-tukey": _freeman_tukey, "chi-squared": _chi_squared}
    if statistic not in stats:
        raise ValueError(f"`statistic` must be one of {list(stats)}")
    if group_by not in ("site", "revisit"):
        raise ValueError("`group_by` must be either 'site' or 'revisit'")
    stat = stats[statistic]
    obs = np.asarray(obs, dtype=np.float64)                                     # (S, N, T, J)
    y_rep = np.asarray(posterior_samples["y"], dtype=np.float64)
    p = np.asarray(posterior_samples["prob_detection"], dtype=np.float64)
    psi = np.asarray(posterior_samples["psi"], dtype=np.float64)
    if y_rep.ndim == 5:
        y_rep = y_rep.transpose((0, 4, 3, 2, 1))                                # -> (draws, S, N, T, J)
    if p.ndim == 5:
        p = p.transpose((0, 4, 3, 2, 1))
    if psi.ndim == 3:
        psi = psi[:, None, ...]
    elif psi.ndim == 2:
        psi = psi[:, None, :, None]
    expected = psi.transpose((0, 3, 2, 1))[..., None] * p                       # (draws, S, N, T, J)
    seen = np.isfinite(obs)[None]
    axes_obs, axes_rep = ((2, 3), (3, 4)) if group_by == "site" else ((1,), (2,))
    obs_g = np.nansum(obs, axis=axes_obs)
    rep_g = np.where(seen, y_rep, 0.0).sum(axis=axes_rep)
    exp_g = np.where(seen, expected, 0.0).sum(axis=axes_rep)
    reduce_axes = tuple(range(1, exp_g.ndim))
    d_obs = stat(obs_g[None], exp_g).sum(axis=reduce_axes)
    d_rep = stat(rep_g, exp_g).sum(axis=reduce_axes)
    return float(np.mean(d_rep > d_obs))
