"""Pointwise log-likelihood, lppd and WAIC -- the consumers of ``predict()``'s output
(biolith/evaluation/log_likelihood.py:10-96, lppd.py:10-95, waic.py:9-124).

The reference's ``log_likelihood`` substitutes the predictive sample into the model and asks
NumPyro for the log-probability of the observed ``y`` site, i.e. of
``Bernoulli(prob_detection_fp).log_prob(obs)`` with NumPyro's probability clamp
(``clip(p, finfo.tiny, 1 - finfo.eps)`` in float32) under ``mask_missing_obs``.  Those sites are plain
arrays in what ``biolith_amd.utils.predict`` returns, so the same numbers come out of NumPy here;
the ``*_manual`` variants are the marginal (``psi * p``) forms the reference cross-checks against.
Host code: array reductions over (draws, J, T, N, S), nothing here is on the sampling path.
"""
from __future__ import annotations

from typing import Callable, Dict

import numpy as np
from scipy.special import logsumexp

_TINY = np.float32(np.finfo(np.float32).tiny)
_ONE_MINUS_EPS = np.float32(1.0) - np.float32(np.finfo(np.float32).eps)


def _valid_obs(site_covs, obs_covs, obs):
    """(S, N, T, J) mask of observations that enter the likelihood (lppd.py:50-54)."""
    site_covs, obs_covs, obs = (np.asarray(a, dtype=np.float64) for a in (site_covs, obs_covs, obs))
    return (np.isfinite(obs) & np.isfinite(obs_covs).all(axis=-1)[None, ...]
            & np.isfinite(site_covs).all(axis=-1)[None, :, None, None])


def _detection_probability(model_fn, ps):
    """P(y = 1 | latent state) per draw, (n, J, T, N, S), from the predictive sites."""
    name = getattr(model_fn, "__biolith_amd_model__", None)
    if name is None:
        raise TypeError("model_fn must be a biolith_amd model (biolith_amd.models.occu / occu_rn)")
    p = np.asarray(ps["prob_detection"], dtype=np.float32)
    if name == "occu_rn":      # occu_rn.py:209-218
        n_i = np.asarray(ps["N_i"], dtype=np.float32)[:, None]
        return np.float32(1.0) - (np.float32(1.0) - p) ** n_i
    if "prob_detection_fp" in ps:  # occu.py:229-241
        return np.asarray(ps["prob_detection_fp"], dtype=np.float32)
    return p * np.asarray(ps["z"], dtype=np.float32)[:, None]


def log_likelihood(model_fn: Callable, posterior_samples: Dict[str, np.ndarray], observation_keys: set = {"y", "s"},
                   **kwargs) -> Dict[str, np.ndarray]:
    """Log-likelihood of every observation under every predictive draw: ``{"y": (n, J, T, N, S)}``.

    ``posterior_samples`` is what :func:`biolith_amd.utils.predict` returned; ``kwargs`` carry the data
    (``site_covs``, ``obs_covs``, ``obs``) as in the reference (log_likelihood.py:10-53).  Entries whose
    observation or covariates are missing are 0, as under ``mask_missing_obs`` (modeling.py:8-19).

    Examples
    --------
    >>> from biolith_amd.models import simulate, occu
    >>> from biolith_amd.utils import fit, predict
    >>> from biolith_amd.evaluation import log_likelihood
    >>> data, _ = simulate()
    >>> results = fit(occu, **data)
    >>> preds = predict(occu, results.mcmc, **data)
    >>> log_likelihood(occu, preds, **data)
    """
    ps = {k: v for k, v in posterior_samples.items() if k not in observation_keys}
    obs = np.asarray(kwargs["obs"], dtype=np.float32)
    valid = _valid_obs(kwargs["site_covs"], kwargs["obs_covs"], obs).transpose((3, 2, 1, 0))  # (J, T, N, S)
    if getattr(model_fn, "__biolith_amd_model__", None) == "occu_cs":
        # the observed site is the score: Normal((1 - f) mu0 + f mu1, (1 - f) sigma0 + f sigma1).log_prob(obs)  (occu_cs.py:224-232)
        f = np.asarray(ps["f"], dtype=np.float32)
        shape = (-1,) + (1,) * 4
        mu = np.where(f > 0, np.asarray(ps["mu1"], np.float32).reshape(shape), np.asarray(ps["mu0"], np.float32).reshape(shape))
        sg = np.where(f > 0, np.asarray(ps["sigma1"], np.float32).reshape(shape), np.asarray(ps["sigma0"], np.float32).reshape(shape))
        sc = np.where(valid, obs.transpose((3, 2, 1, 0)), np.float32(0.0))[None]
        ll = -0.5 * ((sc - mu) / sg) ** 2 - np.log(sg) - np.float32(0.5 * np.log(2 * np.pi))
        return {"s": np.where(valid[None], ll, np.float32(0.0)).astype(np.float32)}
    y = np.where(valid, obs.transpose((3, 2, 1, 0)), np.float32(0.0)).astype(np.float32)
    name = getattr(model_fn, "__biolith_amd_model__", None)
    if name == "nmixture":
        # y ~ Binomial(N_i, prob_detection)  (nmixture.py:206-220): numpyro's BinomialProbs.log_prob, -inf above N_i
        from scipy.special import gammaln, xlog1py, xlogy

        n_i = np.asarray(ps["N_i"], dtype=np.float64)[:, None]                      # (n, 1, T, N, S)
        p = np.clip(np.asarray(ps["prob_detection"], dtype=np.float64), _TINY, _ONE_MINUS_EPS)
        yy = y[None].astype(np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            ll = (gammaln(n_i + 1.0) - gammaln(yy + 1.0) - gammaln(np.maximum(n_i - yy, 0.0) + 1.0)
                  + xlogy(yy, p) + xlog1py(n_i - yy, -p))
        ll = np.where(yy > n_i, -np.inf, ll)
        return {"y": np.where(valid[None], ll, 0.0).astype(np.float32)}
    if name == "occu_cop":
        # y ~ Poisson(session_duration (z rate_detection + (1 - z) f_u + f_c))  (occu_cop.py:222-255)
        from scipy.special import gammaln, xlogy

        dur = np.asarray(kwargs["session_duration"], dtype=np.float64)
        if dur.ndim == 3:
            dur = dur[None]
        dur = np.broadcast_to(dur, obs.shape).transpose((3, 2, 1, 0))[None]        # (1, J, T, N, S)
        z = np.asarray(ps["z"], dtype=np.float64)[:, None]
        lam = np.asarray(ps["rate_detection"], dtype=np.float64)
        shape = (-1,) + (1,) * 4
        # Predictive leaves the posterior's own sites out of predict()'s result (as numpyro's does), so a model with a
        # false-positive rate needs that site merged in by the caller: {**predictions, "rate_fp_constant": fit.samples[...]}
        for opt, site in (("false_positives_constant", "rate_fp_constant"), ("false_positives_unoccupied", "rate_fp_unoccupied")):
            if kwargs.get(opt) and site not in ps:
                raise ValueError(f"log_likelihood(occu_cop, {opt}=True): posterior_samples lacks the site {site!r} "
                                 "(merge it in from fit().samples; predict() returns predictive sites only)")
        f_c = np.asarray(ps["rate_fp_constant"], np.float64).reshape(shape) if "rate_fp_constant" in ps else 0.0
        f_u = np.asarray(ps["rate_fp_unoccupied"], np.float64).reshape(shape) if "rate_fp_unoccupied" in ps else 0.0
        mu = dur * (z * lam + (1.0 - z) * f_u + f_c)
        yy = y[None].astype(np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            ll = xlogy(yy, mu) - mu - gammaln(yy + 1.0)        # rate 0: 0 for a zero count, -inf otherwise
        return {"y": np.where(valid[None], ll, 0.0).astype(np.float32)}
    prob = np.clip(_detection_probability(model_fn, ps), _TINY, _ONE_MINUS_EPS)
    with np.errstate(divide="ignore"):
        ll = y[None] * np.log(prob) + (np.float32(1.0) - y[None]) * np.log1p(-prob)
    return {"y": np.where(valid[None], ll, np.float32(0.0))}


def log_likelihood_manual(posterior_samples: Dict[str, np.ndarray], data: Dict[str, np.ndarray], eps=1e-10) -> np.ndarray:
    """Marginal per-observation log-likelihood ``y log(psi p) + (1 - y) log(1 - psi p)`` of the
    no-false-positive Bernoulli model, (n, S, N, T, J) (log_likelihood.py:56-96).  NaN where ``obs`` is NaN."""
    y = np.asarray(data["obs"], dtype=np.float64).transpose((3, 2, 1, 0))[None]     # (1, J, T, N, S)
    p = np.asarray(posterior_samples["prob_detection"], dtype=np.float64)
    psi = np.asarray(posterior_samples["psi"], dtype=np.float64)
    if psi.ndim == 2:
        psi = psi[:, None, :, None]
    elif psi.ndim == 3:
        psi = psi[:, None, ...]
    if psi.shape[1] != y.shape[2]:
        psi = np.broadcast_to(psi, (psi.shape[0], y.shape[2], psi.shape[2], psi.shape[3]))
    joint = p * psi[:, None]
    ll = np.log(np.clip(joint, eps, 1 - eps)) * y + np.log(np.clip(1 - joint, eps, 1 - eps)) * (1 - y)
    return ll.transpose((0, 4, 3, 2, 1))


def _pointwise(ll_valid):
    """(lppd, p_waic) of a (draws, n_valid) log-likelihood matrix (lppd.py:59-62, waic.py:70-78)."""
    lppd = float(np.sum(logsumexp(ll_valid, axis=0) - np.log(ll_valid.shape[0])))
    p_waic = float(np.sum(np.var(ll_valid, axis=0, ddof=1)))
    return lppd, p_waic


def _model_ll(model_fn, posterior_samples, kwargs):
    valid = _valid_obs(kwargs["site_covs"], kwargs["obs_covs"], kwargs["obs"])
    ll = next(iter(log_likelihood(model_fn, posterior_samples, **kwargs).values())).transpose((0, 4, 3, 2, 1))
    return ll[:, valid].astype(np.float64)


def _manual_ll(posterior_samples, data):
    valid = _valid_obs(data["site_covs"], data["obs_covs"], data["obs"])
    return log_likelihood_manual(posterior_samples, data)[:, valid]


def lppd(model_fn: Callable, posterior_samples: Dict[str, np.ndarray], **kwargs) -> float:
    """Log pointwise predictive density ``sum_i log mean_q p(y_i | theta_q)`` over valid observations (lppd.py:10-63)."""
    return _pointwise(_model_ll(model_fn, posterior_samples, kwargs))[0]


def lppd_manual(posterior_samples: Dict[str, np.ndarray], data: Dict[str, np.ndarray]) -> float:
    """lppd from the marginal ``psi * p`` likelihood (lppd.py:66-95)."""
    return _pointwise(_manual_ll(posterior_samples, data))[0]


def waic(model_fn: Callable, posterior_samples: Dict[str, np.ndarray], **kwargs) -> Dict[str, float]:
    """``{"waic": -2 (lppd - p_waic), "p_waic", "lppd"}`` (waic.py:9-82)."""
    l, p = _pointwise(_model_ll(model_fn, posterior_samples, kwargs))
    return {"waic": -2 * (l - p), "p_waic": p, "lppd": l}


def waic_manual(posterior_samples: Dict[str, np.ndarray], data: Dict[str, np.ndarray]) -> Dict[str, float]:
    """WAIC from the marginal ``psi * p`` likelihood (waic.py:85-124)."""
    l, p = _pointwise(_manual_ll(posterior_samples, data))
    return {"waic": -2 * (l - p), "p_waic": p, "lppd": l}
