"""Convergence diagnostics -- counterpart of biolith/evaluation/diagnostics.py:8-66.

The reference delegates to ``numpyro.diagnostics.summary`` (diagnostics.py:23).  NumPyro is not a
dependency here, so its estimators (Geyer initial-monotone-sequence ESS on
FFT autocovariances, split R-hat) are implemented below; ``tests/`` checks them against the
independent restatement in ``oracle/``.
"""
from __future__ import annotations

from operator import attrgetter

import numpy as np

try:
    from scipy import fft as _fft
except Exception:  # pragma: no cover
    _fft = None

_CHUNK_ELEMS = 1 << 22  # FFT work buffer bound (elements) when a site has many entries


def _fast_len(n: int) -> int:
    if _fft is not None:
        return int(_fft.next_fast_len(n))
    m = 1
    while m < n:
        m *= 2
    return m


def _between_within(x):
    """x (C, n, m) -> (W, var_plus) per numpyro _compute_chain_variance_stats."""
    n = x.shape[1]
    within = x.var(axis=1, ddof=1).mean(axis=0)
    plus = within * (n - 1) / n
    if x.shape[0] > 1:
        plus = plus + x.mean(axis=1).var(axis=0, ddof=1)
    else:
        within = plus
    return within, plus


def _ess_block(x):
    """x (C, n, m) float64 -> n_eff (m,)."""
    c, n, m = x.shape
    nfft = 2 * _fast_len(n)
    xc = x - x.mean(axis=1, keepdims=True)
    rfft, irfft = (_fft.rfft, _fft.irfft) if _fft is not None else (np.fft.rfft, np.fft.irfft)
    spec = rfft(xc, n=nfft, axis=1)
    # biased autocovariance at lags 0..n-1, averaged over chains
    gamma = irfft(spec.real ** 2 + spec.imag ** 2, n=nfft, axis=1)[:, :n].mean(axis=0) / n
    within, plus = _between_within(x)
    with np.errstate(invalid="ignore", divide="ignore"):
        rho = 1.0 - (within - gamma) / plus
    rho[0] = 1.0
    pairs = rho[: (n // 2) * 2].reshape(n // 2, 2, m).sum(axis=1)
    tail = np.minimum.accumulate(np.clip(pairs[1:], 0.0, None), axis=0)
    tau = -1.0 + 2.0 * (pairs[0] + tail.sum(axis=0))
    return c * n / tau


def effective_sample_size(x):
    """ESS of draws ``x`` shaped (chains, draws, ...) -> array shaped ``x.shape[2:]``."""
    x = np.asarray(x)
    if x.ndim < 2 or x.shape[1] < 2:
        raise ValueError("need (chains, draws>=2, ...)")
    c, n = x.shape[:2]
    tail = x.shape[2:]
    flat = x.reshape(c, n, -1)
    m = flat.shape[2]
    out = np.empty(m)
    step = max(1, _CHUNK_ELEMS // (c * 2 * _fast_len(n)))
    for s in range(0, m, step):
        out[s:s + step] = _ess_block(flat[:, :, s:s + step].astype(np.float64))
    return out.reshape(tail)


def split_gelman_rubin(x):
    """Split R-hat of draws shaped (chains, draws, ...)."""
    x = np.asarray(x, dtype=np.float64)
    c, n = x.shape[:2]
    h = n // 2
    halves = np.concatenate([x[:, :h], x[:, n - h:]], axis=0).reshape(2 * c, h, -1)
    within, plus = _between_within(halves)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = np.sqrt(plus / within)
    return r.reshape(x.shape[2:])


def hpdi(x, prob: float = 0.9, axis: int = 0):
    """Narrowest interval holding ``prob`` of the mass along ``axis``."""
    x = np.sort(np.moveaxis(np.asarray(x), axis, 0), axis=0)
    n = x.shape[0]
    width = int(np.floor(prob * n))
    lows, highs = x[: n - width], x[width:]
    idx = np.argmin(highs - lows, axis=0)[None]
    return np.take_along_axis(lows, idx, 0)[0], np.take_along_axis(highs, idx, 0)[0]


def summary(sites, prob: float = 0.9, group_by_chain: bool = True):
    """name -> {mean, std, median, lo%, hi%, n_eff, r_hat} for arrays shaped (chains, draws, ...)."""
    lo, hi = f"{50 * (1 - prob):.1f}%", f"{50 * (1 + prob):.1f}%"
    out = {}
    for name, v in sites.items():
        v = np.asarray(v)
        if not group_by_chain:
            v = v[None]
        flat = v.reshape((-1,) + v.shape[2:]).astype(np.float64)
        l, h = hpdi(flat, prob)
        out[name] = {
            "mean": flat.mean(axis=0), "std": flat.std(axis=0, ddof=1), "median": np.median(flat, axis=0),
            lo: l, hi: h, "n_eff": effective_sample_size(v), "r_hat": split_gelman_rubin(v),
        }
    return out


def diagnostics(mcmc, exclude_deterministic: bool = True):
    """mean R-hat, mean ESS fraction, divergence fraction, mean sd of beta / alpha (diagnostics.py:8-66)."""
    sites = mcmc._states[mcmc._sample_field]
    if isinstance(sites, dict) and exclude_deterministic:
        latent = attrgetter(mcmc._sample_field)(mcmc._last_state)
        if isinstance(latent, dict):
            sites = {k: v for k, v in sites.items() if k in latent}
    sites = {k: (v() if callable(v) else v) for k, v in sites.items()}
    table = summary(sites)
    total = mcmc.num_samples * mcmc.num_chains
    mean_r_hat = sum(float(np.mean(v["r_hat"])) for v in table.values()) / len(table)
    mean_frac_eff = sum(float(np.mean(v["n_eff"])) for v in table.values()) / len(table) / total
    extra = mcmc.get_extra_fields()
    if extra is not None and "diverging" in extra:
        frac_diverging = float(np.sum(extra["diverging"])) / total
    else:
        frac_diverging = float("nan")

    def mean_sd(name):
        return float(np.mean(table[name]["std"])) if name in table else float("nan")

    return dict(mean_r_hat=mean_r_hat, mean_frac_eff=mean_frac_eff, frac_diverging=frac_diverging,
                mean_beta_sd=mean_sd("beta"), mean_alpha_sd=mean_sd("alpha"))
