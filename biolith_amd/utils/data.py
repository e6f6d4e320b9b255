"""Input normalisation and sample renaming -- counterpart of biolith/utils/data.py.

``prepare_data`` accepts the same inputs as the reference (ndarrays or pandas DataFrames, with
MultiIndex columns for per-visit tables), returns the same shapes and the same covariate names, and
hands float32 arrays on (the reference converts with ``jnp.array`` without x64: data.py:135-140).
"""
from __future__ import annotations

import copy
from typing import List, Optional

import numpy as np

try:  # pandas is optional at import time; DataFrame inputs need it
    import pandas as pd
except Exception:  # pragma: no cover
    pd = None


def _is_df(x) -> bool:
    return pd is not None and isinstance(x, pd.DataFrame)


def _levels(df, what: str, allowed):
    if not isinstance(df.columns, pd.MultiIndex):
        return None
    n = len(df.columns.levels)
    if n not in allowed:
        raise ValueError(f"{what} with MultiIndex columns must have {' or '.join(map(str, allowed))} levels.")
    return [len(lv) for lv in df.columns.levels]


def prepare_data(site_covs=None, obs_covs=None, obs=None, session_duration=None):
    """-> (site_covs, obs_covs, obs, session_duration, site_covs_names, obs_covs_names)  (data.py:9-142)."""
    site_names = obs_names = None

    # the first DataFrame met (obs, site_covs, obs_covs, session_duration) fixes the row order (data.py:14-36)
    ref_index = next((t.index for t in (obs, site_covs, obs_covs, session_duration) if _is_df(t)), None)

    def align(t):
        if not _is_df(t) or ref_index is None:
            return t
        a, b = pd.Index(ref_index), pd.Index(t.index)
        same_members = len(a) == len(b) and a.isin(b).all() and b.isin(a).all()
        return t.loc[ref_index] if same_members and not b.equals(a) else t

    site_covs, obs_covs, session_duration, obs = map(align, (site_covs, obs_covs, session_duration, obs))

    if _is_df(site_covs):
        site_names = ["intercept"] + list(site_covs.columns)
        site_covs = site_covs.to_numpy()
    if _is_df(obs_covs):
        if not isinstance(obs_covs.columns, pd.MultiIndex):
            raise ValueError(
                "obs_covs DataFrame must use MultiIndex columns with levels "
                "(covariate, period, replicate) for multi-season data."
            )
        sizes = _levels(obs_covs, "obs_covs", (2, 3))
        obs_names = ["intercept"] + list(obs_covs.columns.levels[0])
        values = obs_covs.to_numpy()
        if len(sizes) == 2:   # (covariate, replicate) -> (site, 1, replicate, covariate)
            obs_covs = values.reshape(values.shape[0], *sizes).transpose(0, 2, 1)[:, None, :, :]
        else:                 # (covariate, period, replicate) -> (site, period, replicate, covariate)
            obs_covs = values.reshape(values.shape[0], *sizes).transpose(0, 2, 3, 1)

    def per_visit_table(t, what):
        if not _is_df(t):
            return t
        sizes = _levels(t, what, (2,))
        return t.to_numpy() if sizes is None else t.to_numpy().reshape(t.shape[0], *sizes)

    session_duration = per_visit_table(session_duration, "session_duration")
    obs = per_visit_table(obs, "obs")

    # insert the period axis on season-less ndarray inputs (data.py:113-128)
    def with_period_axis(a, name):
        if a is None or not isinstance(a, np.ndarray):
            return a
        if name == "obs_covs":
            if a.ndim == 2:
                a = a[:, :, None]
            if a.ndim == 3:
                a = a[:, None, :, :]
        elif a.ndim == 2:
            a = a[:, None, :]
        return a

    obs_covs = with_period_axis(obs_covs, "obs_covs")
    obs = with_period_axis(obs, "obs")
    session_duration = with_period_axis(session_duration, "session_duration")

    if site_names is None and site_covs is not None:
        site_names = [str(i) for i in range(np.shape(site_covs)[1] + 1)]
    if obs_names is None and obs_covs is not None:
        obs_names = [str(i) for i in range(np.shape(obs_covs)[-1] + 1)]

    def f32(a):
        return None if a is None else np.asarray(a, dtype=np.float32)

    return f32(site_covs), f32(obs_covs), f32(obs), f32(session_duration), site_names, obs_names


def rename_samples(samples, site_covs_names=None, obs_covs_names: Optional[List[str]] = None):
    """beta[..., i] -> ``cov_state_<name_i>``, alpha[..., i] -> ``cov_det_<name_i>`` (data.py:145-165)."""
    samples = copy.copy(samples)
    for prefix, base, names in (("cov_state_", "beta", site_covs_names), ("cov_det_", "alpha", obs_covs_names)):
        if names is None:
            continue
        for i, name in enumerate(names):
            if f"{base}_{i}" in samples:
                samples[f"{prefix}{name}"] = samples.pop(f"{base}_{i}")
        if base in samples:
            block = samples.pop(base)
            for i, name in enumerate(names):
                samples[f"{prefix}{name}"] = block[..., i]
    return samples
