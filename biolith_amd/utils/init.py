"""Initialisation strategies for ``fit(init_strategy=...)`` (biolith/utils/fit.py:29, 93: ``NUTS(model_fn, init_strategy=init_strategy or
init_to_uniform)``).  The reference hands NumPyro's ``numpyro.infer.init_to_*`` callables through; NumPyro is not part of this engine,
so these are small descriptors of the same names and meaning that ``fit`` resolves into the kernel's ``init_theta`` (one start position
per chain, in the unconstrained space the sampler works in):

* ``init_to_uniform(radius=2)`` -- Uniform(-radius, radius) per unconstrained coordinate (the default; with radius 2 the kernel draws
  it itself from the chain's own xoshiro streams, exactly as ``init_strategy=None``);
* ``init_to_feasible()`` -- every unconstrained coordinate 0;
* ``init_to_value(values={"beta": ..., "alpha": ...})`` -- the named sites at the given values, everything else as ``init_to_uniform``
  (NumPyro's rule);
* ``init_to_mean()`` / ``init_to_median(num_samples=15)`` / ``init_to_sample()`` -- from the coefficients' priors (Normal / Laplace:
  mean = median = loc; a draw; the median of ``num_samples`` draws): plain coefficient models only.

Objects with these names from NumPyro itself (``functools.partial`` of its ``init_to_*`` functions) are recognised by name and keyword
arguments, so a reference-style call keeps working.  Draw-level equality with NumPyro is impossible either way (its keys are threefry).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import numpy as np


@dataclass(frozen=True)
class InitStrategy:
    kind: str
    radius: float = 2.0
    num_samples: int = 15
    values: Dict[str, Any] = field(default_factory=dict)


def init_to_uniform(site=None, radius: float = 2.0) -> InitStrategy:
    return InitStrategy("uniform", radius=float(radius))


def init_to_feasible(site=None) -> InitStrategy:
    return InitStrategy("feasible")


def init_to_value(site=None, values: Optional[Dict[str, Any]] = None) -> InitStrategy:
    return InitStrategy("value", values=dict(values or {}))


def init_to_mean(site=None) -> InitStrategy:
    return InitStrategy("mean")


def init_to_median(site=None, num_samples: int = 15) -> InitStrategy:
    return InitStrategy("median", num_samples=int(num_samples))


def init_to_sample(site=None) -> InitStrategy:
    return InitStrategy("sample")


_BY_NAME = dict(init_to_uniform=init_to_uniform, init_to_feasible=init_to_feasible, init_to_value=init_to_value,
                init_to_mean=init_to_mean, init_to_median=init_to_median, init_to_sample=init_to_sample)


def as_strategy(obj) -> Optional[InitStrategy]:
    """None, one of this module's descriptors (or the function itself, as ``fit(init_strategy=init_to_median)``), or NumPyro's
    ``init_to_*`` callables / their ``functools.partial`` forms, recognised by name."""
    if obj is None or isinstance(obj, InitStrategy):
        return obj
    func, kw = getattr(obj, "func", obj), dict(getattr(obj, "keywords", None) or {})
    name = getattr(func, "__name__", "")
    if name in _BY_NAME:
        kw.pop("site", None)
        return _BY_NAME[name](**kw)
    raise NotImplementedError(f"init_strategy={obj!r}: the HIP engine knows init_to_uniform / _feasible / _value / _mean / _median / _sample "
                              "(biolith_amd.utils.init, or NumPyro's callables of those names)")


def _prior_draws(rng, prior, shape):
    loc, scale = float(prior[0]), float(prior[1])
    if getattr(prior, "family", "normal") == "laplace":
        return rng.laplace(loc, scale, size=shape)
    return rng.normal(loc, scale, size=shape)


def initial_positions(strategy: Optional[InitStrategy], *, D: int, Ks: int, Ko: int, n_species: int, plain: bool, blocks_first: bool = True, prior_beta, prior_alpha,
                      num_chains: int, first_chain: int, seed: int, species: int = 0):
    """Start positions ``(num_chains, D)`` float64 for one launch, or None when the kernel's own draw applies (``init_to_uniform`` at
    radius 2).  theta = [species 0: beta (Ks + 1), alpha (Ko + 1) | species 1: ... | further unconstrained coordinates]; ``n_species``
    species blocks lie in THIS launch's theta (one unless the species are sampled jointly), the launch's first species is ``species``;
    ``plain``: theta holds nothing but those coefficient blocks; ``blocks_first``: it starts with them (all models but the dynamic one).  Chain c of the fit gets its own generator (seed, c): what a chain starts
    from does not depend on how the chains are dealt to launches."""
    if strategy is None or (strategy.kind == "uniform" and strategy.radius == 2.0):
        return None
    Dsp = Ks + Ko + 2
    out = np.empty((num_chains, D), dtype=np.float64)
    for c in range(num_chains):
        rng = np.random.default_rng([int(seed) & 0x7FFFFFFF, first_chain + c, 0x1B1D])
        if strategy.kind == "feasible":
            out[c] = 0.0
            continue
        radius = strategy.radius if strategy.kind == "uniform" else 2.0
        out[c] = rng.uniform(-radius, radius, size=D)
        if strategy.kind == "uniform":
            continue
        if strategy.kind == "value":
            vals = {k: np.asarray(v, dtype=np.float64) for k, v in strategy.values.items()}
            unknown = set(vals) - {"beta", "alpha"}
            if unknown:
                raise NotImplementedError(f"init_to_value: sites {sorted(unknown)} are not coefficient sites of this engine (beta, alpha)")
            if not blocks_first and vals:
                raise NotImplementedError("init_to_value: not built for the dynamic model's coefficient layout")
            for s in range(n_species):
                for name, off, width in (("beta", 0, Ks + 1), ("alpha", Ks + 1, Ko + 1)):
                    if name in vals:
                        v = vals[name]
                        row = v if v.ndim == 1 else v[species + s]   # (n_species, K + 1), the reference's site shape, or one row for all
                        if row.shape != (width,):
                            raise ValueError(f"init_to_value: {name} must have {width} coefficients per species, got shape {v.shape}")
                        out[c, s * Dsp + off: s * Dsp + off + width] = row
            continue
        if not plain:
            raise NotImplementedError(f"init_to_{strategy.kind}: built for models whose coordinates are all regression coefficients "
                                      "(no false-positive rate, random effects or score parameters); use init_to_uniform / _feasible / _value")
        n = 1 if strategy.kind == "sample" else strategy.num_samples
        for s in range(n_species):
            for prior, off, width in ((prior_beta, 0, Ks + 1), (prior_alpha, Ks + 1, Ko + 1)):
                if strategy.kind == "mean":
                    col = np.full(width, float(prior[0]))
                else:
                    col = np.median(_prior_draws(rng, prior, (n, width)), axis=0)
                out[c, s * Dsp + off: s * Dsp + off + width] = col
    return out
