"""Minimal prior objects standing in for ``numpyro.distributions`` in model signatures.

Only what ``occu``'s signature names (biolith/models/occu.py:28-39).  The HIP engine samples
coefficients under ``Normal(loc, scale)`` priors (``prior.expand([K+1]).to_event(1)``,
biolith/regression/linear.py:28); the other classes exist so that reference-style calls that pass
the defaults keep working and non-default ones can be rejected with a clear message.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Normal:
    loc: float = 0.0
    scale: float = 1.0


@dataclass(frozen=True)
class Laplace:
    loc: float = 0.0
    scale: float = 1.0


class LocScale(tuple):
    """``(loc, scale)`` of a coefficient prior; compares equal to the plain tuple and carries the family
    (``"normal"`` or ``"laplace"``: the two ``grid_search_priors`` tries, biolith/utils/grid_search.py:366-371)."""

    def __new__(cls, loc, scale, family="normal"):
        self = super().__new__(cls, (loc, scale))
        self.family = family
        return self

    def __reduce__(self):
        return (LocScale, (self[0], self[1], self.family))


@dataclass(frozen=True)
class HalfNormal:
    scale: float = 1.0


@dataclass(frozen=True)
class Beta:
    concentration1: float = 1.0
    concentration0: float = 1.0


@dataclass(frozen=True)
class Gamma:
    concentration: float = 1.0
    rate: float = 1.0


@dataclass(frozen=True)
class Exponential:
    rate: float = 1.0


def as_half_normal(prior, name: str = "prior") -> float:
    """scale of a HalfNormal prior; duck-types numpyro's ``dist.HalfNormal`` (has .scale, no .loc)."""
    if type(prior).__name__ != "HalfNormal" or not hasattr(prior, "scale"):
        raise NotImplementedError(f"{name}: the HIP engine supports HalfNormal(scale) priors here, got {prior!r}")
    import numpy as np

    scale = np.asarray(prior.scale, dtype=float)
    if scale.size != 1:
        raise NotImplementedError(f"{name}: scalar scale only")
    scale = float(scale.reshape(()))
    if not scale > 0:
        raise ValueError(f"{name}: scale must be positive")
    return scale


def as_gamma(prior, name: str = "prior") -> tuple:
    """(concentration, rate) of a Gamma prior; duck-types numpyro's ``dist.Gamma``."""
    if type(prior).__name__ != "Gamma" or not hasattr(prior, "concentration") or not hasattr(prior, "rate"):
        raise NotImplementedError(f"{name}: the HIP engine supports Gamma(concentration, rate) priors here, got {prior!r}")
    import numpy as np

    a, b = np.asarray(prior.concentration, dtype=float), np.asarray(prior.rate, dtype=float)
    if a.size != 1 or b.size != 1:
        raise NotImplementedError(f"{name}: scalar concentration / rate only")
    a, b = float(a.reshape(())), float(b.reshape(()))
    if not (a > 0 and b > 0):
        raise ValueError(f"{name}: concentration and rate must be positive")
    return a, b


def as_exponential(prior, name: str = "prior") -> float:
    """rate of an Exponential prior; duck-types numpyro's ``dist.Exponential`` (has .rate)."""
    if type(prior).__name__ != "Exponential" or not hasattr(prior, "rate"):
        raise NotImplementedError(f"{name}: the HIP engine supports Exponential(rate) priors here, got {prior!r}")
    import numpy as np

    rate = np.asarray(prior.rate, dtype=float)
    if rate.size != 1:
        raise NotImplementedError(f"{name}: scalar rate only")
    rate = float(rate.reshape(()))
    if not rate > 0:
        raise ValueError(f"{name}: rate must be positive")
    return rate


def as_normal(prior, name: str = "prior") -> tuple:
    """(loc, scale) of a Normal -- or Laplace -- coefficient prior as a :class:`LocScale`; duck-types numpyro's
    ``dist.Normal`` / ``dist.Laplace`` (class name, .loc, .scale)."""
    cls = type(prior).__name__
    if cls not in ("Normal", "Laplace") or not hasattr(prior, "loc") or not hasattr(prior, "scale"):
        raise NotImplementedError(f"{name}: the HIP engine supports Normal(loc, scale) and Laplace(loc, scale) priors, got {prior!r}")
    import numpy as np

    loc, scale = np.asarray(prior.loc, dtype=float), np.asarray(prior.scale, dtype=float)
    if loc.size != 1 or scale.size != 1:
        raise NotImplementedError(f"{name}: scalar loc/scale only (the reference expands one prior over all coefficients)")
    loc, scale = float(loc.reshape(())), float(scale.reshape(()))
    if not scale > 0:
        raise ValueError(f"{name}: scale must be positive")
    return LocScale(loc, scale, cls.lower())


def as_beta(prior, name: str = "prior") -> tuple:
    """(a, b) of a Beta prior; duck-types numpyro's ``dist.Beta`` (concentration1 / concentration0)."""
    if type(prior).__name__ != "Beta" or not hasattr(prior, "concentration1") or not hasattr(prior, "concentration0"):
        raise NotImplementedError(f"{name}: the HIP engine supports Beta(a, b) priors here, got {prior!r}")
    import numpy as np

    a, b = np.asarray(prior.concentration1, dtype=float), np.asarray(prior.concentration0, dtype=float)
    if a.size != 1 or b.size != 1:
        raise NotImplementedError(f"{name}: scalar concentrations only")
    a, b = float(a.reshape(())), float(b.reshape(()))
    if not (a > 0 and b > 0):
        raise ValueError(f"{name}: concentrations must be positive")
    return a, b
