"""``HipMCMC`` -- the object ``fit`` returns as ``FitResult.mcmc``.

Stands where ``numpyro.infer.MCMC`` stood (biolith/utils/fit.py:105-135) and exposes what the
reference's consumers read from it: ``get_samples(group_by_chain)``, ``get_extra_fields()``,
``num_samples``, ``num_chains``, ``print_summary()`` and the private ``_states`` /
``_sample_field`` / ``_last_state`` triple that biolith/evaluation/diagnostics.py:10-13 touches.
"""
from __future__ import annotations

from collections.abc import MutableMapping
from types import SimpleNamespace

import numpy as np


class LazySamples(MutableMapping):
    """Mapping of posterior samples whose large deterministic sites are materialised on first access.

    ``prob_detection`` is (draws, J, T, N, S): 1 GB at 10k sites x 5 visits x 5000 draws
    (SURVEY.md section 3.1 iv).  The reference materialises it eagerly after sampling; here it is
    computed by the device on first ``samples["prob_detection"]`` and then cached.

    A ``collections.abc.MutableMapping``, NOT a ``dict`` subclass: CPython's fast paths for real dicts
    (``dict(s)``, ``{**s}``, ``f(**s)``) read the underlying table directly and would hand out the
    placeholders; for a Mapping they go through ``keys()`` / ``__getitem__`` and get the arrays.
    """

    def __init__(self, *a, **kw):
        self._data = dict(*a, **kw)
        self._lazy = {}

    def set_lazy(self, key, thunk):
        self._lazy[key] = thunk
        self._data[key] = None     # keeps the site's position in iteration order

    def __getitem__(self, key):
        if key in self._lazy:
            self._data[key] = self._lazy.pop(key)()
        return self._data[key]

    def __setitem__(self, key, value):
        self._lazy.pop(key, None)
        self._data[key] = value

    def __delitem__(self, key):
        self._lazy.pop(key, None)
        del self._data[key]

    def __iter__(self):
        return iter(self._data)

    def __len__(self):
        return len(self._data)

    def __contains__(self, key):
        return key in self._data

    def __repr__(self):
        parts = [f"{k!r}: <lazy>" if k in self._lazy else f"{k!r}: array{getattr(v, 'shape', '')}" for k, v in self._data.items()]
        return "LazySamples({" + ", ".join(parts) + "})"

    def __copy__(self):
        new = LazySamples(self._data)
        new._lazy = dict(self._lazy)
        return new

    copy = __copy__


class HipMCMC:
    _sample_field = "z"  # name numpyro's HMCState uses for the latent dict

    def __init__(self, result, latent, deterministic, num_warmup, spec_shape, thinning: int = 1):
        """``latent`` / ``deterministic``: dict name -> array grouped by chain (C, S, ...)."""
        self.num_chains, self.num_samples = result.draws.shape[:2]
        self.num_warmup = num_warmup
        self.thinning = thinning
        self.result = result
        self._latent = latent
        self._deterministic = deterministic
        self._shape = spec_shape
        grouped = dict(latent)
        grouped.update(deterministic)
        self._states = {self._sample_field: grouped,
                        "diverging": result.diverging,
                        "num_steps": result.num_steps,
                        "accept_prob": result.accept_prob,
                        "potential_energy": result.potential_energy}
        last = {k: v[:, -1] for k, v in latent.items()}
        self._last_state = SimpleNamespace(
            z=last,
            adapt_state=SimpleNamespace(step_size=result.step_size, inverse_mass_matrix=result.inv_mass),
        )
        self.last_state = self._last_state
        self.post_warmup_state = None

    @staticmethod
    def _flat(a):
        return a.reshape((-1,) + a.shape[2:])

    def get_samples(self, group_by_chain: bool = False):
        """Latent + deterministic sites, chains concatenated on axis 0 unless ``group_by_chain``."""
        out = LazySamples()
        for k, v in self._states[self._sample_field].items():
            if callable(v):
                out.set_lazy(k, (lambda f=v, g=group_by_chain: f() if g else HipMCMC._flat(f())))
            else:
                out[k] = v if group_by_chain else self._flat(v)
        return out

    def get_extra_fields(self, group_by_chain: bool = False):
        d = self.result.diverging
        return {"diverging": d if group_by_chain else d.reshape(-1)}

    def print_summary(self, prob: float = 0.9, exclude_deterministic: bool = True):
        from ..evaluation.diagnostics import summary

        sites = dict(self._latent)
        if not exclude_deterministic:
            sites.update({k: (v() if callable(v) else v) for k, v in self._deterministic.items()})
        table = summary(sites, prob=prob)
        lo, hi = f"{50 * (1 - prob):.1f}%", f"{50 * (1 + prob):.1f}%"
        print(f"{'':>16} {'mean':>9} {'std':>9} {'median':>9} {lo:>9} {hi:>9} {'n_eff':>9} {'r_hat':>9}")
        for name, st in table.items():
            flat = {k: np.asarray(v).reshape(-1) for k, v in st.items()}
            for i in range(flat["mean"].size):
                label = name if flat["mean"].size == 1 else f"{name}[{i}]"
                print(f"{label:>16} " + " ".join(f"{flat[c][i]:9.2f}" for c in
                                                  ("mean", "std", "median", lo, hi, "n_eff", "r_hat")))
        print(f"\nNumber of divergences: {int(self.result.diverging.sum())}")
