#!/usr/bin/env python3
"""Generate tests/golden/reference_logjoint_<case>.json by EXECUTING the reference's model functions (build container only).

What runs is the reference's own model text, imported from /root/reference: ``biolith.models.occu`` (occu.py:136-242),
``occu_rn`` (occu_rn.py:123-222 + utils/distributions.py:6-40), ``occu_cop`` (occu_cop.py:150-255), ``nmixture``
(nmixture.py:150-220), with ``regression/linear.py:28-66`` and ``utils/modeling.py:8-39`` underneath -- their plates, transposes,
masks, ``prob_detection_fp`` algebra, ``N_i[None, ...]`` broadcasts and the ``N_i_trunc_norm`` factor are the reference's, not a
restatement.  What does NOT run is numpyro / jax / funsor (not installed, not installable: SURVEY.md section 8c).  In their place
this script serves a FUNCTIONAL NumPy shim for exactly the names the model files use:

* ``jax.numpy`` -> numpy (float64), ``jax.nn.sigmoid``, ``jax.scipy.special.logsumexp``;
* ``numpyro.sample / plate / deterministic / factor``, ``numpyro.handlers.mask``;
* ``numpyro.distributions.Normal / Laplace / HalfNormal / Beta / Exponential / Bernoulli / Poisson / Binomial / Categorical``
  with ``.expand(...).to_event(...)`` and ``log_prob``;
* everything else under those roots is a MagicMock (bart.py, mlp.py, spatial.py only need to IMPORT).

UPSTREAM-ASSUMED (restated here from numpyro's published behaviour, cannot be executed in this image):
  (1) ``clamp_probs``: Bernoulli probs clipped to [finfo.tiny, 1 - finfo.eps] of float32 (the reference runs in float32);
  (2) ``Categorical(logits)`` renormalises its logits;
  (3) a positive-support site (HalfNormal, Exponential) is sampled on the log scale, a unit-interval site (Beta) on the logit
      scale, each with its log-Jacobian added to the potential (``biject_to(support)``);
  (4) a plate expands a distribution's batch shape to the plate's size;
  (5) parallel enumeration: the enumerated site's value is ``arange(K)`` on a NEW axis left of all plate axes
      (``-(max_plate_nesting + 1)``), and the log-density is the sum-product contraction: factors are summed over the plates the
      enumerated site is not in, added, ``logsumexp``'d over the enumeration axis and summed over the remaining plates;
  (6) all of NUTS.  Nothing here touches the sampler.

The script cross-checks (5) against a second contraction that never builds an enumeration axis: the model body is run once per
VALUE of the enumerated site (z = 0, z = 1; N = 0 ... K) and the per-(site, period) terms are ``logsumexp``'d afterwards.

Only DATA is written: the simulator's kwargs, the model's kwargs, the unconstrained values of every latent site, the potential
U = -(log joint + log-Jacobians), and for small models a central-difference gradient.  Nothing of the reference's source travels.

Run:  python tests/golden/make_reference_logjoint.py      (needs /root/reference; never run on the GPU box)
"""
import contextlib
import hashlib
import importlib.abc
import importlib.machinery
import io
import json
import os
import sys
import types
from collections import OrderedDict
from unittest import mock

import numpy as np
from scipy import special as sps

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
CLAMP = np.finfo(np.float32)          # (1): the reference's arrays are float32 (utils/data.py:135-140)
MAX_PLATE_NESTING = 4                 # species -1, site -2, period -3, replicate -4 (occu.py:182-237)


# ------------------------------------------------------------------------------------------------------------------
# the handler state: one model run = one trace
# ------------------------------------------------------------------------------------------------------------------
class _Run:
    def __init__(self, unconstrained, enum):
        self.unconstrained = unconstrained      # name -> unconstrained value of a latent site
        self.enum = enum                        # "parallel" | ("fixed", v)
        self.plates = []                        # active (name, size, dim)
        self.masks = []                         # active masks
        self.trace = OrderedDict()              # name -> dict(kind, log_prob, plates, enumerated, value, log_jac)


_RUN = None


def _plate_shape():
    shape = [1] * max([-d for _, _, d in _RUN.plates], default=0)
    for _, size, dim in _RUN.plates:
        shape[dim] = size
    return tuple(shape)


def _apply_mask(lp):
    for m in _RUN.masks:
        lp = np.where(m, lp, 0.0)
    return lp


def _constrain(fn, u):
    """(3): biject_to(support) and its log |det J|."""
    u = np.asarray(u, dtype=np.float64)
    if fn.support == "real":
        return u, 0.0
    if fn.support == "positive":
        return np.exp(u), float(np.sum(u))
    if fn.support == "unit_interval":
        return sps.expit(u), float(np.sum(-np.logaddexp(0.0, -u) - np.logaddexp(0.0, u)))
    raise NotImplementedError(fn.support)


def sample(name, fn, obs=None, infer=None, **kw):
    assert not kw, kw
    assert name not in _RUN.trace, name
    pshape = _plate_shape()
    batch = np.broadcast_shapes(tuple(fn.batch_shape), pshape)       # (4)
    plates = {d: s for _, s, d in _RUN.plates}
    site = dict(kind="sample", plates=plates, enumerated=False, log_jac=0.0)
    if obs is not None:
        value = np.asarray(obs)
        site["observed"] = True
    elif infer and infer.get("enumerate") == "parallel":
        site["enumerated"] = True
        support = fn.enumerate_support()
        if _RUN.enum == "parallel":                                   # (5): a new axis left of every plate axis
            value = support.reshape((-1,) + (1,) * MAX_PLATE_NESTING)
        else:
            value = np.full(batch, support[_RUN.enum[1]])
        site["support_size"] = len(support)
    else:
        assert name in _RUN.unconstrained, f"latent site {name!r} has no value"
        value, site["log_jac"] = _constrain(fn, _RUN.unconstrained[name])
        assert value.shape == batch + tuple(fn.event_shape), (name, value.shape, batch, fn.event_shape)
    lp = fn.log_prob(value)
    lp = np.broadcast_to(lp, np.broadcast_shapes(lp.shape, batch))
    site["log_prob"] = _apply_mask(lp)
    site["value"] = value
    _RUN.trace[name] = site
    return value


def deterministic(name, value):
    _RUN.trace[name] = dict(kind="deterministic", value=np.asarray(value))
    return value


def factor(name, log_factor):
    lp = np.asarray(log_factor, dtype=np.float64)
    lp = np.broadcast_to(lp, np.broadcast_shapes(lp.shape, _plate_shape()))
    _RUN.trace[name] = dict(kind="sample", plates={d: s for _, s, d in _RUN.plates}, enumerated=False, log_jac=0.0,
                            log_prob=_apply_mask(lp), value=None, observed=True)


@contextlib.contextmanager
def plate(name, size, dim=None, **kw):
    assert dim is not None and dim < 0 and not kw
    _RUN.plates.append((name, int(size), dim))
    try:
        yield
    finally:
        _RUN.plates.pop()


@contextlib.contextmanager
def mask(mask=None):
    _RUN.masks.append(np.asarray(mask, dtype=bool))
    try:
        yield
    finally:
        _RUN.masks.pop()


# ------------------------------------------------------------------------------------------------------------------
# distributions (log_prob as numpyro states them)
# ------------------------------------------------------------------------------------------------------------------
class Distribution:
    support = "real"
    event_shape = ()

    def expand(self, batch_shape):
        return _Expanded(self, tuple(batch_shape))

    def to_event(self, n=None):
        return _Independent(self, len(self.batch_shape) if n is None else n)

    def enumerate_support(self):
        raise NotImplementedError


class _Expanded(Distribution):
    def __init__(self, base, shape):
        self.base, self.batch_shape, self.event_shape, self.support = base, shape, base.event_shape, base.support

    def log_prob(self, v):
        lp = self.base.log_prob(v)
        return np.broadcast_to(lp, np.broadcast_shapes(lp.shape, self.batch_shape))

    def enumerate_support(self):
        return self.base.enumerate_support()


class _Independent(Distribution):
    def __init__(self, base, n):
        self.base, self.n, self.support = base, n, base.support
        self.batch_shape = tuple(base.batch_shape[: len(base.batch_shape) - n])
        self.event_shape = tuple(base.batch_shape[len(base.batch_shape) - n:]) + tuple(base.event_shape)

    def log_prob(self, v):
        lp = self.base.log_prob(v)
        return lp.sum(axis=tuple(range(-self.n, 0))) if self.n else lp


def _f(x):
    return np.asarray(x, dtype=np.float64)


class Normal(Distribution):
    def __init__(self, loc=0.0, scale=1.0):
        self.loc, self.scale = _f(loc), _f(scale)
        self.batch_shape = np.broadcast_shapes(self.loc.shape, self.scale.shape)

    def log_prob(self, v):
        return -0.5 * ((v - self.loc) / self.scale) ** 2 - np.log(np.sqrt(2.0 * np.pi) * self.scale)


class Laplace(Normal):
    def log_prob(self, v):
        return -np.abs(v - self.loc) / self.scale - np.log(2.0 * self.scale)


class HalfNormal(Distribution):
    support = "positive"

    def __init__(self, scale=1.0):
        self.scale = _f(scale)
        self.batch_shape = self.scale.shape

    def log_prob(self, v):
        return Normal(0.0, self.scale).log_prob(v) + np.log(2.0)


class Exponential(Distribution):
    support = "positive"

    def __init__(self, rate=1.0):
        self.rate = _f(rate)
        self.batch_shape = self.rate.shape

    def log_prob(self, v):
        return np.log(self.rate) - self.rate * v


class Beta(Distribution):
    support = "unit_interval"

    def __init__(self, concentration1, concentration0):
        self.a, self.b = _f(concentration1), _f(concentration0)
        self.batch_shape = np.broadcast_shapes(self.a.shape, self.b.shape)

    def log_prob(self, v):
        return (self.a - 1.0) * np.log(v) + (self.b - 1.0) * np.log1p(-v) - sps.betaln(self.a, self.b)


class Bernoulli(Distribution):
    support = "discrete"

    def __init__(self, probs=None, logits=None):
        assert probs is not None and logits is None
        self.probs = _f(probs)
        self.batch_shape = self.probs.shape

    def log_prob(self, v):
        ps = np.clip(self.probs, CLAMP.tiny, 1.0 - CLAMP.eps)        # (1) clamp_probs
        v = _f(v)
        with np.errstate(invalid="ignore"):
            return sps.xlogy(v, ps) + sps.xlog1py(1.0 - v, -ps)

    def enumerate_support(self):
        return np.arange(2)


class Poisson(Distribution):
    support = "discrete"

    def __init__(self, rate):
        self.rate = _f(rate)
        self.batch_shape = self.rate.shape

    def log_prob(self, v):
        v = _f(v)
        with np.errstate(invalid="ignore", divide="ignore"):
            return sps.xlogy(v, self.rate) - sps.gammaln(v + 1.0) - self.rate


class Binomial(Distribution):
    support = "discrete"

    def __init__(self, total_count=1, probs=None):
        self.n, self.probs = _f(total_count), _f(probs)
        self.batch_shape = np.broadcast_shapes(self.n.shape, self.probs.shape)

    def log_prob(self, v):
        v = _f(v)
        with np.errstate(invalid="ignore", divide="ignore"):
            return (sps.gammaln(self.n + 1.0) - sps.gammaln(v + 1.0) - sps.gammaln(self.n - v + 1.0)
                    + sps.xlogy(v, self.probs) + sps.xlog1py(self.n - v, -self.probs))


class Categorical(Distribution):
    support = "discrete"

    def __init__(self, probs=None, logits=None):
        assert logits is not None and probs is None
        logits = _f(logits)
        self.logits = logits - sps.logsumexp(logits, axis=-1, keepdims=True)     # (2)
        self.batch_shape = self.logits.shape[:-1]

    def log_prob(self, v):
        v = np.asarray(v)
        shape = np.broadcast_shapes(v.shape, self.batch_shape)
        lg = np.broadcast_to(self.logits, shape + self.logits.shape[-1:])
        idx = np.broadcast_to(v, shape).astype(np.int64)[..., None]
        return np.take_along_axis(lg, idx, axis=-1)[..., 0]

    def enumerate_support(self):
        return np.arange(self.logits.shape[-1])


# ------------------------------------------------------------------------------------------------------------------
# import machinery: functional modules for the names above, MagicMock for the rest of jax / numpyro / funsor
# ------------------------------------------------------------------------------------------------------------------
class _Functional(types.ModuleType):
    """A module whose listed names are real and whose every other attribute is a MagicMock (import-time only)."""

    def __init__(self, name, names, fallback=None):
        super().__init__(name)
        self.__path__ = []
        self.__dict__.update(names)
        self._fallback = fallback

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        if self._fallback is not None and hasattr(self._fallback, item):
            return getattr(self._fallback, item)
        m = mock.MagicMock(name=f"{self.__name__}.{item}")
        setattr(self, item, m)
        return m


class _ConcretizationTypeError(Exception):
    pass


def _functional_modules():
    jnp = _Functional("jax.numpy", {}, fallback=np)
    jnn = _Functional("jax.nn", dict(sigmoid=sps.expit))
    jsp = _Functional("jax.scipy.special", dict(logsumexp=sps.logsumexp))
    jscipy = _Functional("jax.scipy", dict(special=jsp))
    jerr = _Functional("jax.errors", dict(ConcretizationTypeError=_ConcretizationTypeError))
    jax = _Functional("jax", dict(numpy=jnp, nn=jnn, scipy=jscipy, errors=jerr))
    dist = _Functional("numpyro.distributions", dict(
        Distribution=Distribution, Normal=Normal, Laplace=Laplace, HalfNormal=HalfNormal, Exponential=Exponential, Beta=Beta,
        Bernoulli=Bernoulli, Poisson=Poisson, Binomial=Binomial, Categorical=Categorical))
    handlers = _Functional("numpyro.handlers", dict(mask=mask))
    numpyro = _Functional("numpyro", dict(sample=sample, plate=plate, deterministic=deterministic, factor=factor,
                                          distributions=dist, handlers=handlers))
    return {m.__name__: m for m in (jax, jnp, jnn, jscipy, jsp, jerr, numpyro, dist, handlers)}


class _ShimFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    roots = ("jax", "numpyro", "funsor")

    def __init__(self):
        self.functional = _functional_modules()

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        if spec.name in self.functional:
            m = self.functional[spec.name]
        else:
            m = mock.MagicMock(name=spec.name)
            m.__path__ = []
        m.__spec__ = spec
        return m

    def exec_module(self, module):
        pass


def load_reference():
    sys.meta_path.insert(0, _ShimFinder())
    sys.path.insert(0, REFERENCE)
    import biolith.models  # noqa: F401
    import biolith.utils.data  # noqa: F401
    mods = {k: sys.modules[f"biolith.models.{k}"] for k in ("occu", "occu_rn", "occu_cop", "nmixture")}
    return mods, sys.modules["biolith.utils.data"].prepare_data


# ------------------------------------------------------------------------------------------------------------------
# running a model and contracting its trace
# ------------------------------------------------------------------------------------------------------------------
def run_model(model_fn, model_args, unconstrained, enum="parallel"):
    global _RUN
    _RUN = _Run(unconstrained, enum)
    try:
        with contextlib.redirect_stdout(io.StringIO()):       # RightTruncatedPoisson prints its cutoff advice
            model_fn(**model_args)
        return _RUN.trace
    finally:
        _RUN = None


def log_joint_parallel(trace):
    """(5): sum-product over the trace of a run with the enumeration axis in place."""
    e_ax = -(MAX_PLATE_NESTING + 1)
    enum_sites = [s for s in trace.values() if s["kind"] == "sample" and s["enumerated"]]
    assert len(enum_sites) <= 1
    total, jac, dependent = 0.0, 0.0, []
    for name, s in trace.items():
        if s["kind"] != "sample":
            continue
        jac += s["log_jac"]
        lp = s["log_prob"]
        if enum_sites and lp.ndim >= -e_ax and lp.shape[e_ax] > 1:
            assert all(n == 1 for n in lp.shape[:e_ax]), (name, lp.shape)
            dependent.append((s, lp.reshape(lp.shape[e_ax:])))
        else:
            total += float(np.sum(lp))
    if enum_sites:
        zp = enum_sites[0]["plates"]
        acc = 0.0
        for s, lp in dependent:
            extra = tuple(d for d in s["plates"] if d not in zp)          # plates the enumerated site is not in: summed first
            acc = acc + (lp.sum(axis=extra, keepdims=True) if extra else lp)
        total += float(np.sum(sps.logsumexp(acc, axis=0)))
    return total, jac


def log_joint_by_value(model_fn, model_args, unconstrained):
    """The same quantity with NO enumeration axis: the body run once per value of the enumerated site."""
    t0 = run_model(model_fn, model_args, unconstrained, enum=("fixed", 0))
    enum_name = [k for k, s in t0.items() if s["kind"] == "sample" and s["enumerated"]]
    assert len(enum_name) == 1
    K = t0[enum_name[0]]["support_size"]
    zp = t0[enum_name[0]]["plates"]
    zshape = t0[enum_name[0]]["log_prob"].shape
    per_value, fixed_total, jac = [], None, None
    for v in range(K):
        t = t0 if v == 0 else run_model(model_fn, model_args, unconstrained, enum=("fixed", v))
        acc, fixed, j = np.zeros(zshape), 0.0, 0.0
        for name, s in t.items():
            if s["kind"] != "sample":
                continue
            j += s["log_jac"]
            lp = s["log_prob"]
            if name == enum_name[0] or (set(zp) <= set(s["plates"]) and s.get("observed") and name == "y"):
                extra = tuple(d for d in s["plates"] if d not in zp)
                acc = acc + (lp.sum(axis=extra) if extra else lp)
            else:
                fixed += float(np.sum(lp))
        per_value.append(acc)
        if fixed_total is None:
            fixed_total, jac = fixed, j
        else:
            assert abs(fixed - fixed_total) <= 1e-9 * max(1.0, abs(fixed_total)), (fixed, fixed_total)
    return fixed_total + float(np.sum(sps.logsumexp(np.stack(per_value), axis=0))), jac


def potential(model_fn, model_args, unconstrained):
    total, jac = log_joint_parallel(run_model(model_fn, model_args, unconstrained))
    return -(total + jac)


# ------------------------------------------------------------------------------------------------------------------
# cases
# ------------------------------------------------------------------------------------------------------------------
def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def f32(a):
    return None if a is None else np.asarray(a, dtype=np.float32).astype(np.float64)


# case -> (model, simulator kwargs, model kwargs)
CASES = OrderedDict([
    # occu.py:136-242
    ("default", ("occu", dict(), dict())),
    ("missing", ("occu", dict(simulate_missing=True), dict())),
    ("missing_3periods", ("occu", dict(simulate_missing=True, n_periods=3), dict())),
    ("small_3x3", ("occu", dict(n_sites=300, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7), dict())),
    ("two_species", ("occu", dict(n_species=2, n_sites=30, simulate_missing=True), dict())),                       # occu.py:478-492
    ("fp_constant", ("occu", dict(simulate_missing=True, prob_fp_constant=0.1), dict(false_positives_constant=True))),   # occu.py:495-499
    ("fp_unoccupied", ("occu", dict(n_sites=150, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=49, prob_fp_unoccupied=0.08,
                                    random_seed=11), dict(false_positives_unoccupied=True))),
    ("two_species_fp", ("occu", dict(n_species=2, n_sites=40, n_obs_covs=2, simulate_missing=True, prob_fp_constant=0.05, random_seed=4),
                        dict(false_positives_constant=True))),
    ("re_site", ("occu", dict(n_sites=64, n_site_covs=2, n_obs_covs=1, random_seed=7, deployment_days_per_site=56),
                 dict(site_random_effects=True))),                                                                   # occu.py:170-173, 191-196
    ("re_obs", ("occu", dict(n_sites=64, n_site_covs=2, n_obs_covs=1, random_seed=7, deployment_days_per_site=56),
                dict(obs_random_effects=True))),                                                                    # occu.py:215-218
    ("re_both", ("occu", dict(n_sites=48, n_periods=2, n_site_covs=1, n_obs_covs=2, random_seed=9, deployment_days_per_site=42,
                              simulate_missing=True), dict(site_random_effects=True, obs_random_effects=True))),
    ("re_both_fp", ("occu", dict(n_sites=40, n_site_covs=1, n_obs_covs=1, random_seed=2, deployment_days_per_site=35,
                                 prob_fp_constant=0.07), dict(site_random_effects=True, obs_random_effects=True,
                                                              false_positives_constant=True))),
    # occu_rn.py:123-222
    ("rn_default", ("occu_rn", dict(), dict())),
    ("rn_missing", ("occu_rn", dict(n_sites=50, simulate_missing=True, deployment_days_per_site=56, n_periods=2, random_seed=1), dict())),
    ("rn_small_2x2", ("occu_rn", dict(n_sites=60, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=42, random_seed=3),
                      dict(max_abundance=40))),
    ("rn_fp", ("occu_rn", dict(n_sites=60, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=42, random_seed=3),
               dict(false_positives_constant=True))),
    ("rn_re_site", ("occu_rn", dict(n_sites=60, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=42, random_seed=3),
                    dict(site_random_effects=True, max_abundance=50))),
    # occu_cop.py:150-255
    ("cop_default", ("occu_cop", dict(), dict())),
    ("cop_missing", ("occu_cop", dict(simulate_missing=True), dict())),
    ("cop_small_2x2", ("occu_cop", dict(n_sites=80, n_site_covs=2, n_obs_covs=2, n_periods=2, deployment_days_per_site=42, random_seed=5),
                       dict())),
    ("cop_fp_constant", ("occu_cop", dict(n_sites=80, n_site_covs=2, n_obs_covs=2, n_periods=2, deployment_days_per_site=42, random_seed=5),
                         dict(false_positives_constant=True))),
    ("cop_fp_unoccupied", ("occu_cop", dict(simulate_missing=True), dict(false_positives_unoccupied=True))),
    ("cop_re_both", ("occu_cop", dict(n_sites=40, n_site_covs=1, n_obs_covs=1, deployment_days_per_site=35, random_seed=6),
                     dict(site_random_effects=True, obs_random_effects=True))),
    # nmixture.py:150-220
    ("nmix_default", ("nmixture", dict(), dict())),
    ("nmix_ref_test", ("nmixture", dict(simulate_missing=True), dict())),                                          # nmixture.py:372-380
    ("nmix_small_2x2", ("nmixture", dict(n_sites=70, n_site_covs=2, n_obs_covs=2, n_periods=2, deployment_days_per_site=42, random_seed=8),
                        dict(max_abundance=60))),
    ("nmix_re_site", ("nmixture", dict(n_sites=50, n_site_covs=1, n_obs_covs=1, deployment_days_per_site=35, random_seed=12),
                      dict(site_random_effects=True))),
    # non-default priors (a prior is written as [family, parameters...] and handed to the model as the shim's distribution object):
    # regression coefficients Normal(loc, scale) / Laplace(loc, scale) (occu.py:28-29, linear.py:28; utils/grid_search.py:366-371),
    # the false-positive Beta (occu.py:32-33), the effects' HalfNormal scales (occu.py:38-39), occu_cop's Exponential rate (occu_cop.py:33-34)
    ("priors_normal", ("occu", dict(n_sites=120, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=42, random_seed=21),
                       dict(prior_beta=["Normal", 0.3, 2.0], prior_alpha=["Normal", -0.2, 0.8]))),
    ("priors_laplace", ("occu", dict(n_sites=120, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=42, random_seed=21, simulate_missing=True),
                        dict(prior_beta=["Laplace", 0.0, 1.5], prior_alpha=["Laplace", 0.1, 0.7]))),
    ("priors_fp", ("occu", dict(n_sites=100, n_site_covs=1, n_obs_covs=2, deployment_days_per_site=49, prob_fp_unoccupied=0.05, random_seed=22),
                   dict(false_positives_unoccupied=True, prior_prob_fp_unoccupied=["Beta", 3.0, 8.0], prior_beta=["Normal", 0.0, 1.5]))),
    ("priors_re", ("occu", dict(n_sites=40, n_site_covs=1, n_obs_covs=1, random_seed=23, deployment_days_per_site=35),
                   dict(site_random_effects=True, obs_random_effects=True, prior_site_re_sd=["HalfNormal", 0.7], prior_obs_re_sd=["HalfNormal", 1.3],
                        prior_alpha=["Normal", 0.2, 1.2]))),
    ("priors_rn", ("occu_rn", dict(n_sites=60, n_site_covs=2, n_obs_covs=1, deployment_days_per_site=42, random_seed=24),
                   dict(prior_beta=["Normal", -0.3, 0.9], prior_alpha=["Normal", 0.0, 2.0], max_abundance=45))),
    ("priors_cop", ("occu_cop", dict(n_sites=70, n_site_covs=1, n_obs_covs=2, deployment_days_per_site=42, random_seed=25),
                    dict(false_positives_constant=True, prior_rate_fp_constant=["Exponential", 2.0], prior_beta=["Normal", 0.1, 1.5]))),
    ("priors_nmix", ("nmixture", dict(n_sites=60, n_site_covs=1, n_obs_covs=1, deployment_days_per_site=35, random_seed=26),
                     dict(prior_beta=["Normal", 0.5, 0.7], prior_alpha=["Normal", -0.5, 1.4], max_abundance=80))),
])
PRIOR_CLASSES = dict(Normal=Normal, Laplace=Laplace, Beta=Beta, HalfNormal=HalfNormal, Exponential=Exponential)


def materialise(mkw):
    """Model kwargs with every [family, parameters...] prior as the shim's distribution object."""
    return {k: (PRIOR_CLASSES[v[0]](*v[1:]) if isinstance(v, list) and v and v[0] in PRIOR_CLASSES else v) for k, v in mkw.items()}

SIMULATORS = dict(occu="simulate", occu_rn="simulate_rn", occu_cop="simulate_cop", nmixture="simulate_nmixture")
SITE_RE_NAMES = dict(occu=("site_re_occ", "site_re_det"), occu_cop=("site_re_occ", "site_re_det"),
                     occu_rn=("site_re_abu", "site_re_det"), nmixture=("site_re_abu", "site_re_det"))
FP_NAMES = dict(occu=dict(false_positives_constant="prob_fp_constant", false_positives_unoccupied="prob_fp_unoccupied"),
                occu_rn=dict(false_positives_constant="prob_fp_constant"),
                occu_cop=dict(false_positives_constant="rate_fp_constant", false_positives_unoccupied="rate_fp_unoccupied"))


def latent_shapes(model, mkw, S, N, T, J, Ks, Ko):
    """name -> shape of the UNCONSTRAINED value of every latent (non-enumerated) site, in the order the model samples them."""
    out = OrderedDict()
    for flag, name in FP_NAMES.get(model, {}).items():
        if mkw.get(flag):
            out[name] = ()
    if mkw.get("site_random_effects"):
        out["site_re_sd"] = ()
    if mkw.get("obs_random_effects"):
        out["obs_re_sd"] = ()
    out["beta"] = (S, Ks + 1)
    out["alpha"] = (S, Ko + 1)
    if mkw.get("site_random_effects"):
        for nm in SITE_RE_NAMES[model]:
            out[nm] = (N, S)
    if mkw.get("obs_random_effects"):
        out["obs_re"] = (J, T, N, S)
    return out


def make_thetas(model, shapes, truth, rng):
    """Five points: three init_to_uniform-like, one near the simulating truth, one in numpyro's clamp regime."""
    pts = []
    scalars = [k for k, s in shapes.items() if s == ()]

    def draw(width, eff_sd):
        v = OrderedDict()
        for k, s in shapes.items():
            if k in ("beta", "alpha"):
                v[k] = rng.uniform(-width, width, size=s)
            elif k in scalars:
                v[k] = rng.uniform(-1.5, 0.5)
            else:
                v[k] = rng.normal(size=s) * eff_sd
        return v

    for width in (2.0, 2.0, 1.0):
        pts.append(draw(width, 0.5))
    near = draw(0.1, 0.2)
    near["beta"] = near["beta"] + np.asarray(truth["beta"], dtype=np.float64).reshape(-1, near["beta"].shape[1])[: near["beta"].shape[0]]
    near["alpha"] = near["alpha"] + np.asarray(truth["alpha"], dtype=np.float64).reshape(-1, near["alpha"].shape[1])[: near["alpha"].shape[0]]
    pts.append(near)
    clamp = draw(0.5, 0.3)
    # occu-like: a detection probability within eps_f32 of one (|nu| > 15.9); counts: a large negative abundance intercept
    clamp["alpha"][:, 0] = 19.0 if model in ("occu", "occu_rn", "nmixture") else 2.5
    clamp["beta"][:, 0] = 18.0 if model == "occu" else (-3.0 if model in ("occu_rn", "nmixture") else 17.5)
    pts.append(clamp)
    labels = ["uniform2_a", "uniform2_b", "uniform1", "near_truth", "clamp_regime"]
    return list(zip(labels, pts))


def main():
    mods, prepare_data = load_reference()
    index = OrderedDict()
    for case, (model, skw, mkw) in CASES.items():
        mod = mods[model]
        model_fn, simulate = getattr(mod, model), getattr(mod, SIMULATORS[model])
        with contextlib.redirect_stdout(io.StringIO()):
            data, truth = simulate(**skw)
        site_covs, obs_covs, obs, dur, _, _ = prepare_data(data["site_covs"], data["obs_covs"], data["obs"], data.get("session_duration"))
        # float32 values (utils/data.py:135-140 hands the model float32 arrays), float64 arithmetic
        args = dict(site_covs=f32(site_covs), obs_covs=f32(obs_covs), obs=f32(obs), **materialise(mkw))
        if model == "occu_cop":
            args["session_duration"] = f32(dur)
        S, N, T, J = args["obs"].shape
        Ks, Ko = args["site_covs"].shape[1], args["obs_covs"].shape[3]
        shapes = latent_shapes(model, mkw, S, N, T, J, Ks, Ko)
        D = int(sum(int(np.prod(s)) for s in shapes.values()))
        rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) if False else int(hashlib.sha256(case.encode()).hexdigest()[:8], 16))
        points = []
        for label, vals in make_thetas(model, shapes, truth, rng):
            vals = OrderedDict((k, np.asarray(v, dtype=np.float32).astype(np.float64)) for k, v in vals.items())   # float32-exact thetas
            trace = run_model(model_fn, args, vals)
            total, jac = log_joint_parallel(trace)
            total2, jac2 = log_joint_by_value(model_fn, args, vals)
            assert abs(total - total2) <= 1e-10 * max(1.0, abs(total)) and jac == jac2, (case, label, total, total2)
            U = -(total + jac)
            assert np.isfinite(U), (case, label)
            pt = OrderedDict(label=label, U=U, log_joint=total, log_jacobian=jac,
                             unconstrained=OrderedDict((k, v.tolist()) for k, v in vals.items()))
            if D <= 24:      # central differences of the reference's own potential, per unconstrained coordinate in site order
                g = OrderedDict()
                for k, v in vals.items():
                    gk = np.zeros(v.shape)
                    for i in np.ndindex(*v.shape) if v.shape else [()]:
                        h = 1e-5 * max(1.0, abs(float(v[i])))
                        up, dn = OrderedDict(vals), OrderedDict(vals)
                        up[k], dn[k] = v.copy(), v.copy()
                        up[k][i] += h
                        dn[k][i] -= h
                        gk[i] = (potential(model_fn, args, up) - potential(model_fn, args, dn)) / (2.0 * h)
                    g[k] = gk.tolist()
                pt["grad_U_central_difference"] = g
            # what the enumeration does to the deterministic sites (upstream assumption ii)
            pt["deterministic_shapes"] = {k: list(s["value"].shape) for k, s in trace.items() if s["kind"] == "deterministic"}
            points.append(pt)
        entry = OrderedDict(
            case=case, model=model, simulator=SIMULATORS[model], simulator_kwargs=skw, model_kwargs=mkw,
            dims=dict(S=S, N=N, T=T, J=J, Ks=Ks, Ko=Ko, D=D),
            sha256=dict(site_covs=sha(args["site_covs"]), obs_covs=sha(args["obs_covs"]), obs=sha(args["obs"]),
                        **({"session_duration": sha(args["session_duration"])} if model == "occu_cop" else {})),
            latent_sites=OrderedDict((k, list(s)) for k, s in shapes.items()),
            clamp_dtype="float32", points=points,
        )
        with open(os.path.join(HERE, f"reference_logjoint_{case}.json"), "w") as f:
            json.dump(entry, f)
        index[case] = dict(model=model, D=D, U=[p["U"] for p in points])
        print(case, model, dict(S=S, N=N, T=T, J=J, D=D), " ".join(f"{p['U']:.6f}" for p in points))
    with open(os.path.join(HERE, "reference_logjoint_index.json"), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__":
    main()
