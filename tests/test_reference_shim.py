"""The NumPy stand-in for numpyro's distributions that `tests/golden/make_reference_logjoint.py` serves to the reference's model
functions (SURVEY section 8 row c): its nine `log_prob`s, its `.expand(...).to_event(...)` shapes and its
`biject_to(support)` transforms are checked here against TWO independent implementations that are in the build container --
`scipy.stats` and `torch.distributions` (the latter is what numpyro's distributions were modelled on: same parameter names, same
batch / event shape rules, same `biject_to` registry).  Agreement to 1e-12 turns "builder-written log_prob" into "agrees with two
third-party libraries".  NOT covered, and stated as an upstream assumption with its own check of the constants below: numpyro's
`clamp_probs` (Bernoulli probabilities clipped to [finfo(float32).tiny, 1 - finfo(float32).eps]) -- neither library clamps that way.

Where the reference uses them: `regression/linear.py:28` (Normal / Laplace priors through `.expand().to_event()`), `models/occu.py:147-242`
(HalfNormal sds, Beta false-positive probabilities, Bernoulli z and y), `occu_rn.py` / `nmixture.py` (Poisson, Binomial, Categorical
over the truncated abundances), `utils/distributions.py:6-40`, `occu_cop.py` (Exponential / Poisson rates).  CPU only; no GPU, no oracle."""
import importlib.util
import os

import numpy as np
import pytest
import torch
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_reference_logjoint", os.path.join(HERE, "golden", "make_reference_logjoint.py"))
shim = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(shim)          # (defines the classes; reads /root/reference only inside main())

td = torch.distributions
RNG = np.random.default_rng(20261004)
TOL = dict(rtol=1e-12, atol=1e-12)


def _t(x):
    return torch.as_tensor(np.asarray(x, dtype=np.float64))


def _both(got, sp, th):
    np.testing.assert_allclose(got, sp, **TOL)
    np.testing.assert_allclose(got, th.numpy(), **TOL)


def test_normal_laplace_halfnormal_exponential_beta_against_scipy_and_torch():
    loc, scale = RNG.normal(size=(7, 1)), RNG.uniform(0.2, 5.0, size=(1, 5))
    v = RNG.normal(size=(7, 5)) * 3
    _both(shim.Normal(loc, scale).log_prob(v), stats.norm(loc, scale).logpdf(v), td.Normal(_t(loc), _t(scale)).log_prob(_t(v)))
    _both(shim.Laplace(loc, scale).log_prob(v), stats.laplace(loc, scale).logpdf(v), td.Laplace(_t(loc), _t(scale)).log_prob(_t(v)))
    vp = RNG.uniform(1e-3, 9.0, size=(7, 5))
    _both(shim.HalfNormal(scale).log_prob(vp), stats.halfnorm(scale=scale).logpdf(vp), td.HalfNormal(_t(scale)).log_prob(_t(vp)))
    rate = RNG.uniform(0.1, 4.0, size=(1, 5))
    _both(shim.Exponential(rate).log_prob(vp), stats.expon(scale=1.0 / rate).logpdf(vp), td.Exponential(_t(rate)).log_prob(_t(vp)))
    a, b = RNG.uniform(0.5, 6.0, size=(7, 1)), RNG.uniform(0.5, 6.0, size=(1, 5))
    vu = RNG.uniform(1e-4, 1 - 1e-4, size=(7, 5))
    _both(shim.Beta(a, b).log_prob(vu), stats.beta(a, b).logpdf(vu), td.Beta(_t(a), _t(b)).log_prob(_t(vu)))
    # the defaults the model files rely on (Normal() is standard; HalfNormal(scale) / Exponential(rate) positional)
    np.testing.assert_allclose(shim.Normal().log_prob(np.array(0.3)), stats.norm.logpdf(0.3), **TOL)
    assert shim.HalfNormal(2.0).support == "positive" and shim.Exponential(2.0).support == "positive" and shim.Beta(2.0, 3.0).support == "unit_interval"


def test_poisson_binomial_bernoulli_against_scipy_and_torch():
    rate = RNG.uniform(0.05, 30.0, size=(6, 1))
    k = RNG.integers(0, 60, size=(6, 9)).astype(np.float64)
    _both(shim.Poisson(rate).log_prob(k), stats.poisson(rate).logpmf(k), td.Poisson(_t(rate)).log_prob(_t(k)))
    n = RNG.integers(0, 40, size=(6, 1)).astype(np.float64)
    p = RNG.uniform(0.02, 0.98, size=(1, 9))
    kk = np.floor(RNG.uniform(0, 1, size=(6, 9)) * (n + 1))
    _both(shim.Binomial(n, p).log_prob(kk), stats.binom(n, p).logpmf(kk), td.Binomial(_t(n), probs=_t(p)).log_prob(_t(kk)))
    # total_count = 0 and the corners of the support (nmixture.py: Binomial(N, p) with N = 0 rows)
    np.testing.assert_allclose(shim.Binomial(np.array(0.0), np.array(0.3)).log_prob(np.array(0.0)), 0.0, atol=1e-15)
    np.testing.assert_allclose(shim.Binomial(np.array(5.0), np.array(0.3)).log_prob(np.array([0.0, 5.0])),
                               stats.binom(5, 0.3).logpmf([0, 5]), **TOL)
    # Bernoulli away from the clamp: probabilities in [1e-6, 1 - 1e-6] are untouched by it
    pb = RNG.uniform(1e-6, 1 - 1e-6, size=(6, 9))
    y = (RNG.uniform(size=(6, 9)) < 0.5).astype(np.float64)
    _both(shim.Bernoulli(probs=pb).log_prob(y), stats.bernoulli(pb).logpmf(y), td.Bernoulli(probs=_t(pb)).log_prob(_t(y)))
    assert list(shim.Bernoulli(probs=pb).enumerate_support()) == [0, 1]
    assert td.Bernoulli(probs=_t(0.3)).enumerate_support().reshape(-1).tolist() == [0.0, 1.0]


def test_bernoulli_clamp_constants_are_float32s():
    """UPSTREAM-ASSUMED (1), the one thing neither library restates: clamp_probs clips to [tiny, 1 - eps] of the probabilities' dtype,
    float32 in the reference (utils/data.py:135-140).  The constants, and that the clamp is what makes log_prob finite at p = 0 and 1."""
    assert shim.CLAMP.eps == np.float32(2.0) ** -23 and shim.CLAMP.tiny == np.float32(2.0) ** -126
    lp = shim.Bernoulli(probs=np.array([0.0, 1.0, 0.0, 1.0])).log_prob(np.array([1.0, 0.0, 0.0, 1.0]))
    np.testing.assert_allclose(lp, [np.log(2.0 ** -126), np.log(2.0 ** -23), np.log1p(-(2.0 ** -126)), np.log1p(-(2.0 ** -23))], rtol=1e-12)
    # torch clamps too (clamp_probs, to ITS dtype's eps on both sides): same mechanism, different constants -- hence excluded above
    assert np.isfinite(td.Bernoulli(probs=_t([0.0, 1.0])).log_prob(_t([1.0, 0.0])).numpy()).all()


def test_categorical_logits_are_renormalised_as_in_torch():
    """UPSTREAM-ASSUMED (2): Categorical(logits) subtracts logsumexp (occu_rn.py / nmixture.py hand it UNnormalised truncated-Poisson logits)."""
    logits = RNG.normal(size=(5, 3, 11)) * 4 + 7.0                  # far from normalised
    v = RNG.integers(0, 11, size=(5, 3))
    got = shim.Categorical(logits=logits).log_prob(v)
    th = td.Categorical(logits=_t(logits))
    np.testing.assert_allclose(got, th.log_prob(torch.as_tensor(v)).numpy(), **TOL)
    ref = logits - np.log(np.exp(logits - logits.max(-1, keepdims=True)).sum(-1, keepdims=True)) - logits.max(-1, keepdims=True)
    np.testing.assert_allclose(got, np.take_along_axis(ref, v[..., None], -1)[..., 0], **TOL)
    assert shim.Categorical(logits=logits).batch_shape == tuple(th.batch_shape) == (5, 3)
    assert list(shim.Categorical(logits=logits).enumerate_support()) == th.enumerate_support(expand=False).reshape(-1).tolist()
    # a value array with an enumeration axis on the left broadcasts against the batch shape (parallel enumeration, assumption (5))
    ve = np.arange(11).reshape(11, 1, 1)
    got = shim.Categorical(logits=logits).log_prob(ve)
    np.testing.assert_allclose(got, th.log_prob(torch.as_tensor(ve)).numpy(), **TOL)
    np.testing.assert_allclose(np.exp(got).sum(0), 1.0, rtol=1e-12)


def test_expand_and_to_event_shapes_and_sums_as_in_torch():
    """linear.py:28: `Normal(0, sd).expand([n_covs + 1, n_species]).to_event(1)` style priors."""
    for base_s, base_t in ((shim.Normal(0.0, 1.5), td.Normal(_t(0.0), _t(1.5))), (shim.Laplace(0.2, 0.7), td.Laplace(_t(0.2), _t(0.7)))):
        for shape, n in (((4, 3), 1), ((4, 3), 2), ((4,), 1), ((2, 4, 3), 2), ((4, 3), 0)):
            s = base_s.expand(list(shape)).to_event(n)
            t = td.Independent(base_t.expand(list(shape)), n)
            assert tuple(s.batch_shape) == tuple(t.batch_shape) and tuple(s.event_shape) == tuple(t.event_shape), (shape, n)
            v = RNG.normal(size=shape)
            np.testing.assert_allclose(s.log_prob(v), t.log_prob(_t(v)).numpy(), **TOL)
        s = base_s.expand([4, 3]).to_event()                       # numpyro: to_event() with no argument takes every batch dimension
        assert s.batch_shape == () and s.event_shape == (4, 3)
    # a batch-shaped base: HalfNormal(scale[3]) expanded to (4, 3)
    sc = RNG.uniform(0.5, 2.0, size=3)
    v = RNG.uniform(0.1, 3.0, size=(4, 3))
    np.testing.assert_allclose(shim.HalfNormal(sc).expand([4, 3]).to_event(1).log_prob(v),
                               td.Independent(td.HalfNormal(_t(sc)).expand([4, 3]), 1).log_prob(_t(v)).numpy(), **TOL)


def test_unconstrained_transforms_and_jacobians_as_torch_biject_to():
    """UPSTREAM-ASSUMED (3): a positive site is sampled as exp(u), a unit-interval site as sigmoid(u), with log |dx/du| added."""
    u = RNG.normal(size=(3, 4)) * 2.5
    for dist_s, dist_t in ((shim.HalfNormal(1.0), td.HalfNormal(_t(1.0))), (shim.Exponential(2.0), td.Exponential(_t(2.0))),
                           (shim.Beta(2.0, 3.0), td.Beta(_t(2.0), _t(3.0))), (shim.Normal(0.0, 1.0), td.Normal(_t(0.0), _t(1.0)))):
        tr = td.biject_to(dist_t.support)
        x, lj = shim._constrain(dist_s, u)
        xt = tr(_t(u))
        np.testing.assert_allclose(x, xt.numpy(), rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(lj, float(tr.log_abs_det_jacobian(_t(u), xt).sum()), rtol=1e-12, atol=1e-12)
    # ... and the potential of the unconstrained site is the density of u: it integrates to one (trapezoid on a wide grid)
    g = np.linspace(-30.0, 12.0, 200001)
    for d in (shim.HalfNormal(1.7), shim.Exponential(0.6), shim.Beta(2.0, 5.0)):
        x = shim._constrain(d, g)[0]
        lj = {"positive": g, "unit_interval": -np.logaddexp(0.0, -g) - np.logaddexp(0.0, g)}[d.support]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            f = np.exp(d.log_prob(x) + lj)
        f = np.where(np.isfinite(f), f, 0.0)
        assert abs(np.trapezoid(f, g) - 1.0) < 1e-6, type(d).__name__


@pytest.mark.parametrize("name", ["Normal", "Laplace", "HalfNormal", "Exponential", "Beta", "Bernoulli", "Poisson", "Binomial", "Categorical"])
def test_every_served_distribution_is_covered(name):
    """The nine names `numpyro.distributions` serves to the model files (make_reference_logjoint.py `_functional_modules`)."""
    assert hasattr(shim, name) and issubclass(getattr(shim, name), shim.Distribution)
    served = shim._functional_modules()
    dist_mod = [m for k, m in served.items() if k.endswith("numpyro.distributions")]
    assert dist_mod and getattr(dist_mod[0], name) is getattr(shim, name)
