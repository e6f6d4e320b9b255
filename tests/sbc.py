"""Simulation-based calibration of a sampler + density pair (TEST INFRASTRUCTURE; Talts, Betancourt, Simpson, Vehtari, Gelman 2018).

    theta~ ~ prior,   y~ ~ p(y | theta~),   theta_1 .. theta_M ~ p(theta | y~)   =>   rank(theta~ among theta_1 .. theta_M) ~ Uniform{0 .. M}

exactly, for every coordinate and every data size -- a property of the MODEL, so it needs no restatement of the density and no
numpyro: a sampler that targets anything but the posterior of the generative model below (wrong mask rule, a transposed plate, a
biased transition, a step size adapted past the end of warm-up ...) bends the ranks' histogram.  The generative models are the
reference's (biolith/models/occu.py:136-242: psi per site, z per (site, period), y ~ Bernoulli(z p); occu_rn.py:123-222:
N ~ Poisson(lambda) truncated at max_abundance and renormalised, y ~ Bernoulli(1 - (1 - r)^N); nmixture.py:150-220: N ~ Poisson(lambda)
-- its `numpyro.factor` undoes the renormalisation, max_abundance is only the bound of the sum --, y ~ Binomial(N, p)), priors Normal(0, 1)
on every coefficient (their defaults) unless a test narrows one and says so.
"""
import numpy as np
from scipy.stats import chi2


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def prior_predictive(rng, model, n_sites, n_visits, ks, ko, n_periods=1, missing=0.1, max_abundance=100, fp=None, beta_scale=1.0):
    """-> (site_covs, obs_covs, obs, theta, kwargs): theta drawn from the prior, in the flat UNCONSTRAINED layout of the oracle / engine
    [beta_0 .. beta_ks, alpha_0 .. alpha_ko | logit(prob_fp) or log(rate_fp)] (a rank is invariant under the monotone maps); kwargs =
    what oracle.OracleData / engine.OccuDataset need beyond the arrays.  Priors are the reference's defaults: Normal(0, 1) coefficients,
    Beta(2, 5) false-positive probability (occu.py:32-33), Exponential(1) false-positive rate (occu_cop.py:32-33).
    (Random effects: prior_predictive_re below.)"""
    beta, alpha = rng.normal(size=ks + 1) * beta_scale, rng.normal(size=ko + 1)
    X = rng.normal(size=(n_sites, ks))
    W = rng.normal(size=(n_sites, n_periods, n_visits, ko))
    if model == "occu_dyn":   # the builder's dynamic model (oracle.literal_log_joint_dyn): theta = [beta_psi | beta_gamma | beta_eps | alpha]
        assert fp is None
        assert beta_scale == 1.0
        bg, be = rng.normal(size=ks + 1), rng.normal(size=ks + 1)
        psi, gam, eps = (_sigmoid(b[0] + X @ b[1:]) for b in (beta, bg, be))
        z = np.empty((n_sites, n_periods), dtype=bool)
        z[:, 0] = rng.uniform(size=n_sites) < psi
        for t in range(1, n_periods):
            z[:, t] = rng.uniform(size=n_sites) < np.where(z[:, t - 1], 1.0 - eps, gam)
        Y = ((rng.uniform(size=W.shape[:3]) < _sigmoid(alpha[0] + W @ alpha[1:])) & z[:, :, None]) * 1.0
        Y[rng.uniform(size=Y.shape) < missing] = np.nan
        return X.astype(np.float32), W.astype(np.float32), Y[None].astype(np.float32), np.concatenate([beta, bg, be, alpha]), dict(model="occu_dyn")
    parts, kw = [beta, alpha], dict(prior_beta=(0.0, float(beta_scale)))
    fp_c = fp_u = 0.0
    if fp is not None:
        v = rng.exponential() if model == "occu_cop" else rng.beta(2.0, 5.0)
        fp_c, fp_u = (v, 0.0) if fp == "constant" else (0.0, v)
        parts.append(np.array([np.log(v) if model == "occu_cop" else np.log(v / (1.0 - v))]))
    eta = beta[0] + X @ beta[1:]
    nu = alpha[0] + W @ alpha[1:]                                       # (N, T, J)
    kw["model"] = {"occu": "occu_fp" if fp else "occu"}.get(model, model)
    if model == "occu_rn":
        if fp:
            kw["re_fp_mode"] = fp
    elif fp or model == "occu_cop":
        kw["fp_mode"] = fp
    if model in ("occu", "occu_cop"):
        z = (rng.uniform(size=(n_sites, n_periods)) < _sigmoid(eta)[:, None])[:, :, None] * 1.0
        if model == "occu":       # occu.py:222-235
            Y = (rng.uniform(size=nu.shape) < 1.0 - (1.0 - z * _sigmoid(nu)) * (1.0 - fp_c) * (1.0 - (1.0 - z) * fp_u)) * 1.0
        else:                     # occu_cop.py:228-248
            dur = rng.uniform(0.5, 2.0, size=nu.shape)
            Y = rng.poisson(dur * (z * np.exp(nu) + (1.0 - z) * fp_u + fp_c)) * 1.0
            kw["session_duration"] = dur.astype(np.float32)
    else:
        assert n_periods == 1 and (fp is None or (model == "occu_rn" and fp == "constant"))   # occu_rn.py:135-137, 216-219
        kw["max_abundance"] = max_abundance
        p = _sigmoid(nu)
        if model == "occu_rn":
            # truncated Poisson, as the reference states it (utils/distributions.py:6-40): Categorical(logits = Poisson(lambda).log_prob(0 .. K))
            from scipy.special import gammaln
            n = np.arange(max_abundance + 1)
            logits = n[None, :] * eta[:, None] - gammaln(n + 1.0)[None, :]
            pmf = np.exp(logits - logits.max(1, keepdims=True))
            cdf = np.cumsum(pmf / pmf.sum(1, keepdims=True), axis=1)
            N = np.minimum((rng.uniform(size=n_sites)[:, None] > cdf).sum(1), max_abundance)
        else:
            # nmixture.py:183-194 adds the Categorical's normaliser back (numpyro.factor): the density is p(theta) sum_{N <= K} Poisson(N;
            # lambda) Binomial(y | N, p) = the joint of theta, y AND the event "every N <= K" under N ~ Poisson(lambda) untruncated.  Its
            # posterior is therefore the one given y and that event: a replication with some N > K is rejected and drawn again, whole.
            N = rng.poisson(np.minimum(np.exp(eta), 1e6))
            if N.max() > max_abundance:
                return prior_predictive(rng, model, n_sites, n_visits, ks, ko, n_periods, missing, max_abundance, fp, beta_scale)
        if model == "occu_rn":
            Y = (rng.uniform(size=p.shape) < 1.0 - (1.0 - p) ** N[:, None, None] * (1.0 - fp_c)) * 1.0
        elif model == "nmixture":
            Y = rng.binomial(N[:, None, None], p) * 1.0
        else:
            raise ValueError(model)
    Y[rng.uniform(size=Y.shape) < missing] = np.nan
    return X.astype(np.float32), W.astype(np.float32), Y[None].astype(np.float32), np.concatenate(parts), kw


def prior_predictive_re(rng, n_sites, n_visits, ks, ko, site_re=True, obs_re=False, missing=0.1):
    """occu with random effects (occu.py:167-171, 190-215): sd ~ HalfNormal(1), site effects on both linear predictors, observation
    effects on the detection's.  theta = [beta, alpha | log site_re_sd | log obs_re_sd |
    site_re_occ [N] | site_re_det [N] | obs_re [N][1][J]]."""
    beta, alpha = rng.normal(size=ks + 1), rng.normal(size=ko + 1)
    X = rng.normal(size=(n_sites, ks))
    W = rng.normal(size=(n_sites, 1, n_visits, ko))
    parts, tail = [beta, alpha], []
    re_occ = re_det = np.zeros(n_sites)
    re_obs = np.zeros((n_sites, 1, n_visits))

    def half_normal():
        return abs(rng.normal())

    if site_re:
        sd = half_normal()
        parts.append(np.array([np.log(sd)]))
        re_occ, re_det = rng.normal(size=n_sites) * sd, rng.normal(size=n_sites) * sd
        tail += [re_occ, re_det]
    if obs_re:
        sdo = half_normal()
        parts.append(np.array([np.log(sdo)]))
        re_obs = rng.normal(size=re_obs.shape) * sdo
        tail.append(re_obs.reshape(-1))
    z = (rng.uniform(size=(n_sites, 1)) < _sigmoid(beta[0] + X @ beta[1:] + re_occ)[:, None])[:, :, None]
    Y = ((rng.uniform(size=re_obs.shape) < _sigmoid(alpha[0] + W @ alpha[1:] + re_det[:, None, None] + re_obs)) & z) * 1.0
    Y[rng.uniform(size=Y.shape) < missing] = np.nan
    kw = dict(model="occu_re", site_random_effects=site_re, obs_random_effects=obs_re)
    return X.astype(np.float32), W.astype(np.float32), Y[None].astype(np.float32), np.concatenate(parts + tail), kw


def rank_of_truth(draws, theta, thin, keep=None):
    """draws (chains, n, D) -> (ranks (D,), M): every `thin`-th draw of every chain (the first `keep` of them, so that M + 1 rank
    values split into equal bins), rank = how many lie below the truth."""
    kept = draws[:, thin - 1::thin].reshape(-1, draws.shape[-1])[:keep]
    return (kept < theta[None, :]).sum(0), kept.shape[0]


def uniformity(ranks, M, bins=10):
    """ranks (L, D) in 0 .. M -> chi-square statistic per coordinate against the uniform law on `bins` equal groups of rank values
    ((M + 1) must divide by `bins`), and the 99.9 % point of chi2(bins - 1)."""
    assert (M + 1) % bins == 0
    L = ranks.shape[0]
    counts = np.stack([np.bincount(ranks[:, d] // ((M + 1) // bins), minlength=bins) for d in range(ranks.shape[1])])
    stat = ((counts - L / bins) ** 2 / (L / bins)).sum(1)
    return stat, float(chi2.ppf(0.999, bins - 1)), counts


def run(sample_posterior, model, replications, seed, thin, keep=None, **shape):
    """`sample_posterior(X, W, Y, replication, **kwargs of the replication) -> draws (chains, n, D)`; -> (ranks (L, D), M)."""
    rng = np.random.default_rng(seed)
    ranks, M = [], None
    for l in range(replications):
        X, W, Y, theta, kw = prior_predictive(rng, model, **shape)
        r, M = rank_of_truth(np.asarray(sample_posterior(X, W, Y, l, **kw), dtype=np.float64), theta, thin, keep)
        ranks.append(r)
    return np.stack(ranks), M
