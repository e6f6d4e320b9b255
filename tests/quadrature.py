"""Posterior moments by NUMERICAL INTEGRATION for occupancy models with two or three coefficients (TEST INFRASTRUCTURE).

An anchor for the SAMPLER that neither side of the HIP-vs-oracle comparison wrote: exp(-U) on a tensor grid, U from the oracle's
potential (itself equal to the reference's own model functions to 2e-15: tests/test_reference_logjoint.py).  The grid is centred on
the mode and reaches, per axis, to where the potential has risen by 30 nats; the integrand is below 1e-9 of its peak on every face."""
import numpy as np


def tiny_occupancy_data(n_sites=80, n_visits=4, ks=0, seed=0):
    """occu with `ks` site covariates and no observation covariate: theta = (beta_0 .. beta_ks, alpha_0)."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n_sites, ks)) * 0.8
    W = np.zeros((n_sites, 1, n_visits, 0))
    eta = 0.4 + (X @ np.full(ks, -0.7) if ks else 0.0)
    z = rng.uniform(size=n_sites) < 1.0 / (1.0 + np.exp(-eta))
    Y = ((rng.uniform(size=(n_sites, 1, n_visits)) < 0.35) & z[:, None, None]) * 1.0
    Y[rng.uniform(size=Y.shape) < 0.1] = np.nan
    return X.astype(np.float32), W.astype(np.float32), Y[None].astype(np.float32)


def grid_posterior(od, points_per_axis):
    """-> dict(mean, sd, corr, marginal quantile function of coordinate 0) of exp(-U) by the trapezoidal rule on a tensor grid."""
    D = od.D
    # mode by Newton-free descent on the oracle's gradient (BFGS from scipy), curvature by central differences of the gradient
    from scipy.optimize import minimize

    res = minimize(lambda t: od.potential_grad(t)[0], np.zeros(D), jac=lambda t: od.potential_grad(t)[1], method="BFGS")
    mode = res.x
    H = np.empty((D, D))
    for i in range(D):
        h = 1e-4
        e = np.zeros(D); e[i] = h
        H[i] = (od.potential_grad(mode + e)[1] - od.potential_grad(mode - e)[1]) / (2 * h)
    sd0 = np.sqrt(np.diag(np.linalg.inv(0.5 * (H + H.T))))
    # per axis, walk outwards from the mode until the potential has risen by 30 nats on the line through the mode (the occupancy intercept's
    # posterior has a long right tail: psi saturates and only the prior holds it), then a margin; every face of the box is checked below
    lo, hi = mode.copy(), mode.copy()
    for d in range(D):
        for side, edge in ((-1.0, lo), (1.0, hi)):
            t = mode.copy()
            while od.potential_grad(t)[0] - res.fun < 30.0:
                t[d] += side * sd0[d]
            edge[d] = t[d] + side * 2.0 * sd0[d]
    axes = [np.linspace(lo[d], hi[d], points_per_axis) for d in range(D)]
    grid = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, D)
    U = od.potential_grad(grid)[0].reshape((points_per_axis,) * D)
    w = np.exp(-(U - U.min()))
    for d in range(D):                                     # the integrand has died out on every face of the box
        assert max(np.take(w, 0, axis=d).max(), np.take(w, -1, axis=d).max()) < 1e-9, (d, np.take(w, 0, axis=d).max(), np.take(w, -1, axis=d).max())
    w /= w.sum()
    mean = np.array([(w * g).sum() for g in np.meshgrid(*axes, indexing="ij")])
    cov = np.empty((D, D))
    G = np.meshgrid(*axes, indexing="ij")
    for i in range(D):
        for j in range(D):
            cov[i, j] = (w * (G[i] - mean[i]) * (G[j] - mean[j])).sum()
    sd = np.sqrt(np.diag(cov))
    marg0 = w.sum(axis=tuple(range(1, D)))
    cdf0 = np.cumsum(marg0) - 0.5 * marg0        # (the cell around a grid point is half below it)
    return dict(mean=mean, sd=sd, corr=cov / np.outer(sd, sd), axis0=axes[0], cdf0=cdf0, mode=mode, laplace_sd=sd0)


def check_draws(draws, q, ess_fn, sd_rtol=0.03):
    """draws (chains, n, D) against the quadrature `q`: means within 4 Monte-Carlo standard errors (ESS-based), standard deviations within
    `sd_rtol`, correlations within 0.03, and the empirical CDF of coordinate 0 at the quadrature's 5 / 25 / 50 / 75 / 95 % points."""
    C, n, D = draws.shape
    flat = draws.reshape(-1, D).astype(np.float64)
    ess = np.asarray(ess_fn(draws.astype(np.float64)))
    mcse = flat.std(0) / np.sqrt(ess)
    assert np.all(np.abs(flat.mean(0) - q["mean"]) <= 4.0 * mcse), (flat.mean(0), q["mean"], mcse)
    assert np.all(np.abs(flat.std(0) / q["sd"] - 1.0) <= sd_rtol), (flat.std(0), q["sd"])
    assert np.all(np.abs(np.corrcoef(flat.T) - q["corr"]) <= 0.03), (np.corrcoef(flat.T), q["corr"])
    for level in (0.05, 0.25, 0.5, 0.75, 0.95):
        x = np.interp(level, q["cdf0"], q["axis0"])
        emp = (flat[:, 0] <= x).mean()
        assert abs(emp - level) <= 4.0 * np.sqrt(level * (1 - level) / ess[0]) + 2e-3, (level, emp)
