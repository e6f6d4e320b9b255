"""The EXACT marginal posterior of a random-effects scale, by nested numerical integration (TEST INFRASTRUCTURE).

occu with site or observation random effects (biolith/models/occu.py:170-173, 191-196, 215-218): sd ~ HalfNormal(1), the effects
Normal(0, sd) -- a funnel in (log sd, effects) that NUTS with one step size explores badly where sd is small.  To tell the funnel's
bias from an implementation's, the one quantity the simulation-based calibration fails on -- the marginal of
u = log sd -- is computed here without any sampler:

    p(beta0, alpha0, u | y)  is proportional to  prior(beta0) prior(alpha0) HalfNormal(e^u) e^u  x  prod_i L_i(beta0, alpha0, e^u)

for data whose covariates are all ZERO (the slopes then have their Normal(0, 1) prior as posterior and drop out), where a site's
likelihood with its effects integrated out is, by Gauss-Hermite quadrature,

    site effects a_i (occupancy), b_i (detection), both ~ Normal(0, sd):
        L_i = E_a[psi(beta0 + a)] * E_b[ prod_j p(alpha0 + b)^y (1 - p(alpha0 + b))^(1 - y) ]  +  (1 - E_a[psi(beta0 + a)]) * [no detection at i]
    observation effects c_ij ~ Normal(0, sd), one per observation:
        L_i = psi(beta0) * prod_j E_c[ p(alpha0 + c)^y (1 - p(alpha0 + c))^(1 - y) ]  +  (1 - psi(beta0)) * [no detection at i]

(z_i summed out as the model's enumeration does; a missing visit contributes 1), on a tensor grid over (beta0, alpha0, u).  The grid's
faces are checked to hold less than 1e-7 of the mass."""
import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def zero_covariate_data(n_sites=40, n_visits=6, sd=0.2, site_re=True, seed=0, beta0=0.3, alpha0=-0.2, missing=0.1):
    """Data from the generative model with intercepts only (the covariate columns exist and are zero: the kernels want Ks, Ko >= 1)."""
    rng = np.random.default_rng(seed)
    X = np.zeros((n_sites, 1), dtype=np.float32)
    W = np.zeros((n_sites, 1, n_visits, 1), dtype=np.float32)
    a = rng.normal(size=n_sites) * sd if site_re else np.zeros(n_sites)
    b = rng.normal(size=n_sites) * sd if site_re else np.zeros(n_sites)
    c = np.zeros((n_sites, 1, n_visits)) if site_re else rng.normal(size=(n_sites, 1, n_visits)) * sd
    z = rng.uniform(size=n_sites) < _sigmoid(beta0 + a)
    Y = ((rng.uniform(size=c.shape) < _sigmoid(alpha0 + b[:, None, None] + c)) & z[:, None, None]) * 1.0
    Y[rng.uniform(size=Y.shape) < missing] = np.nan
    return X, W, Y[None].astype(np.float32)


def log_sd_marginal(Y, site_re=True, prior_sd_scale=1.0, n_beta=61, n_alpha=61, n_u=161, gh=48,
                    box=((-7.5, 7.5), (-7.5, 7.5), (-9.0, 2.2))):
    """-> (u axis, density of u = log sd on it (normalised, trapezoid), cdf on it).  Y: (1, N, 1, J) with NaN = missing."""
    y = np.asarray(Y, dtype=np.float64)[0, :, 0, :]                    # (N, J)
    seen = ~np.isnan(y)
    det = np.where(seen, y, 0.0)
    nondet = np.where(seen, 1.0 - y, 0.0)
    none = (det.sum(1) == 0).astype(np.float64)                          # no detection at the site
    t, w = np.polynomial.hermite_e.hermegauss(gh)                        # E_{x ~ N(0,1)} f(x) = sum w f(t) / sqrt(2 pi)
    w = w / np.sqrt(2.0 * np.pi)
    B, A, U = (np.linspace(lo, hi, n) for (lo, hi), n in zip(box, (n_beta, n_alpha, n_u)))
    logpost = np.empty((n_beta, n_alpha, n_u))
    for iu, u in enumerate(U):
        sd = np.exp(u)
        if site_re:
            Epsi = (_sigmoid(B[:, None] + sd * t[None, :]) * w[None, :]).sum(1)                      # (n_beta,)
            x = A[:, None] + sd * t[None, :]                                                          # (n_alpha, gh)
            lp, lq = -np.logaddexp(0.0, -x), -np.logaddexp(0.0, x)                                    # log p, log(1 - p), no saturation
            ll = det.sum(1)[:, None, None] * lp[None] + nondet.sum(1)[:, None, None] * lq[None]       # (N, n_alpha, gh): visits share b_i
            Edet = (np.exp(ll) * w[None, None, :]).sum(2)                                             # (N, n_alpha)
            Li = Epsi[:, None, None] * Edet.T[None] + (1.0 - Epsi)[:, None, None] * none[None, None, :]   # (n_beta, n_alpha, N)
        else:
            p = _sigmoid(A[:, None] + sd * t[None, :])
            Ep = (p * w[None, :]).sum(1)                                                              # (n_alpha,): E_c p(alpha0 + c)
            Eq = (_sigmoid(-(A[:, None] + sd * t[None, :])) * w[None, :]).sum(1)                      # E_c (1 - p): no cancellation
            lEp, lEq = np.log(Ep), np.log(Eq)
            ll = det.sum(1)[:, None] * lEp[None, :] + nondet.sum(1)[:, None] * lEq[None, :]           # (N, n_alpha)
            psi = _sigmoid(B)
            Li = psi[:, None, None] * np.exp(ll).T[None] + (1.0 - psi)[:, None, None] * none[None, None, :]
        lprior_u = -0.5 * (sd / prior_sd_scale) ** 2 + u                                              # HalfNormal(scale) on the log scale
        logpost[:, :, iu] = np.log(Li).sum(2) - 0.5 * B[:, None] ** 2 - 0.5 * A[None, :] ** 2 + lprior_u
    wgt = np.exp(logpost - logpost.max())
    total = wgt.sum()
    for axis in range(2):                                                 # beta0, alpha0: the mass has died out on the faces
        assert (np.take(wgt, 0, axis=axis).sum() + np.take(wgt, -1, axis=axis).sum()) / total < 1e-7, axis
    assert wgt[:, :, -1].sum() / total < 1e-7                             # u's upper face; the lower one is the prior's e^u tail, added below
    dens = wgt.sum((0, 1))
    du = U[1] - U[0]
    # below the box the likelihood no longer depends on u (sd < e^-9 changes nothing) and the density falls as e^u: tail mass = dens[0] * 1
    tail = dens[0] * 1.0 / du
    norm = (np.trapezoid(dens, U) / du + tail)
    dens = dens / (norm * du)
    cdf = tail / norm + np.concatenate([[0.0], np.cumsum(0.5 * (dens[1:] + dens[:-1]) * du)])
    return U, dens, cdf


def cdf_at(U, cdf, x):
    return np.interp(x, U, cdf)
