"""ctypes front-end of ``occu_oracle.c`` + two pure-NumPy restatements (TEST INFRASTRUCTURE).

* ``OracleData`` / ``nuts_run``: the C restatement (float64) of the marginalised occupancy
  log-density (biolith/models/occu.py:136-242) and of NumPyro's NUTS.
* ``literal_log_joint``: a line-by-line NumPy statement of the *generative* model with the
  latent z summed by brute force -- used to cross-check the closed form in the C file.
* ``effective_sample_size`` / ``split_gelman_rubin``: restatement of numpyro.diagnostics
  (what biolith/evaluation/diagnostics.py:23 calls).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LOCK = threading.Lock()

TINY_F32 = float(np.finfo(np.float32).tiny)
EPS_F32 = float(np.finfo(np.float32).eps)


def build(force: bool = False) -> str:
    """Compile ``liboccu_oracle.so`` with gcc (a few hundred ms).  ``OCCU_ORACLE_FLAVOR=native`` (set by bench.py's
    cpu_baseline leg only) selects the -O3 -march=native build, rebuilt on the host that runs it."""
    if os.environ.get("OCCU_ORACLE_FLAVOR") == "native":
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "native"], check=True)
        return os.path.join(_HERE, "_native", "liboccu_oracle_native.so")
    if os.environ.get("OCCU_ORACLE_FLAVOR") == "asan":   # tests/test_oracle_sanitize.py: ASan + UBSan build, child process only
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "liboccu_oracle_asan.so"], check=True)
        return os.path.join(_HERE, "liboccu_oracle_asan.so")
    so = os.path.join(_HERE, "liboccu_oracle.so")
    src = os.path.join(_HERE, "occu_oracle.c")
    if force or not os.path.exists(so) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)
    ):
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "liboccu_oracle.so"], check=True)
    return so


def lib():
    global _LIB
    with _LOCK:
        if _LIB is None:
            L = C.CDLL(build())
            dp = C.POINTER(C.c_double)
            L.orc_data_create.restype = C.c_void_p
            L.orc_data_create.argtypes = [C.c_int] * 5 + [dp, dp, dp] + [C.c_double] * 4
            L.orc_data_destroy.argtypes = [C.c_void_p]
            L.orc_data_dim.argtypes = [C.c_void_p]
            L.orc_data_set_model.argtypes = [C.c_void_p, C.c_int, C.c_int]
            L.orc_data_set_rn_cut.argtypes = [C.c_void_p, C.c_int]
            L.orc_data_set_fp.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
            L.orc_data_set_cop.argtypes = [C.c_void_p, dp, dp, C.c_int, C.c_double]
            L.orc_data_set_nmix.argtypes = [C.c_void_p, dp, C.c_int]
            L.orc_data_set_re.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
            L.orc_data_set_re_fp.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
            L.orc_data_set_nmix_re.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
            L.orc_data_set_rn_re.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
            L.orc_data_set_cop_re.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
            L.orc_data_set_rn_fp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
            L.orc_data_set_prior_family.argtypes = [C.c_void_p, C.c_int, C.c_int]
            L.orc_data_set_cs.argtypes = [C.c_void_p, dp, dp, dp]
            L.orc_data_set_dyn.argtypes = [C.c_void_p]
            L.orc_data_set_species.argtypes = [C.c_void_p, C.c_int, dp, C.POINTER(C.c_ubyte)]
            L.orc_potential_grad.restype = C.c_double
            L.orc_potential_grad.argtypes = [C.c_void_p, dp, dp]
            L.orc_rng_streams.argtypes = [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
            L.orc_rng_streams_strided.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
            L.orc_rng_next.restype = C.c_uint32
            L.orc_rng_next.argtypes = [C.POINTER(C.c_uint32)]
            L.orc_rng_jump.argtypes = [C.POINTER(C.c_uint32)]
            L.orc_rng_uniform.restype = C.c_double
            L.orc_rng_uniform.argtypes = [C.POINTER(C.c_uint32)]
            L.orc_rng_normal.restype = C.c_double
            L.orc_rng_normal.argtypes = [C.POINTER(C.c_uint32)]
            L.orc_adaptation_schedule.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
            L.orc_nuts_run.restype = C.c_int
            L.orc_nuts_run.argtypes = [
                C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double,
                dp, dp, C.POINTER(C.c_int), dp, C.POINTER(C.c_ubyte), dp, dp, dp,
                C.POINTER(C.c_longlong), dp, C.POINTER(C.c_int), dp,
            ]
            _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _as_f32_f64(a):
    """The reference feeds float32 arrays (utils/data.py:135-140); keep those exact values."""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).astype(np.float64))


class OracleData:
    """Prepared dataset for ONE species: site_covs (N,Ks), obs_covs (N,T,J,Ko), obs (N,T,J)."""

    def __init__(self, site_covs, obs_covs, obs, prior_beta=(0.0, 1.0), prior_alpha=(0.0, 1.0), model="occu",
                 max_abundance=100, fp_mode="constant", prior_fp=(2.0, 5.0), session_duration=None, prior_fp_rate=1.0,
                 site_random_effects=False, obs_random_effects=False, prior_site_re_sd=1.0, prior_obs_re_sd=1.0,
                 prior_family=("normal", "normal"), prior_mu=((0.0, 10.0), (0.0, 10.0)), prior_sigma=((5.0, 1.0), (5.0, 1.0)),
                 re_fp_mode=None):
        X = _as_f32_f64(site_covs)
        W = _as_f32_f64(obs_covs)
        Y = _as_f32_f64(obs)
        Y_all = None
        if Y.ndim == 4:
            if Y.shape[0] > 1:   # several species under ONE chain (occu.py:182-186): occu with or without false positives
                assert model in ("occu", "occu_fp", "occu_re"), "joint species: occu / occu_fp / occu_re"
                Y_all = np.ascontiguousarray(Y)
            Y = np.ascontiguousarray(Y[0])
        assert X.ndim == 2 and W.ndim == 4 and Y.ndim == 3
        N, Ks = X.shape
        _, T, J, Ko = W.shape
        assert W.shape[0] == N and Y.shape == (N, T, J)
        self.N, self.T, self.J, self.Ks, self.Ko = N, T, J, Ks, Ko
        self.D = Ks + Ko + 2
        self._h = lib().orc_data_create(
            N, T, J, Ks, Ko, _dp(X), _dp(W), _dp(Y),
            float(prior_beta[0]), float(prior_beta[1]), float(prior_alpha[0]), float(prior_alpha[1]),
        )
        self.X, self.W, self.Y = X, W, Y
        self.prior_beta, self.prior_alpha = tuple(prior_beta), tuple(prior_alpha)
        self.model, self.max_abundance = model, int(max_abundance)
        assert model in ("occu", "occu_rn", "occu_fp", "occu_cop", "nmixture", "occu_re", "occu_cs", "occu_dyn")
        lib().orc_data_set_model(self._h, 1 if model == "occu_rn" else 0, int(max_abundance))
        if model == "occu_rn" and re_fp_mode is not None:   # occu_rn.py:133-138, 214-221 (+ the random effects): [beta, alpha, phi, log sds, effects]
            assert re_fp_mode == "constant"
            lib().orc_data_set_rn_fp(self._h, int(bool(site_random_effects)), int(bool(obs_random_effects)),
                                     float(prior_site_re_sd), float(prior_obs_re_sd), float(prior_fp[0]), float(prior_fp[1]))
            self.D = int(lib().orc_data_dim(self._h))
        elif model == "occu_rn" and (site_random_effects or obs_random_effects):   # occu_rn.py:151-154, 172-184, 199-212
            lib().orc_data_set_rn_re(self._h, int(bool(site_random_effects)), int(bool(obs_random_effects)),
                                     float(prior_site_re_sd), float(prior_obs_re_sd))
            self.D = int(lib().orc_data_dim(self._h))
        if model == "occu_fp":  # theta gains phi = logit(false-positive rate) as its last coordinate
            assert fp_mode in ("constant", "unoccupied")
            lib().orc_data_set_fp(self._h, 1 if fp_mode == "constant" else 2, float(prior_fp[0]), float(prior_fp[1]))
            self.D += 1
        if model == "nmixture":  # counts, N enumerated over 0..max_abundance with raw (un-renormalised) Poisson weights
            lib().orc_data_set_nmix(self._h, _dp(Y), int(max_abundance))
            if site_random_effects or obs_random_effects:   # nmixture.py:139-141, 166, 199: the layout of occu's random effects
                lib().orc_data_set_nmix_re(self._h, int(bool(site_random_effects)), int(bool(obs_random_effects)),
                                           float(prior_site_re_sd), float(prior_obs_re_sd))
                self.D = int(lib().orc_data_dim(self._h))
        if model == "occu_cop":  # counts + exposure; optional false-positive rate as trailing phi = log(rate)
            assert fp_mode in (None, "constant", "unoccupied")
            Dur = _as_f32_f64(session_duration)
            assert Dur.shape == (N, T, J)
            self.Dur = Dur
            lib().orc_data_set_cop(self._h, _dp(Y), _dp(Dur), {None: 0, "constant": 1, "unoccupied": 2}[fp_mode],
                                   float(prior_fp_rate))
            self.D += 1 if fp_mode else 0
            if site_random_effects or obs_random_effects:   # occu_cop.py:183-186, 204-210, 229-243 (a rate stays: [beta, alpha, phi, log sds, effects])
                lib().orc_data_set_cop_re(self._h, int(bool(site_random_effects)), int(bool(obs_random_effects)),
                                          float(prior_site_re_sd), float(prior_obs_re_sd))
                self.D = int(lib().orc_data_dim(self._h))
        self.fp_mode, self.prior_fp, self.prior_fp_rate = fp_mode, tuple(prior_fp), float(prior_fp_rate)
        if model == "occu_cs":
            # theta = [beta, alpha, mu0, log(mu1 - mu0), log sigma0, log sigma1]; obs holds the scores
            pm = np.ascontiguousarray(np.asarray(prior_mu, dtype=np.float64).reshape(4))
            ps = np.ascontiguousarray(np.asarray(prior_sigma, dtype=np.float64).reshape(4))
            lib().orc_data_set_cs(self._h, _dp(Y), _dp(pm), _dp(ps))
            self.D += 4
            self.prior_mu, self.prior_sigma = pm.reshape(2, 2), ps.reshape(2, 2)
        if model == "occu_re":
            # theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_occ[N], site_re_det[N]), (obs_re[N][T][J])]
            assert site_random_effects or obs_random_effects
            lib().orc_data_set_re(self._h, int(bool(site_random_effects)), int(bool(obs_random_effects)),
                                  float(prior_site_re_sd), float(prior_obs_re_sd))
            if re_fp_mode is not None:   # random effects together with a false-positive rate: theta = [beta, alpha, phi, log sds, effects]
                assert re_fp_mode in ("constant", "unoccupied") and Y_all is None
                lib().orc_data_set_re_fp(self._h, 1 if re_fp_mode == "constant" else 2, float(prior_fp[0]), float(prior_fp[1]))
            self.D = int(lib().orc_data_dim(self._h))
        if model == "occu_dyn":
            # builder-defined dynamic occupancy (no reference counterpart): theta = [b_psi | b_gamma | b_eps (Ks+1 each) | alpha (Ko+1)]
            lib().orc_data_set_dyn(self._h)
            self.D = 3 * (Ks + 1) + Ko + 1
        self.n_species = 1
        if Y_all is not None:
            # theta = [species 0: beta, alpha | species 1: ... | (phi)]; with random effects (set before the species, so that the
            # dimension counts them): [... | log sds | site_re_occ [S][N] | site_re_det [S][N] | obs_re [S][N][T][J]]
            cov_nan = (np.isnan(np.asarray(obs_covs, dtype=np.float64)).any(-1)
                       | np.isnan(np.asarray(site_covs, dtype=np.float64)).any(-1)[:, None, None]).astype(np.uint8)
            cov_nan = np.ascontiguousarray(cov_nan)
            lib().orc_data_set_species(self._h, int(Y_all.shape[0]), _dp(Y_all), cov_nan.ctypes.data_as(C.POINTER(C.c_ubyte)))
            self.n_species = int(Y_all.shape[0])
            self.D = int(lib().orc_data_dim(self._h))
            self.Y_all = Y_all
        # prior family of beta / alpha: "normal" (occu.py:28-29) or "laplace" (utils/grid_search.py:366-371), same (loc, scale)
        assert all(f in ("normal", "laplace") for f in prior_family)
        self.prior_family = tuple(prior_family)
        lib().orc_data_set_prior_family(self._h, int(prior_family[0] == "laplace"), int(prior_family[1] == "laplace"))
        self.site_re, self.obs_re = bool(site_random_effects), bool(obs_random_effects)
        self.prior_site_re_sd, self.prior_obs_re_sd = float(prior_site_re_sd), float(prior_obs_re_sd)

    def __del__(self):
        try:
            if self._h:
                lib().orc_data_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_rn_cut(self, on: bool):
        """occu_rn: stop the sums over n where the terms have died out (default) or sum every n <= max_abundance (tests of the stop)."""
        lib().orc_data_set_rn_cut(self._h, int(bool(on)))

    def potential_grad(self, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        if th.ndim == 1:
            g = np.empty(self.D)
            u = lib().orc_potential_grad(self._h, _dp(th), _dp(g))
            return u, g
        U = np.empty(th.shape[0])
        G = np.empty_like(th)
        for b in range(th.shape[0]):
            U[b] = lib().orc_potential_grad(self._h, _dp(th[b]), _dp(G[b]))
        return U, G


def rng_streams(seed: int, chain: int, nstreams: int = 64) -> np.ndarray:
    out = np.zeros((nstreams, 4), dtype=np.uint32)
    lib().orc_rng_streams(seed, chain, nstreams, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def adaptation_schedule(num_warmup: int):
    s = (C.c_int * 40)()
    e = (C.c_int * 40)()
    n = lib().orc_adaptation_schedule(num_warmup, s, e)
    return [(s[i], e[i]) for i in range(n)]


def _run_chain(data, num_warmup, num_samples, seed, chain, max_tree_depth, target_accept, init, trace):
    D = data.D
    draws = np.empty((num_samples, D))
    steps = np.empty(num_samples, dtype=np.int32)
    acc = np.empty(num_samples)
    div = np.empty(num_samples, dtype=np.uint8)
    pot = np.empty(num_samples)
    eps = C.c_double()
    minv = np.empty(D)
    nleap = (C.c_longlong * 2)()
    tot = num_warmup + num_samples
    tr_t = np.empty((tot, D)) if trace else None
    tr_s = np.empty(tot, dtype=np.int32) if trace else None
    tr_e = np.empty(tot) if trace else None
    init_a = np.ascontiguousarray(init, dtype=np.float64) if init is not None else None
    rc = lib().orc_nuts_run(
        data._h, num_warmup, num_samples, seed, chain, max_tree_depth, target_accept,
        _dp(init_a), _dp(draws), steps.ctypes.data_as(C.POINTER(C.c_int)), _dp(acc),
        div.ctypes.data_as(C.POINTER(C.c_ubyte)), _dp(pot), C.byref(eps), _dp(minv), nleap,
        _dp(tr_t), tr_s.ctypes.data_as(C.POINTER(C.c_int)) if trace else None, _dp(tr_e),
    )
    if rc != 0:
        raise RuntimeError(f"orc_nuts_run failed rc={rc}")
    out = dict(draws=draws, num_steps=steps, accept_prob=acc, diverging=div, potential=pot,
               step_size=eps.value, inv_mass=minv, n_leapfrog=(nleap[0], nleap[1]))
    if trace:
        out.update(trace_theta=tr_t, trace_steps=tr_s, trace_eps=tr_e)
    return out


def nuts_run(data: OracleData, num_warmup=1000, num_samples=1000, num_chains=1, seed=0,
             chain_offset=0, max_tree_depth=10, target_accept=0.8, init=None, trace=False,
             threads=None):
    """Run ``num_chains`` oracle chains (one OS thread each; ctypes drops the GIL)."""
    threads = threads or min(num_chains, os.cpu_count() or 1)
    lib()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        futs = [
            ex.submit(_run_chain, data, num_warmup, num_samples, seed, chain_offset + c,
                      max_tree_depth, target_accept, None if init is None else init[c], trace)
            for c in range(num_chains)
        ]
        res = [f.result() for f in futs]
    out = {k: np.stack([r[k] for r in res]) for k in res[0] if k != "n_leapfrog"}
    out["n_leapfrog"] = np.array([r["n_leapfrog"] for r in res])
    out["threads"] = threads
    return out


# --------------------------------------------------------------------------------------------
# Literal NumPy statement of the generative model (biolith/models/occu.py:136-242), z summed by
# brute force.  Independent of the closed form in occu_oracle.c.
# --------------------------------------------------------------------------------------------
def _bernoulli_logpmf_clamped(p, y):
    """numpyro BernoulliProbs.log_prob with clamp_probs (float32 tiny / eps) [UPSTREAM]."""
    pc = np.clip(p, TINY_F32, 1.0 - EPS_F32)
    return y * np.log(pc) + (1.0 - y) * np.log1p(-pc)


def literal_log_joint(theta, site_covs, obs_covs, obs, prior_beta=(0.0, 1.0), prior_alpha=(0.0, 1.0),
                      clamp_z1=False, prob_fp_constant=0.0, prob_fp_unoccupied=0.0, prior_family=("normal", "normal")):
    """log p(theta, y) for one species; obs (N,T,J).  ``clamp_z1`` applies numpyro's prob clamp
    in the z=1 branch too (the C oracle does not; they differ only for |nu| > ~15.9).
    ``prob_fp_*``: the false-positive rates of occu.py:146-157 as given numbers (their own prior is
    added by :func:`literal_log_joint_fp`)."""
    X = _as_f32_f64(site_covs)
    W = _as_f32_f64(obs_covs)
    Y = _as_f32_f64(obs)
    if Y.ndim == 4:
        Y = Y[0]
    N, Ks = X.shape
    Ko = W.shape[-1]
    theta = np.asarray(theta, dtype=np.float64)
    beta, alpha = theta[: Ks + 1], theta[Ks + 1:]
    # occu.py:136-142
    obs_mask = np.isnan(W).any(-1) | np.isnan(X).any(-1)[:, None, None]
    Y = np.where(obs_mask, np.nan, Y)
    W = np.nan_to_num(W)
    X = np.nan_to_num(X)
    occ_linear = beta[0] + X @ beta[1:]                        # occu.py:198-202, linear.py:59-66
    psi = 1.0 / (1.0 + np.exp(-occ_linear))                    # occu.py:207
    det_linear = alpha[0] + np.tensordot(W, alpha[1:], axes=([3], [0]))
    p = 1.0 / (1.0 + np.exp(-det_linear))                      # occu.py:221-228
    finite = np.isfinite(Y)                                    # modeling.py:15-17
    y0 = np.where(finite, Y, 0.0)
    per_z = []
    for z in (0.0, 1.0):
        p_fp = 1.0 - (1.0 - z * p) * (1.0 - prob_fp_constant) * (1.0 - (1.0 - z) * prob_fp_unoccupied)   # occu.py:229-235
        if z == 1.0 and not clamp_z1 and prob_fp_constant == 0.0:
            ly = y0 * (-np.logaddexp(0.0, -det_linear)) + (1.0 - y0) * (-np.logaddexp(0.0, det_linear))
        else:
            ly = _bernoulli_logpmf_clamped(p_fp, y0)
        ly = np.where(finite, ly, 0.0).sum(axis=2)             # over replicates -> (N,T)
        lz = _bernoulli_logpmf_clamped(psi, z)[:, None]        # z ~ Bernoulli(psi) per period
        per_z.append(lz + ly)
    ll = np.logaddexp(per_z[0], per_z[1]).sum()

    def prior_logpdf(v, loc, scale, family):   # dist.Normal / dist.Laplace .log_prob
        if family == "laplace":
            return (-np.abs(v - loc) / scale - np.log(2.0 * scale)).sum()
        return (-0.5 * ((v - loc) / scale) ** 2 - np.log(scale) - 0.5 * np.log(2 * np.pi)).sum()

    return ll + prior_logpdf(beta, *prior_beta, prior_family[0]) + prior_logpdf(alpha, *prior_alpha, prior_family[1])


def literal_log_joint_re(theta, site_covs, obs_covs, obs, site_random_effects=True, obs_random_effects=False,
                         prior_site_re_sd=1.0, prior_obs_re_sd=1.0, prior_beta=(0.0, 1.0), prior_alpha=(0.0, 1.0),
                         re_fp_mode=None, prior_fp=(2.0, 5.0)):
    """log density of occu with random effects (biolith/models/occu.py:170-173, 191-196, 215-218) in NumPyro's
    unconstrained space, z summed by brute force.  theta = [beta, alpha, (log site_re_sd), (log obs_re_sd),
    (site_re_occ[N], site_re_det[N]), (obs_re[N][T][J])]; a HalfNormal site lives on the log scale (+ log-Jacobian)."""
    X, W, Y = (_as_f32_f64(a) for a in (site_covs, obs_covs, obs))
    if Y.ndim == 4:
        Y = Y[0]
    N, Ks = X.shape
    _, T, J, Ko = W.shape
    theta = np.asarray(theta, dtype=np.float64)
    at = Ks + Ko + 2
    beta, alpha = theta[: Ks + 1], theta[Ks + 1: at]
    lp = 0.0
    f_c = f_u = 0.0
    if re_fp_mode is not None:   # with a false-positive rate (occu.py:146-157): theta = [beta, alpha, phi = logit(rate), ...]
        from scipy.special import betaln

        f = 1.0 / (1.0 + np.exp(-theta[at]))
        at += 1
        f_c, f_u = (f, 0.0) if re_fp_mode == "constant" else (0.0, f)
        lp += (prior_fp[0] - 1.0) * np.log(f) + (prior_fp[1] - 1.0) * np.log1p(-f) - betaln(*prior_fp) + np.log(f) + np.log1p(-f)

    def half_normal_on_log_scale(phi, scale):      # dist.HalfNormal(scale).log_prob(sd) + log|d sd / d phi|
        sd = np.exp(phi)
        return 0.5 * np.log(2.0 / np.pi) - np.log(scale) - 0.5 * (sd / scale) ** 2 + phi

    def normal_logpdf(v, loc, scale):
        return (-0.5 * ((v - loc) / scale) ** 2 - np.log(scale) - 0.5 * np.log(2 * np.pi)).sum()

    sd_site = sd_obs = None
    if site_random_effects:
        lp += half_normal_on_log_scale(theta[at], prior_site_re_sd)
        sd_site = np.exp(theta[at]); at += 1
    if obs_random_effects:
        lp += half_normal_on_log_scale(theta[at], prior_obs_re_sd)
        sd_obs = np.exp(theta[at]); at += 1
    re_occ = re_det = np.zeros(N)
    obs_re = np.zeros((N, T, J))
    if site_random_effects:
        re_occ, re_det = theta[at: at + N], theta[at + N: at + 2 * N]
        at += 2 * N
        lp += normal_logpdf(re_occ, 0.0, sd_site) + normal_logpdf(re_det, 0.0, sd_site)
    if obs_random_effects:
        obs_re = theta[at: at + N * T * J].reshape(N, T, J)
        at += N * T * J
        lp += normal_logpdf(obs_re, 0.0, sd_obs)
    assert at == theta.size
    obs_mask = np.isnan(W).any(-1) | np.isnan(X).any(-1)[:, None, None]       # occu.py:136-142
    Y = np.where(obs_mask, np.nan, Y)
    W, X = np.nan_to_num(W), np.nan_to_num(X)
    occ_linear = beta[0] + X @ beta[1:] + re_occ                               # occu.py:198-202
    psi = 1.0 / (1.0 + np.exp(-occ_linear))
    det_linear = alpha[0] + np.tensordot(W, alpha[1:], axes=([3], [0])) + re_det[:, None, None] + obs_re   # occu.py:221-228
    finite = np.isfinite(Y)
    y0 = np.where(finite, Y, 0.0)
    per_z = []
    for z in (0.0, 1.0):
        if re_fp_mode is not None:   # occu.py:229-241: 1 - (1 - z p)(1 - f_c)(1 - (1 - z) f_u), numpyro's clamp in both branches
            pdet = 1.0 / (1.0 + np.exp(-det_linear))
            ly = _bernoulli_logpmf_clamped(1.0 - (1.0 - z * pdet) * (1.0 - f_c) * (1.0 - (1.0 - z) * f_u), y0)
        elif z == 1.0:   # exact log-sigmoid in the z = 1 branch, as literal_log_joint(clamp_z1=False)
            ly = y0 * (-np.logaddexp(0.0, -det_linear)) + (1.0 - y0) * (-np.logaddexp(0.0, det_linear))
        else:
            ly = _bernoulli_logpmf_clamped(np.zeros_like(det_linear), y0)
        ly = np.where(finite, ly, 0.0).sum(axis=2)
        per_z.append(_bernoulli_logpmf_clamped(psi, z)[:, None] + ly)
    ll = np.logaddexp(per_z[0], per_z[1]).sum()
    return ll + lp + normal_logpdf(beta, *prior_beta) + normal_logpdf(alpha, *prior_alpha)


def literal_log_joint_cs(theta, site_covs, obs_covs, scores, prior_mu=((0.0, 10.0), (0.0, 10.0)), prior_sigma=((5.0, 1.0), (5.0, 1.0)),
                         prior_beta=(0.0, 1.0), prior_alpha=(0.0, 1.0)):
    """log density of the continuous-score model (biolith/models/occu_cs.py:120-232) in NumPyro's unconstrained space, z
    and f summed by brute force over an explicit (z, f) axis.  theta = [beta, alpha, mu0, x1, log sigma0, log sigma1]."""
    from scipy.special import erfc, gammaln, logsumexp

    X, W, S = (_as_f32_f64(a) for a in (site_covs, obs_covs, scores))
    if S.ndim == 4:
        S = S[0]
    Ks, Ko = X.shape[1], W.shape[-1]
    theta = np.asarray(theta, dtype=np.float64)
    beta, alpha = theta[: Ks + 1], theta[Ks + 1: Ks + Ko + 2]
    mu0, x1, ls0, ls1 = theta[Ks + Ko + 2:]
    mu1, sg = mu0 + np.exp(x1), np.exp([ls0, ls1])
    mask = np.isnan(W).any(-1) | np.isnan(X).any(-1)[:, None, None]            # occu_cs.py:120-126
    S = np.where(mask, np.nan, S)
    W, X = np.nan_to_num(W), np.nan_to_num(X)
    psi = 1.0 / (1.0 + np.exp(-(beta[0] + X @ beta[1:])))
    p = 1.0 / (1.0 + np.exp(-(alpha[0] + np.tensordot(W, alpha[1:], axes=([3], [0])))))
    finite = np.isfinite(S)
    s0 = np.where(finite, S, 0.0)
    mus = np.array([mu0, mu1])
    lphi = -0.5 * ((s0[..., None] - mus) / sg) ** 2 - np.log(sg) - 0.5 * np.log(2 * np.pi)       # (N, T, J, f)
    per_z = []
    for z in (0.0, 1.0):
        pf = np.stack([_bernoulli_logpmf_clamped(z * p, 0.0), _bernoulli_logpmf_clamped(z * p, 1.0)], axis=-1)   # log P(f | z)
        lj = np.where(finite, logsumexp(pf + lphi, axis=-1), 0.0).sum(axis=2)                    # (N, T)
        lz = -np.logaddexp(0.0, -(beta[0] + X @ beta[1:])) if z == 1.0 else -np.logaddexp(0.0, beta[0] + X @ beta[1:])
        per_z.append(lz[:, None] + lj)
    ll = np.logaddexp(per_z[0], per_z[1]).sum()

    def normal_logpdf(v, loc, scale):
        return (-0.5 * ((v - loc) / scale) ** 2 - np.log(scale) - 0.5 * np.log(2 * np.pi)).sum()

    (l0, c0), (l1, c1) = prior_mu
    lp = normal_logpdf(mu0, l0, c0)
    lp += normal_logpdf(mu1, l1, c1) - np.log(0.5 * erfc((mu0 - l1) / c1 / np.sqrt(2.0))) + x1       # truncated below at mu0
    for (a, b), ls in zip(prior_sigma, (ls0, ls1)):
        lp += a * np.log(b) - gammaln(a) + (a - 1.0) * ls - b * np.exp(ls) + ls                      # Gamma + log-Jacobian
    return ll + lp + normal_logpdf(beta, *prior_beta) + normal_logpdf(alpha, *prior_alpha)


def literal_log_joint_fp(theta, site_covs, obs_covs, obs, fp_mode="constant", prior_fp=(2.0, 5.0),
                         prior_beta=(0.0, 1.0), prior_alpha=(0.0, 1.0), clamp_z1=False):
    """log density of the false-positive model in NumPyro's unconstrained space: theta = [beta, alpha, phi],
    rate = sigmoid(phi) ~ Beta(a, b) (occu.py:32-33,146-157), plus the log-Jacobian of the sigmoid."""
    from scipy.special import betaln

    theta = np.asarray(theta, dtype=np.float64)
    phi = theta[-1]
    f = 1.0 / (1.0 + np.exp(-phi))
    kw = {"prob_fp_constant": f} if fp_mode == "constant" else {"prob_fp_unoccupied": f}
    a, b = prior_fp
    beta_logpdf = (a - 1.0) * np.log(f) + (b - 1.0) * np.log1p(-f) - betaln(a, b)
    return (literal_log_joint(theta[:-1], site_covs, obs_covs, obs, prior_beta, prior_alpha, clamp_z1=clamp_z1, **kw)
            + beta_logpdf + np.log(f) + np.log1p(-f))


def literal_log_joint_cop(theta, site_covs, obs_covs, obs, session_duration, fp_mode=None, prior_fp_rate=1.0,
                          prior_beta=(0.0, 1.0), prior_alpha=(0.0, 1.0),
                          site_random_effects=False, obs_random_effects=False, prior_site_re_sd=1.0, prior_obs_re_sd=1.0):
    """log density of the count occupancy model (biolith/models/occu_cop.py:150-255) in NumPyro's unconstrained
    space, z summed by brute force: theta = [beta, alpha (, phi = log rate_fp)]."""
    from scipy.special import gammaln, xlogy

    X, W, Y, Dur = (_as_f32_f64(a) for a in (site_covs, obs_covs, obs, session_duration))
    if Y.ndim == 4:
        Y = Y[0]
    Ks, Ko = X.shape[1], W.shape[-1]
    theta = np.asarray(theta, dtype=np.float64)
    beta, alpha = theta[: Ks + 1], theta[Ks + 1: Ks + Ko + 2]
    i_fp = Ks + Ko + 2 if (site_random_effects or obs_random_effects) else -1   # (with random effects the rate sits right behind the coefficients)
    f = np.exp(theta[i_fp]) if fp_mode else 0.0
    f_c, f_u = (f if fp_mode == "constant" else 0.0), (f if fp_mode == "unoccupied" else 0.0)
    # random effects (occu_cop.py:183-186, 204-210, 229-243):
    # theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_occ[N], site_re_det[N]), (obs_re[N][T][J])]
    N_, T_, J_ = Y.shape
    at, lp_re = Ks + Ko + 2 + (1 if fp_mode else 0), 0.0
    re_occ = re_det = np.zeros(N_)
    obs_re = np.zeros((N_, T_, J_))
    if site_random_effects or obs_random_effects:

        def half_normal_on_log_scale(phi, scale):
            return 0.5 * np.log(2.0 / np.pi) - np.log(scale) - 0.5 * (np.exp(phi) / scale) ** 2 + phi

        def normal0(v, sd):
            return (-0.5 * (v / sd) ** 2 - np.log(sd) - 0.5 * np.log(2 * np.pi)).sum()

        sd_s = sd_o = None
        if site_random_effects:
            lp_re += half_normal_on_log_scale(theta[at], prior_site_re_sd); sd_s = np.exp(theta[at]); at += 1
        if obs_random_effects:
            lp_re += half_normal_on_log_scale(theta[at], prior_obs_re_sd); sd_o = np.exp(theta[at]); at += 1
        if site_random_effects:
            re_occ, re_det = theta[at: at + N_], theta[at + N_: at + 2 * N_]; at += 2 * N_
            lp_re += normal0(re_occ, sd_s) + normal0(re_det, sd_s)
        if obs_random_effects:
            obs_re = theta[at: at + N_ * T_ * J_].reshape(N_, T_, J_); at += N_ * T_ * J_
            lp_re += normal0(obs_re, sd_o)
        assert at == theta.size
    obs_mask = np.isnan(W).any(-1) | np.isnan(X).any(-1)[:, None, None]     # occu_cop.py:150-156
    Y = np.where(obs_mask, np.nan, Y)
    W, X = np.nan_to_num(W), np.nan_to_num(X)
    psi = 1.0 / (1.0 + np.exp(-(beta[0] + X @ beta[1:] + re_occ)))          # occu_cop.py:222-227
    rate_detection = np.exp(alpha[0] + np.tensordot(W, alpha[1:], axes=([3], [0])) + re_det[:, None, None] + obs_re)   # occu_cop.py:236-243
    finite = np.isfinite(Y)
    y0 = np.where(finite, Y, 0.0)
    per_z = []
    for z in (0.0, 1.0):
        l_det = z * rate_detection + (1.0 - z) * f_u + f_c                  # occu_cop.py:244-248
        rate = Dur * l_det
        with np.errstate(divide="ignore", invalid="ignore"):
            ly = xlogy(y0, rate) - gammaln(y0 + 1.0) - rate                 # numpyro Poisson.log_prob
        ly = np.where(finite, ly, 0.0).sum(axis=2)
        per_z.append(_bernoulli_logpmf_clamped(psi, z)[:, None] + ly)
    ll = np.logaddexp(per_z[0], per_z[1]).sum()

    def normal_logpdf(v, loc, scale):
        return (-0.5 * ((v - loc) / scale) ** 2 - np.log(scale) - 0.5 * np.log(2 * np.pi)).sum()

    out = ll + lp_re + normal_logpdf(beta, *prior_beta) + normal_logpdf(alpha, *prior_alpha)
    if fp_mode:
        out += np.log(prior_fp_rate) - prior_fp_rate * f + theta[i_fp]     # Exponential log-pdf + log|d f / d phi|
    return out


def literal_log_joint_nmix(theta, site_covs, obs_covs, obs, max_abundance=100, prior_beta=(0.0, 1.0), prior_alpha=(0.0, 1.0),
                           site_random_effects=False, obs_random_effects=False, prior_site_re_sd=1.0, prior_obs_re_sd=1.0):
    """log density of the N-mixture model (biolith/models/nmixture.py:150-220), N summed by brute force with the
    model's own ingredients: Poisson logits masked below the largest count, the ``N_i_trunc_norm`` factor and the
    (normalising) Categorical, Binomial log-pmf as numpyro states it."""
    from scipy.special import gammaln, logsumexp, xlog1py, xlogy

    X, W, Y = (_as_f32_f64(a) for a in (site_covs, obs_covs, obs))
    if Y.ndim == 4:
        Y = Y[0]
    Ks, Ko = X.shape[1], W.shape[-1]
    theta = np.asarray(theta, dtype=np.float64)
    beta, alpha = theta[: Ks + 1], theta[Ks + 1: Ks + Ko + 2]
    # random effects (nmixture.py:139-141, 166-172, 199-214): theta = [beta, alpha, (log site_re_sd), (log obs_re_sd),
    # (site_re_abu[N], site_re_det[N]), (obs_re[N][T][J])]; a HalfNormal site lives on the log scale (+ log-Jacobian)
    N_, T_, J_ = Y.shape
    at, lp_re = Ks + Ko + 2, 0.0
    re_abu = re_det = np.zeros(N_)
    obs_re = np.zeros((N_, T_, J_))

    def half_normal_on_log_scale(phi, scale):
        return 0.5 * np.log(2.0 / np.pi) - np.log(scale) - 0.5 * (np.exp(phi) / scale) ** 2 + phi

    def normal0(v, sd):
        return (-0.5 * (v / sd) ** 2 - np.log(sd) - 0.5 * np.log(2 * np.pi)).sum()

    sd_s = sd_o = None
    if site_random_effects:
        lp_re += half_normal_on_log_scale(theta[at], prior_site_re_sd); sd_s = np.exp(theta[at]); at += 1
    if obs_random_effects:
        lp_re += half_normal_on_log_scale(theta[at], prior_obs_re_sd); sd_o = np.exp(theta[at]); at += 1
    if site_random_effects:
        re_abu, re_det = theta[at: at + N_], theta[at + N_: at + 2 * N_]; at += 2 * N_
        lp_re += normal0(re_abu, sd_s) + normal0(re_det, sd_s)
    if obs_random_effects:
        obs_re = theta[at: at + N_ * T_ * J_].reshape(N_, T_, J_); at += N_ * T_ * J_
        lp_re += normal0(obs_re, sd_o)
    assert at == theta.size
    obs_mask = np.isnan(W).any(-1) | np.isnan(X).any(-1)[:, None, None]     # nmixture.py:117-123
    Y = np.where(obs_mask, np.nan, Y)
    W, X = np.nan_to_num(W), np.nan_to_num(X)
    with np.errstate(invalid="ignore"):
        obs_max = np.max(np.where(np.isnan(Y), -np.inf, Y), axis=2)          # nmixture.py:150-155 (over replicates)
    min_counts = np.where(np.isfinite(obs_max), obs_max, 0).astype(int)     # (N, T)
    abundance = np.exp(beta[0] + X @ beta[1:] + re_abu)                     # (N,)
    support = np.arange(max_abundance + 1)
    logits = xlogy(support[None, :], abundance[:, None]) - gammaln(support + 1.0)[None, :] - abundance[:, None]   # Poisson.log_prob
    logits = np.broadcast_to(logits[:, None, :], min_counts.shape + (max_abundance + 1,)).copy()
    logits[support[None, None, :] < min_counts[..., None]] = -np.inf        # nmixture.py:186-190
    trunc_norm = logsumexp(logits, axis=-1)                                 # factor "N_i_trunc_norm"
    log_cat = logits - trunc_norm[..., None]                                # Categorical(logits).log_prob
    p = 1.0 / (1.0 + np.exp(-(alpha[0] + np.tensordot(W, alpha[1:], axes=([3], [0])) + re_det[:, None, None] + obs_re)))   # (N, T, J)
    finite = np.isfinite(Y)
    y0 = np.where(finite, Y, 0.0)
    n = support[None, None, None, :].astype(np.float64)                     # (1,1,1,K+1)
    yy, pp = y0[..., None], p[..., None]
    with np.errstate(invalid="ignore", divide="ignore"):
        lb = (gammaln(n + 1.0) - gammaln(yy + 1.0) - gammaln(n - yy + 1.0) + xlogy(yy, pp) + xlog1py(n - yy, -pp))
    lb = np.where(finite[..., None], lb, 0.0).sum(axis=2)                   # (N, T, K+1)
    with np.errstate(invalid="ignore"):
        per_n = np.where(np.isfinite(log_cat), log_cat + lb, -np.inf)
    ll = (trunc_norm + logsumexp(per_n, axis=-1)).sum()

    def normal_logpdf(v, loc, scale):
        return (-0.5 * ((v - loc) / scale) ** 2 - np.log(scale) - 0.5 * np.log(2 * np.pi)).sum()

    return ll + lp_re + normal_logpdf(beta, *prior_beta) + normal_logpdf(alpha, *prior_alpha)


def literal_log_joint_dyn(theta, site_covs, obs_covs, obs, prior_beta=(0.0, 1.0), prior_alpha=(0.0, 1.0)):
    """The dynamic occupancy model (builder-defined: occu_oracle.c, potential_grad_dyn) stated literally: the log joint density of
    theta and y with every latent path z_i1..z_iT summed by BRUTE FORCE over the 2^T paths (small T only).  Returns log p(theta, y)."""
    import itertools

    X, W, Y = _as_f32_f64(site_covs), _as_f32_f64(obs_covs), _as_f32_f64(obs)
    if Y.ndim == 4:
        Y = Y[0]
    N, Ks = X.shape
    _, T, J, Ko = W.shape
    B = Ks + 1
    th = np.asarray(theta, dtype=np.float64)
    bp, bg, be, al = th[:B], th[B:2 * B], th[2 * B:3 * B], th[3 * B:]
    mask = np.isfinite(Y) & ~np.isnan(W).any(-1) & ~np.isnan(X).any(-1)[:, None, None]
    Xc, Wc = np.nan_to_num(X), np.nan_to_num(W)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))  # noqa: E731
    psi, gam, eps = sig(bp[0] + Xc @ bp[1:]), sig(bg[0] + Xc @ bg[1:]), sig(be[0] + Xc @ be[1:])
    p = sig(al[0] + Wc @ al[1:])                                              # (N, T, J)
    tiny = float(np.finfo(np.float32).tiny)
    total = 0.0
    for i in range(N):
        terms = []
        for z in itertools.product((0, 1), repeat=T):
            lp = np.log(psi[i] if z[0] else 1.0 - psi[i])
            for t in range(1, T):
                pr1 = (1.0 - eps[i]) if z[t - 1] else gam[i]
                lp += np.log(pr1 if z[t] else 1.0 - pr1)
            for t in range(T):
                for j in range(J):
                    if not mask[i, t, j]:
                        continue
                    pd = p[i, t, j] if z[t] else tiny                         # P(y = 1 | z): numpyro's clamp at z = 0, as in occu
                    lp += np.log(pd) if Y[i, t, j] != 0 else np.log1p(-pd)
            terms.append(lp)
        terms = np.array(terms)
        total += terms.max() + np.log(np.exp(terms - terms.max()).sum())

    def normal_logpdf(v, loc, scale):
        return -0.5 * ((v - loc) / scale) ** 2 - np.log(scale) - 0.5 * np.log(2 * np.pi)

    total += normal_logpdf(th[:3 * B], *prior_beta).sum() + normal_logpdf(al, *prior_alpha).sum()
    return float(total)


def literal_log_joint_rn(theta, site_covs, obs_covs, obs, max_abundance=100, prior_beta=(0.0, 1.0), prior_alpha=(0.0, 1.0),
                         site_random_effects=False, obs_random_effects=False, prior_site_re_sd=1.0, prior_obs_re_sd=1.0,
                         false_positives_constant=False, prior_fp=(2.0, 5.0)):
    """log p(theta, y) of the Royle-Nichols model, stated literally (biolith/models/occu_rn.py:123-222):
    N enumerated over 0..max_abundance under Categorical(logits=Poisson(lambda).log_prob(support))
    (utils/distributions.py:31-40; Categorical renormalises), Bernoulli(1-(1-r)^N) with clamp_probs."""
    from scipy.special import gammaln, logsumexp

    X = _as_f32_f64(site_covs)
    W = _as_f32_f64(obs_covs)
    Y = _as_f32_f64(obs)
    if Y.ndim == 4:
        Y = Y[0]
    Ks, Ko = X.shape[1], W.shape[-1]
    theta = np.asarray(theta, dtype=np.float64)
    beta, alpha = theta[: Ks + 1], theta[Ks + 1: Ks + Ko + 2]
    # random effects (occu_rn.py:151-154, 172-184, 199-212): theta = [beta, alpha, (log site_re_sd), (log obs_re_sd),
    # (site_re_abu[N], site_re_det[N]), (obs_re[N][T][J])]; a HalfNormal site lives on the log scale (+ log-Jacobian)
    N_, T_, J_ = Y.shape
    at, lp_re = Ks + Ko + 2, 0.0
    re_abu = re_det = np.zeros(N_)
    obs_re = np.zeros((N_, T_, J_))

    def half_normal_on_log_scale(phi, scale):
        return 0.5 * np.log(2.0 / np.pi) - np.log(scale) - 0.5 * (np.exp(phi) / scale) ** 2 + phi

    def normal0(v, sd):
        return (-0.5 * (v / sd) ** 2 - np.log(sd) - 0.5 * np.log(2 * np.pi)).sum()

    fpr = 0.0
    if false_positives_constant:   # occu_rn.py:133-138: the rate lives on the logit scale (+ log-Jacobian), right behind the coefficients
        from scipy.special import betaln
        phi = theta[at]; at += 1
        fpr = 1.0 / (1.0 + np.exp(-phi))
        a_, b_ = prior_fp
        lp_re += (a_ - 1.0) * np.log(fpr) + (b_ - 1.0) * np.log1p(-fpr) - betaln(a_, b_) + np.log(fpr) + np.log1p(-fpr)
    sd_s = sd_o = None
    if site_random_effects:
        lp_re += half_normal_on_log_scale(theta[at], prior_site_re_sd); sd_s = np.exp(theta[at]); at += 1
    if obs_random_effects:
        lp_re += half_normal_on_log_scale(theta[at], prior_obs_re_sd); sd_o = np.exp(theta[at]); at += 1
    if site_random_effects:
        re_abu, re_det = theta[at: at + N_], theta[at + N_: at + 2 * N_]; at += 2 * N_
        lp_re += normal0(re_abu, sd_s) + normal0(re_det, sd_s)
    if obs_random_effects:
        obs_re = theta[at: at + N_ * T_ * J_].reshape(N_, T_, J_); at += N_ * T_ * J_
        lp_re += normal0(obs_re, sd_o)
    assert at == theta.size
    obs_mask = np.isnan(W).any(-1) | np.isnan(X).any(-1)[:, None, None]       # occu_rn.py:124-130
    Y = np.where(obs_mask, np.nan, Y)
    W = np.nan_to_num(W)
    X = np.nan_to_num(X)
    abundance = np.exp(beta[0] + X @ beta[1:] + re_abu)                         # occu_rn.py:179-192
    support = np.arange(max_abundance + 1)
    logits = np.log(abundance)[:, None] * support - gammaln(support + 1) - abundance[:, None]   # Poisson.log_prob
    log_prior = logits - logsumexp(logits, axis=1, keepdims=True)              # (N, K+1)
    r = 1.0 / (1.0 + np.exp(-(alpha[0] + np.tensordot(W, alpha[1:], axes=([3], [0])) + re_det[:, None, None] + obs_re)))   # (N,T,J)  occu_rn.py:209-218
    finite = np.isfinite(Y)
    y0 = np.where(finite, Y, 0.0)
    p = 1.0 - (1.0 - r[..., None]) ** support                                   # (N,T,J,K+1)  occu_rn.py:219
    p = 1.0 - (1.0 - p) * (1.0 - fpr)                                           # occu_rn.py:214-221
    ly = _bernoulli_logpmf_clamped(p, y0[..., None])
    ly = np.where(finite[..., None], ly, 0.0).sum(axis=2)                       # (N,T,K+1)
    ll = logsumexp(log_prior[:, None, :] + ly, axis=2).sum()

    def normal_logpdf(v, loc, scale):
        return (-0.5 * ((v - loc) / scale) ** 2 - np.log(scale) - 0.5 * np.log(2 * np.pi)).sum()

    return ll + lp_re + normal_logpdf(beta, *prior_beta) + normal_logpdf(alpha, *prior_alpha)


# --------------------------------------------------------------------------------------------
# numpyro.diagnostics restatement [UPSTREAM] -- SURVEY.md Appendix B.6
# --------------------------------------------------------------------------------------------
def _next_fast_len(n: int) -> int:
    try:
        from scipy.fft import next_fast_len
        return int(next_fast_len(n))
    except Exception:  # pragma: no cover
        m = 1
        while m < n:
            m *= 2
        return m


def _autocovariance(x, axis):
    x = np.swapaxes(x, axis, -1)
    n = x.shape[-1]
    m2 = 2 * _next_fast_len(n)
    xc = x - x.mean(axis=-1, keepdims=True)
    f = np.fft.rfft(xc, n=m2, axis=-1)
    ac = np.fft.irfft(f * np.conjugate(f), n=m2, axis=-1)[..., :n]
    with np.errstate(invalid="ignore", divide="ignore"):
        ac = ac / ac[..., :1]
    ac = ac * x.var(axis=-1, keepdims=True)
    return np.swapaxes(ac, axis, -1)


def _chain_variance_stats(x):
    n = x.shape[1]
    var_within = x.var(axis=1, ddof=1).mean(axis=0)
    var_estimator = var_within * (n - 1) / n
    if x.shape[0] > 1:
        var_estimator = var_estimator + x.mean(axis=1).var(axis=0, ddof=1)
    else:
        var_within = var_estimator
    return var_within, var_estimator


def effective_sample_size(x):
    """x: (chains, draws, ...) -> n_eff of shape x.shape[2:]."""
    x = np.asarray(x, dtype=np.float64)
    assert x.ndim >= 2 and x.shape[1] >= 2
    gamma = _autocovariance(x, axis=1)
    var_within, var_estimator = _chain_variance_stats(x)
    with np.errstate(invalid="ignore", divide="ignore"):
        rho = (var_estimator - var_within + gamma.mean(axis=0)) / var_estimator
    rho[0] = 1.0
    if rho.shape[0] % 2:
        rho = rho[:-1]
    P = rho.reshape((-1, 2) + rho.shape[1:]).sum(axis=1)
    P = np.concatenate([P[:1], np.minimum.accumulate(P[1:].clip(min=0), axis=0)], axis=0)
    tau = -1.0 + 2.0 * P.sum(axis=0)
    return x.shape[0] * x.shape[1] / tau


def split_gelman_rubin(x):
    x = np.asarray(x, dtype=np.float64)
    h = x.shape[1] // 2
    y = np.concatenate([x[:, :h], x[:, -h:]], axis=0)
    var_within, var_estimator = _chain_variance_stats(y)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.sqrt(var_estimator / var_within)
