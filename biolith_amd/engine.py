"""Python handle over the HIP engine's C-ABI: dataset upload, log-density hook, NUTS driver.

This is the host-side counterpart of what ``mcmc.run`` did in the reference
(biolith/utils/fit.py:105-130).  It only marshals NumPy buffers through ``ctypes``; all arithmetic
of the hot path runs in the gfx950 kernels under ``csrc/``.
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _ffi


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


class _PinnedPool:
    """Page-locked host buffers for the large deterministic sites (psi at the headline size: 160 MB).  Copying into page-locked
    memory runs at PCIe rate (3 ms instead of 10 for those 160 MB), but pinning it is slow (60 ms), so: the FIRST request of a size
    class gets ordinary pages at once while a buffer of that class is pinned in the background; later requests take a pooled
    buffer, which returns to the pool when the array that used it is gone.  At most ``cap`` bytes stay pooled."""

    def __init__(self, cap=1 << 30):
        import threading

        self.free, self.cap, self.held, self.pending, self.wanted, self.lock = {}, cap, 0, set(), set(), threading.Lock()

    def empty(self, lib, shape, dtype):
        import weakref

        count = int(np.prod(shape))
        size = 1 << max(23, (count * np.dtype(dtype).itemsize - 1).bit_length())   # power-of-two size classes
        with self.lock:
            bucket = self.free.get(size)
            ptr = bucket.pop() if bucket else None
            if ptr is not None:
                self.held -= size
            elif size not in self.pending and size not in self.wanted and self.held + size <= self.cap:
                self.wanted.add(size)   # (pinned once the copy that is about to run is over: kick())
        if ptr is None:
            return np.empty(shape, dtype=dtype)
        buf = (C.c_char * size).from_address(ptr)
        arr = np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)
        weakref.finalize(buf, self._give_back, lib, size, ptr)       # (the array keeps ``buf`` alive through its base chain)
        return arr

    def kick(self, lib):
        """Start pinning the size classes asked for since the last call (after the device copy: the two would contend)."""
        import threading

        with self.lock:
            todo, self.wanted = self.wanted, set()
            self.pending |= todo
        for size in todo:
            threading.Thread(target=self._warm, args=(lib, size)).start()   # (not a daemon: never killed inside a driver call at exit)

    def _warm(self, lib, size):
        p = C.c_void_p()
        ok = lib.bl_host_alloc(size, C.byref(p)) == 0
        with self.lock:
            self.pending.discard(size)
        if ok:
            self._give_back(lib, size, p.value)

    def _give_back(self, lib, size, ptr):
        with self.lock:
            keep = self.held + size <= self.cap
            if keep:
                self.free.setdefault(size, []).append(ptr)
                self.held += size
        if not keep:
            lib.bl_host_free(C.c_void_p(ptr))


_PINNED = _PinnedPool()


@dataclass
class NutsResult:
    draws: np.ndarray             # (C, S, D) float32
    diverging: np.ndarray         # (C, S) bool
    num_steps: np.ndarray         # (C, S) int32
    accept_prob: np.ndarray       # (C, S) float32
    potential_energy: np.ndarray  # (C, S) float32
    step_size: np.ndarray         # (C,)
    inv_mass: np.ndarray          # (C, D)
    n_leapfrog: np.ndarray        # (C, 2) int64: warmup, sampling gradient evaluations
    kernel_ms: float              # HIP-event time of the persistent kernel
    wgs_per_chain: int
    lds_bytes: int
    lds_staged: bool
    chains_l2_local: int = 0      # chains that ran the verified same-XCD (L2-local) exchange
    threads_per_wg: int = 0       # 64 x (compute waves + 1 control wave)
    lds_vector_tier: int = 0      # random-effects / occu_cs kernels: sampler vectors kept in LDS (0 none, 1 the leaf in flight, 2 all a leapfrog touches)
    comm_init_ms: float = 0.0     # fit(devices=[...]): wall time of ncclCommInitAll (outside the sampling clock)
    lane_group: tuple = (1, 1)    # lanes that shared one site pair: (period lanes, visit lanes); (1, 1) = one pair per lane
    kernel_name: str = ""         # the sampler instantiation that ran, as rocprofv3 names it
    env_overrides: str = ""       # BIOLITH_HIP_* knobs set at the launch ("" = none: the engine's own geometry and kernel form)


class OccuDataset:
    """Device-resident occupancy dataset (one species).

    Parameters mirror ``occu``'s data arguments (biolith/models/occu.py:19-40):
    ``site_covs (N, Ks)``, ``obs_covs (N, T, J, Ko)``, ``obs (S=1, N, T, J)``; NaN = missing.
    """

    def __init__(self, site_covs, obs_covs, obs, prior_beta=(0.0, 1.0), prior_alpha=(0.0, 1.0), device: int = 0,
                 model: str = "occu", max_abundance: int = 100, fp_mode: Optional[str] = "constant", prior_fp=(2.0, 5.0),
                 session_duration=None, prior_fp_rate: float = 1.0, site_random_effects: bool = False,
                 obs_random_effects: bool = False, prior_site_re_sd: float = 1.0, prior_obs_re_sd: float = 1.0,
                 prior_mu=((0.0, 10.0), (0.0, 10.0)), prior_sigma=((5.0, 1.0), (5.0, 1.0)), re_fp_mode: Optional[str] = None):
        lib = _ffi.load()
        if model not in ("occu", "occu_rn", "occu_fp", "occu_cop", "nmixture", "occu_re", "occu_cs", "occu_dyn"):
            raise ValueError(f"unknown model {model!r}")
        if fp_mode not in ("constant", "unoccupied") and not (model == "occu_cop" and fp_mode is None):
            raise ValueError(f"unknown fp_mode {fp_mode!r}")
        self.model, self.max_abundance = model, int(max_abundance)
        self.fp_mode, self.prior_fp = fp_mode, (float(prior_fp[0]), float(prior_fp[1]))
        X = np.ascontiguousarray(site_covs, dtype=np.float32)
        W = np.ascontiguousarray(obs_covs, dtype=np.float32)
        Y = np.ascontiguousarray(obs, dtype=np.float32)
        if X.ndim != 2:
            raise ValueError("site_covs must be of shape (n_sites, n_site_covs)")
        if W.ndim != 4:
            raise ValueError("obs_covs must be of shape (n_sites, n_periods, n_replicates, n_obs_covs)")
        if Y.ndim != 4:
            raise ValueError("obs must be of shape (n_species, n_sites, n_periods, n_replicates)")
        N, Ks = X.shape
        _, T, J, Ko = W.shape
        if W.shape[0] != N:
            raise ValueError("site_covs and obs_covs must have the same number of sites")
        if Y.shape[1:] != (N, T, J):
            raise ValueError("obs must have shape (n_species, n_sites, n_periods, n_replicates) matching obs_covs")
        self.dims = _ffi.bl_dims(Y.shape[0], N, T, J, Ks, Ko)
        self.N, self.T, self.J, self.Ks, self.Ko, self.S = N, T, J, Ks, Ko, Y.shape[0]
        # occu_fp: trailing phi = logit(false-positive rate); occu_cop with a false-positive rate: phi = log(rate)
        self.D = Ks + Ko + 2 + (1 if model == "occu_fp" or (model == "occu_cop" and fp_mode is not None) else 0)
        if self.S > 1:
            # several species in one handle: ONE chain over theta = [species 0: beta, alpha | species 1: ... | (shared phi)]
            # (the species plate of occu.py:182-186 under one NUTS); occu with or without false positives, or with random
            # effects: theta = [... | log sds | site_re_occ [S][N] | site_re_det [S][N] | obs_re [S][N][T][J]], the sds shared
            if model not in ("occu", "occu_fp", "occu_re"):
                raise NotImplementedError(f"{model}: one species per dataset (joint sampling is built for occu / occu_fp / occu_re)")
            self.D += (self.S - 1) * (Ks + Ko + 2)
        self.device = device
        pb = _ffi.bl_normal_prior(float(prior_beta[0]), float(prior_beta[1]))
        pa = _ffi.bl_normal_prior(float(prior_alpha[0]), float(prior_alpha[1]))
        h = C.c_void_p()
        if model == "occu_dyn":
            # builder-defined dynamic occupancy (no reference counterpart): theta = [b_psi | b_gamma | b_eps (Ks+1 each) | alpha (Ko+1)]
            if self.S != 1:
                raise NotImplementedError("occu_dyn: one species per dataset")
            _ffi.check(lib.bl_dataset_create_dyn(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), C.byref(pb), C.byref(pa), device, C.byref(h)))
            self.D = 3 * (Ks + 1) + Ko + 1
        elif model == "occu_rn" and re_fp_mode is not None:
            # Royle-Nichols with a false-positive rate (with or without random effects): theta = [beta, alpha, phi = logit(rate), (log sds), (effects)]
            if re_fp_mode != "constant":
                raise ValueError("occu_rn: the false-positive rate acts on every site (re_fp_mode='constant')")
            pf = _ffi.bl_beta_prior(*self.prior_fp)
            _ffi.check(lib.bl_dataset_create_rn_fp(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), int(max_abundance),
                                                   int(bool(site_random_effects)), int(bool(obs_random_effects)),
                                                   float(prior_site_re_sd), float(prior_obs_re_sd), C.byref(pf), C.byref(pb), C.byref(pa),
                                                   device, C.byref(h)))
            d = C.c_int()
            _ffi.check(lib.bl_dataset_param_dim(h, C.byref(d)))
            self.D = int(d.value)
        elif model == "occu_rn" and (site_random_effects or obs_random_effects):
            # theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_abu[N], site_re_det[N]), (obs_re[N][T][J])]
            _ffi.check(lib.bl_dataset_create_rn_re(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), int(max_abundance),
                                                   int(bool(site_random_effects)), int(bool(obs_random_effects)),
                                                   float(prior_site_re_sd), float(prior_obs_re_sd), C.byref(pb), C.byref(pa),
                                                   device, C.byref(h)))
            d = C.c_int()
            _ffi.check(lib.bl_dataset_param_dim(h, C.byref(d)))
            self.D = int(d.value)
        elif model == "occu_rn":
            _ffi.check(lib.bl_dataset_create_rn(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), int(max_abundance),
                                                C.byref(pb), C.byref(pa), device, C.byref(h)))
        elif model == "nmixture":
            with np.errstate(invalid="ignore"):
                if np.isfinite(Y).any() and np.nanmax(Y) > max_abundance:
                    raise ValueError(f"max_abundance={max_abundance} is below the largest count {np.nanmax(Y):g}")
            if site_random_effects or obs_random_effects:
                # theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_abu[N], site_re_det[N]), (obs_re[N][T][J])]
                _ffi.check(lib.bl_dataset_create_nmix_re(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), int(max_abundance),
                                                         int(bool(site_random_effects)), int(bool(obs_random_effects)),
                                                         float(prior_site_re_sd), float(prior_obs_re_sd), C.byref(pb), C.byref(pa),
                                                         device, C.byref(h)))
                d = C.c_int()
                _ffi.check(lib.bl_dataset_param_dim(h, C.byref(d)))
                self.D = int(d.value)
            else:
                _ffi.check(lib.bl_dataset_create_nmix(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), int(max_abundance),
                                                      C.byref(pb), C.byref(pa), device, C.byref(h)))
        elif model == "occu_cop":
            if session_duration is None:
                raise ValueError("occu_cop needs session_duration (n_sites, n_periods, n_replicates)")
            Dur = np.ascontiguousarray(session_duration, dtype=np.float32)
            if Dur.shape != (N, T, J):
                raise ValueError("session_duration must have shape (n_sites, n_periods, n_replicates)")
            mode = {None: 0, "constant": _ffi.FP_CONSTANT, "unoccupied": _ffi.FP_UNOCCUPIED}[fp_mode]
            if site_random_effects or obs_random_effects:
                # theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_occ[N], site_re_det[N]), (obs_re[N][T][J])]
                # (+ phi = log rate_fp right behind the coefficients with a false-positive rate)
                _ffi.check(lib.bl_dataset_create_cop_re(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), _fp(Dur), mode, float(prior_fp_rate),
                                                        int(bool(site_random_effects)), int(bool(obs_random_effects)),
                                                        float(prior_site_re_sd), float(prior_obs_re_sd), C.byref(pb), C.byref(pa), device, C.byref(h)))
                d = C.c_int()
                _ffi.check(lib.bl_dataset_param_dim(h, C.byref(d)))
                self.D = int(d.value)
            else:
                _ffi.check(lib.bl_dataset_create_cop(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), _fp(Dur), mode,
                                                     float(prior_fp_rate), C.byref(pb), C.byref(pa), device, C.byref(h)))
        elif model == "occu_cs":
            # obs holds the scores; theta = [beta, alpha, mu0, log(mu1 - mu0), log sigma0, log sigma1]
            pm = np.ascontiguousarray(np.asarray(prior_mu, dtype=np.float64).reshape(4))
            ps = np.ascontiguousarray(np.asarray(prior_sigma, dtype=np.float64).reshape(4))
            _ffi.check(lib.bl_dataset_create_cs(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), _dp(pm), _dp(ps),
                                                C.byref(pb), C.byref(pa), device, C.byref(h)))
            self.D = Ks + Ko + 6
        elif model == "occu_re":
            # theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_occ[N], site_re_det[N]), (obs_re[N][T][J])]
            if not (site_random_effects or obs_random_effects):
                raise ValueError("occu_re needs site_random_effects and / or obs_random_effects")
            if re_fp_mode is not None:
                # ... together with a false-positive rate (occu.py:146-157): theta = [beta, alpha, phi = logit(rate), log sds, effects]
                if re_fp_mode not in ("constant", "unoccupied"):
                    raise ValueError(f"unknown re_fp_mode {re_fp_mode!r}")
                pf = _ffi.bl_beta_prior(*self.prior_fp)
                _ffi.check(lib.bl_dataset_create_re_fp(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), int(bool(site_random_effects)),
                                                       int(bool(obs_random_effects)), float(prior_site_re_sd), float(prior_obs_re_sd),
                                                       _ffi.FP_CONSTANT if re_fp_mode == "constant" else _ffi.FP_UNOCCUPIED, C.byref(pf),
                                                       C.byref(pb), C.byref(pa), device, C.byref(h)))
            else:
                _ffi.check(lib.bl_dataset_create_re(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), int(bool(site_random_effects)),
                                                    int(bool(obs_random_effects)), float(prior_site_re_sd), float(prior_obs_re_sd),
                                                    C.byref(pb), C.byref(pa), device, C.byref(h)))
            d = C.c_int()
            _ffi.check(lib.bl_dataset_param_dim(h, C.byref(d)))
            self.D = int(d.value)
        elif model == "occu_fp":
            pf = _ffi.bl_beta_prior(*self.prior_fp)
            mode = _ffi.FP_CONSTANT if fp_mode == "constant" else _ffi.FP_UNOCCUPIED
            _ffi.check(lib.bl_dataset_create_fp(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), mode, C.byref(pf),
                                                C.byref(pb), C.byref(pa), device, C.byref(h)))
        else:
            _ffi.check(lib.bl_dataset_create(C.byref(self.dims), _fp(X), _fp(W), _fp(Y), C.byref(pb), C.byref(pa),
                                             device, C.byref(h)))
        self._h = h
        self._lib = lib
        # Laplace instead of Normal coefficient priors (distributions.LocScale.family; utils/grid_search.py:366-371)
        fam = [getattr(p, "family", "normal") for p in (prior_beta, prior_alpha)]
        if any(f not in ("normal", "laplace") for f in fam):
            raise ValueError(f"unknown prior family {fam}")
        if "laplace" in fam:
            _ffi.check(lib.bl_dataset_set_prior_family(h, int(fam[0] == "laplace"), int(fam[1] == "laplace")))
        self.site_re, self.obs_re = bool(site_random_effects), bool(obs_random_effects)
        self.re_fp_mode = re_fp_mode if model in ("occu_re", "occu_rn") else None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bl_dataset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ K1 parity hook ----
    def logp_grad(self, theta, staged: bool = True):
        """U(theta) = -log p(theta, y) and dU/dtheta; theta (D,) or (B, D)."""
        th = np.ascontiguousarray(theta, dtype=np.float64)
        single = th.ndim == 1
        th2 = th.reshape(1, -1) if single else th
        if th2.shape[1] != self.D:
            raise ValueError(f"theta must have {self.D} columns")
        B = th2.shape[0]
        U = np.empty(B)
        G = np.empty((B, self.D))
        _ffi.check(self._lib.bl_logp_grad(self._h, B, _dp(th2), _dp(U), _dp(G), int(staged)))
        return (U[0], G[0]) if single else (U, G)

    # --------------------------------------------------------------------------- NUTS ----
    def _config(self, num_warmup, num_samples, num_chains, seed, chain_offset, max_tree_depth, target_accept,
                wgs_per_chain, init_theta):
        cfg = _ffi.bl_nuts_config()
        cfg.num_warmup, cfg.num_samples, cfg.num_chains = int(num_warmup), int(num_samples), int(num_chains)
        cfg.chain_offset, cfg.seed = int(chain_offset), int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.max_tree_depth, cfg.wgs_per_chain = int(max_tree_depth), int(wgs_per_chain)
        cfg.target_accept = float(target_accept)
        keep = None
        if init_theta is not None:
            keep = np.ascontiguousarray(init_theta, dtype=np.float64)
            if keep.shape != (num_chains, self.D):
                raise ValueError("init_theta must have shape (num_chains, D)")
            cfg.init_theta = _dp(keep)
        return cfg, keep

    def launch(self, num_warmup=1000, num_samples=1000, num_chains=1, seed=0, chain_offset=0, max_tree_depth=10,
               target_accept=0.8, wgs_per_chain=0, init_theta=None, stream: Optional[int] = None):
        """Asynchronous launch of the persistent kernel (returns immediately)."""
        cfg, keep = self._config(num_warmup, num_samples, num_chains, seed, chain_offset, max_tree_depth,
                                 target_accept, wgs_per_chain, init_theta)
        _ffi.check(self._lib.bl_nuts_launch(self._h, C.byref(cfg), C.c_void_p(stream or 0)))
        self._shape = (int(num_chains), int(num_samples))

    def done(self) -> bool:
        d = C.c_int(0)
        _ffi.check(self._lib.bl_nuts_poll(self._h, C.byref(d)))
        return bool(d.value)

    def abort(self):
        self._lib.bl_nuts_abort(self._h)

    def wait(self):
        _ffi.check(self._lib.bl_nuts_wait(self._h))

    def wgs_per_chain(self) -> int:
        """Workgroups per chain of the last launch (bl_nuts_geometry)."""
        k, thr, lds, staged, loc = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _ffi.check(self._lib.bl_nuts_geometry(self._h, C.byref(k), C.byref(thr), C.byref(lds), C.byref(staged), C.byref(loc)))
        return int(k.value)

    def _kernel_name(self) -> str:
        buf = C.create_string_buffer(192)
        _ffi.check(self._lib.bl_nuts_kernel_name(self._h, buf, 192))
        return buf.value.decode()

    def env_overrides(self) -> str:
        """BIOLITH_HIP_* knobs that were set when the last launch read the environment ("" = none; bl_nuts_env_overrides)."""
        if not hasattr(self._lib, "bl_nuts_env_overrides"):   # (a library built before round 6, loaded by name for an A/B)
            return ""
        buf = C.create_string_buffer(512)
        _ffi.check(self._lib.bl_nuts_env_overrides(self._h, buf, 512))
        return buf.value.decode()

    def elapsed_ms(self) -> float:
        ms = C.c_float(0)
        _ffi.check(self._lib.bl_nuts_elapsed_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def debug_counters(self, n: int = 32):
        """In-kernel phase cycle counters (zeros unless a BL_STAMPS diagnostic library is loaded)."""
        out = np.zeros(n, dtype=np.int64)
        _ffi.check(self._lib.bl_nuts_debug_counters(self._h, out.ctypes.data_as(C.POINTER(C.c_int64)), n))
        return out

    def device_draws(self):
        """(device pointer, bytes) of the last launch's draws [C][S][D] float32."""
        p, n = C.c_void_p(), C.c_size_t()
        _ffi.check(self._lib.bl_nuts_device_draws(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def _output(self, Cn: int, S: int):
        """Host arrays for ``Cn`` chains x ``S`` draws and the ``bl_nuts_output`` that points at them."""
        D = self.D
        a = dict(draws=np.empty((Cn, S, D), dtype=np.float32), diverging=np.empty((Cn, S), dtype=np.uint8),
                 num_steps=np.empty((Cn, S), dtype=np.int32), accept_prob=np.empty((Cn, S), dtype=np.float32),
                 potential_energy=np.empty((Cn, S), dtype=np.float32), step_size=np.empty(Cn, dtype=np.float32),
                 inv_mass=np.empty((Cn, D), dtype=np.float32), n_leapfrog=np.empty((Cn, 2), dtype=np.int64))
        out = _ffi.bl_nuts_output(
            _fp(a["draws"]), a["diverging"].ctypes.data_as(C.POINTER(C.c_uint8)), a["num_steps"].ctypes.data_as(C.POINTER(C.c_int32)),
            _fp(a["accept_prob"]), _fp(a["potential_energy"]), _fp(a["step_size"]), _fp(a["inv_mass"]),
            a["n_leapfrog"].ctypes.data_as(C.POINTER(C.c_int64)),
        )
        return a, out

    def _result(self, a) -> NutsResult:
        k, thr, lds, staged, loc = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _ffi.check(self._lib.bl_nuts_geometry(self._h, C.byref(k), C.byref(thr), C.byref(lds), C.byref(staged), C.byref(loc)))
        gt, gj = C.c_int(1), C.c_int(1)
        _ffi.check(self._lib.bl_nuts_lane_group(self._h, C.byref(gt), C.byref(gj)))
        return NutsResult(a["draws"], a["diverging"].astype(bool), a["num_steps"], a["accept_prob"], a["potential_energy"],
                          a["step_size"], a["inv_mass"], a["n_leapfrog"], self.elapsed_ms(),
                          k.value, lds.value, bool(staged.value & 1), loc.value, thr.value, staged.value >> 1,
                          lane_group=(int(gt.value), int(gj.value)), kernel_name=self._kernel_name(), env_overrides=self.env_overrides())

    def fetch(self) -> NutsResult:
        Cn, S = self._shape
        a, out = self._output(Cn, S)
        _ffi.check(self._lib.bl_nuts_fetch(self._h, C.byref(out)))
        return self._result(a)

    def nuts(self, timeout: Optional[float] = None, **kw) -> NutsResult:
        """launch + wait + fetch.  With ``timeout`` (seconds) the kernel is aborted through its
        host-mapped flag and ``TimeoutError`` raised (fit(timeout=...), biolith/utils/fit.py:124-128)."""
        self.launch(**kw)
        if timeout is not None:
            t_end = time.monotonic() + float(timeout)
            try:
                while not self.done():
                    if time.monotonic() > t_end:
                        raise TimeoutError("Timed out")
                    time.sleep(0.0005)
            except BaseException:
                # also reached for the reference-style SIGALRM TimeoutException or Ctrl-C
                self.abort()
                try:
                    self.wait()
                except Exception:
                    pass
                raise
        self.wait()
        return self.fetch()

    # ------------------------------------------------------------ deterministic sites ----
    def _draw_matrix(self, draws):
        """draws (..., D) -> contiguous float32 (n, D); a different trailing size (e.g. a random-effects posterior of
        another number of sites) is rejected instead of being silently re-shaped."""
        d = np.asarray(draws, dtype=np.float32)
        if d.ndim == 0 or d.shape[-1] != self.D:
            raise ValueError(f"draws must have {self.D} coordinates on the last axis for this dataset, got shape {d.shape}")
        return np.ascontiguousarray(d).reshape(-1, self.D)

    def _big_empty(self, shape, dtype=np.float32):
        """Output array for a device -> host copy: page-locked (pooled, see ``_PinnedPool``) from 8 MB on."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if nbytes < (8 << 20):
            return np.empty(shape, dtype=dtype)
        return _PINNED.empty(self._lib, shape, dtype)

    def deterministic(self, draws, psi: bool = True, prob_detection: bool = False):
        """psi -- or, for occu_rn, abundance -- (n, T, N) and/or prob_detection (n, J, T, N) for draws (n, D)
        (occu.py:207,221-228; occu_rn.py:192,209-218)."""
        d = self._draw_matrix(draws)
        n = d.shape[0]
        out_psi = self._big_empty((n, self.T, self.N)) if psi else None
        out_pd = self._big_empty((n, self.J, self.T, self.N)) if prob_detection else None
        _ffi.check(self._lib.bl_deterministic(self._h, n, _fp(d), _fp(out_psi), _fp(out_pd)))
        _PINNED.kick(self._lib)
        return out_psi, out_pd


    def predictive(self, draws, seed: int = 0, latent: bool = True, y: bool = True):
        """Posterior predictive draws of the discrete sites for draws (n, D): the latent state
        (``z`` for occu / occu_cop, ``N_i`` for occu_rn / nmixture) as (n, T, N) and ``y`` as (n, J, T, N); uint8, or
        int32 for the count models
        (biolith/utils/predict.py:66-92; occu.py:208-241; occu_rn.py:194-221)."""
        d = self._draw_matrix(draws)
        n = d.shape[0]
        counts = self.model in ("occu_cop", "nmixture")  # their sampled sites are counts: int32 (bl_predict_counts)
        dt, ct = (np.int32, C.c_int32) if counts else (np.uint8, C.c_uint8)
        out_l = np.empty((n, self.T, self.N), dtype=dt) if latent else None
        out_y = np.empty((n, self.J, self.T, self.N), dtype=dt) if y else None
        if n:
            ptr = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(ct))
            fn = self._lib.bl_predict_counts if counts else self._lib.bl_predict
            _ffi.check(fn(self._h, n, _fp(d), C.c_uint64(int(seed) & (2 ** 64 - 1)), ptr(out_l), ptr(out_y)))
        return out_l, out_y


def _predictive_scores(self, draws, seed: int = 0):
    """occu_cs: posterior predictive ``z`` (n, T, N), ``f`` (n, J, T, N) as uint8 and the scores ``s`` (n, J, T, N) float32
    (biolith/models/occu_cs.py:196-232 with obs withheld)."""
    d = self._draw_matrix(draws)
    n = d.shape[0]
    z = np.empty((n, self.T, self.N), dtype=np.uint8)
    f = np.empty((n, self.J, self.T, self.N), dtype=np.uint8)
    s = np.empty((n, self.J, self.T, self.N), dtype=np.float32)
    if n:
        u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
        _ffi.check(self._lib.bl_predict_scores(self._h, n, _fp(d), C.c_uint64(int(seed) & (2 ** 64 - 1)), u8(z), u8(f), _fp(s)))
    return z, f, s


OccuDataset.predictive_scores = _predictive_scores


def rng_streams(seed: int, chain: int, nstreams: int = _ffi.RNG_STREAMS_PER_CHAIN) -> np.ndarray:
    out = np.zeros((nstreams, 4), dtype=np.uint32)
    _ffi.check(_ffi.load().bl_rng_streams(int(seed), int(chain), int(nstreams),
                                          out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out


def adaptation_schedule(num_warmup: int):
    s = (C.c_int32 * 40)()
    e = (C.c_int32 * 40)()
    n = _ffi.load().bl_adaptation_schedule(int(num_warmup), s, e, 40)
    return [(s[i], e[i]) for i in range(n)]
