"""``predict`` -- drop-in for biolith/utils/predict.py:9-94 on the HIP engine.

The reference wraps the model in ``numpyro.infer.Predictive(model_fn, posterior_samples=mcmc.get_samples())``
and calls it on the (possibly new) covariates with ``obs`` withheld (predict.py:66-85), which
recomputes the deterministic sites for every posterior draw and samples the discrete sites.  Here the
same sites come from two C-ABI calls per species -- ``bl_deterministic`` and ``bl_predict``
(``include/biolith_hip.h``) -- on a device-resident copy of the covariates.  No NumPyro/JAX, no CPU fallback.
"""
from __future__ import annotations

import warnings
from typing import Callable, Optional

import numpy as np

from .data import prepare_data, rename_samples
from .mcmc import LazySamples


def predict(
    model_fn: Callable,
    mcmc,
    site_covs=None,
    obs_covs=None,
    obs=None,
    session_duration=None,
    num_samples: int = 1000,
    random_seed: int = 0,
    infer_discrete: bool = False,
    timeout: Optional[int] = None,
    **kwargs,
) -> dict:
    """Posterior predictive samples from a fitted model.

    Parameters are those of the reference ``predict`` (predict.py:9-62).  ``mcmc`` is the
    ``FitResult.mcmc`` that :func:`biolith_amd.utils.fit` returned; ``obs`` is accepted and ignored, as
    the reference drops it before calling the model (predict.py:78-80).  As with ``Predictive``, one
    predictive draw is made per posterior draw, so ``num_samples`` only triggers NumPyro's warning when
    it differs from the number of posterior draws.

    ``infer_discrete=True`` (predict.py:18, 71): NumPyro's ``Predictive`` then draws the discrete sites from their posterior
    given the OBSERVED sites of the model call -- but the reference withholds ``obs`` from that call (predict.py:78-80), so
    ``y`` is itself an unobserved discrete site and the pair (``z``, ``y``) is drawn from its joint distribution given the
    posterior draw of the continuous sites: the same distribution the default path samples ancestrally.  [UPSTREAM: numpyro's
    ``_predictive`` / ``_sample_posterior``; not executable in this image.]  Both settings therefore run the same kernels here;
    draw-level equality with NumPyro is impossible either way (threefry keys are not reproduced).

    Returns
    -------
    dict
        occu: ``psi`` (n, T, N, S), ``z`` (n, T, N, S) int32, ``prob_detection`` / ``prob_detection_fp``
        (n, J, T, N, S), ``y`` (n, J, T, N, S) int32 -- the sites of occu.py:207-241.
        occu_rn / nmixture: ``abundance``, ``N_i``, ``prob_detection``, ``y`` (occu_rn.py:192-221, nmixture.py:181-220).
        occu_cop: ``psi``, ``z``, ``rate_detection``, ``y`` (counts; occu_cop.py:222-255).
        occu_dyn (builder-defined): ``psi``, ``gamma``, ``epsilon`` (n, N, S), ``z`` (n, T, N, S), ``prob_detection``, ``y``.
        The three replicate-level arrays are materialised on first access.

    Examples
    --------
    >>> from biolith_amd.models import simulate, occu
    >>> from biolith_amd.utils import fit, predict
    >>> data, _ = simulate()
    >>> results = fit(occu, **data, num_samples=10, num_warmup=10, num_chains=1)
    >>> preds = predict(occu, results.mcmc, **data, num_samples=5)
    """
    if not callable(model_fn) or getattr(model_fn, "__biolith_amd_model__", None) is None:
        raise TypeError("predict(): model_fn must be a biolith_amd model (biolith_amd.models.occu / occu_rn)")
    infer_discrete = bool(infer_discrete)   # (same sites, same distribution: see the docstring)
    # (no warning: the reference hands the flag to Predictive silently, predict.py:67-72, and pipelines that run with warnings as
    # errors must not break on a design caveat -- the caveat is the docstring's and DESIGN.md section 3's)
    device = int(kwargs.pop("device", 0))

    site_covs, obs_covs, obs, session_duration, site_names, obs_names = prepare_data(
        site_covs, obs_covs, obs, session_duration
    )
    posterior = mcmc.get_samples()
    if getattr(model_fn, "__biolith_amd_model__", None) == "occu_dyn":
        return _predict_dyn(model_fn, posterior, site_covs, obs_covs, site_names, obs_names, num_samples, random_seed, kwargs)
    beta = np.asarray(posterior["beta"], dtype=np.float32)    # (n, S, Ks+1)
    alpha = np.asarray(posterior["alpha"], dtype=np.float32)  # (n, S, Ko+1)
    n, n_species = beta.shape[0], beta.shape[1]
    if num_samples is not None and num_samples != n:
        warnings.warn(f"Sample's batch dimension size {n} is different from the provided {num_samples} "
                      f"num_samples argument. Defaulting to {n}.", UserWarning, stacklevel=2)

    # the model is called without obs (predict.py:78-80); the validators want the species count, so an
    # all-missing observation array of the fitted species count stands in for it
    arguments = dict(site_covs=site_covs, obs_covs=obs_covs, session_duration=session_duration)
    valid = {k: v for k, v in arguments.items() if v is not None}
    blank = np.full((n_species,) + np.shape(obs_covs)[:3], np.nan, dtype=np.float32)
    spec = model_fn(**valid, obs=blank, **kwargs)
    if beta.shape[2] != spec.site_covs.shape[1] + 1 or alpha.shape[2] != spec.obs_covs.shape[3] + 1:
        raise ValueError("predict(): covariate counts differ from the fitted model's coefficients")

    from ..engine import OccuDataset
    from .fit import engine_options

    fp_site = f"prob_fp_{spec.extras['fp_mode']}" if spec.model == "occu_fp" else None
    if spec.model in ("occu_re", "occu_rn") and spec.extras.get("re_fp_mode") is not None:   # (random effects +) a false-positive rate: [beta, alpha, phi, log sds, effects]
        fp_site = f"prob_fp_{spec.extras['re_fp_mode']}"
    if fp_site is not None:
        rate = np.clip(np.asarray(posterior[fp_site], dtype=np.float64).reshape(n), 1e-300, 1 - 1e-16)
        phi = np.log(rate / (1.0 - rate)).astype(np.float32)[:, None]   # the engine's coordinate: logit(rate)
    if spec.model == "occu_cop" and spec.extras["fp_mode"] is not None:
        fp_site = f"rate_fp_{spec.extras['fp_mode']}"
        rate = np.maximum(np.asarray(posterior[fp_site], dtype=np.float64).reshape(n), 1e-300)
        phi = np.log(rate).astype(np.float32)[:, None]                  # the engine's coordinate: log(rate)

    def re_block(sp):
        """The random-effects coordinates of species ``sp`` in the engine's one-species layout (see fit._assemble)."""
        if spec.model != "occu_re" and not (spec.model in ("occu_rn", "nmixture", "occu_cop") and "site_random_effects" in spec.extras):
            return None
        cols = []
        if spec.extras["site_random_effects"]:
            cols.append(np.log(np.maximum(np.asarray(posterior["site_re_sd"], dtype=np.float64).reshape(n, 1), 1e-300)))
        if spec.extras["obs_random_effects"]:
            cols.append(np.log(np.maximum(np.asarray(posterior["obs_re_sd"], dtype=np.float64).reshape(n, 1), 1e-300)))
        if spec.extras["site_random_effects"]:   # (n, N, species)
            first = "site_re_abu" if spec.model in ("occu_rn", "nmixture") else "site_re_occ"   # (occu_rn.py:172-176, nmixture.py:166-169)
            cols += [np.asarray(posterior[first])[..., sp].reshape(n, -1), np.asarray(posterior["site_re_det"])[..., sp].reshape(n, -1)]
        if spec.extras["obs_random_effects"]:   # (n, J, T, N, species) -> [N][T][J]
            cols.append(np.asarray(posterior["obs_re"])[..., sp].transpose(0, 3, 2, 1).reshape(n, -1))
        return np.concatenate(cols, axis=1).astype(np.float32) if cols else None   # (occu_rn with a false-positive rate and no effects)

    if spec.model == "occu_cs":   # sites psi, z, f, s (occu_cs.py:196-232); mu / sigma travel in the engine's coordinates
        from ..engine import OccuDataset
        from .fit import engine_options

        post = {k: np.asarray(posterior[k], dtype=np.float64).reshape(n) for k in ("mu0", "mu1", "sigma0", "sigma1")}
        extra = np.stack([post["mu0"], np.log(np.maximum(post["mu1"] - post["mu0"], 1e-300)), np.log(post["sigma0"]),
                          np.log(post["sigma1"])], axis=1).astype(np.float32)
        ds = OccuDataset(spec.site_covs, spec.obs_covs, spec.obs, spec.prior_beta, spec.prior_alpha, device=device, model="occu_cs",
                         **engine_options(spec))
        draws = np.concatenate([beta[:, 0, :], alpha[:, 0, :], extra], axis=1)

        def run_cs():
            psi = ds.deterministic(draws, psi=True, prob_detection=False)[0]
            return (psi,) + ds.predictive_scores(draws, seed=int(random_seed) & (2 ** 64 - 1))

        if timeout is not None:
            from .misc import time_limit

            with time_limit(timeout):
                psi, z, f, s = run_cs()
        else:
            psi, z, f, s = run_cs()
        out = LazySamples()
        out["psi"], out["z"] = psi[..., None], z[..., None].astype(np.int32)
        out["f"], out["s"] = f[..., None].astype(np.int32), s[..., None]
        for k, v in post.items():   # (Predictive leaves the posterior's own sites out; log_likelihood needs them next to f)
            out[k] = v.astype(np.float32)
        return rename_samples(out, site_names, obs_names)

    def run():
        handles, first, latent, y8 = [], [], [], []
        for sp in range(n_species):
            ds = OccuDataset(spec.site_covs, spec.obs_covs, spec.obs[sp:sp + 1], spec.prior_beta, spec.prior_alpha,
                             device=device, model=spec.model, **engine_options(spec))
            rb = re_block(sp)
            draws = np.concatenate([beta[:, sp, :], alpha[:, sp, :]] + ([phi] if fp_site else []) + ([rb] if rb is not None else []), axis=1)
            first.append(ds.deterministic(draws, psi=True, prob_detection=False)[0])
            lat, yy = ds.predictive(draws, seed=(int(random_seed) + (sp << 32)) & (2 ** 64 - 1))
            latent.append(lat)
            y8.append(yy)
            handles.append((ds, draws))
        return handles, first, latent, y8

    if timeout is not None:
        from .misc import time_limit

        with time_limit(timeout):
            handles, first, latent, y8 = run()
    else:
        handles, first, latent, y8 = run()

    rn = spec.model in ("occu_rn", "nmixture")   # the abundance models: sites "abundance" and "N_i"
    out = LazySamples()
    out["abundance" if rn else "psi"] = np.stack(first, axis=-1)                      # (n, T, N, S)
    out["N_i" if rn else "z"] = np.stack(latent, axis=-1).astype(np.int32)            # (n, T, N, S)

    def prob_detection():
        return np.stack([d.deterministic(dr, psi=False, prob_detection=True)[1] for d, dr in handles], axis=-1)

    # occu_cop's replicate-level deterministic site is the detection rate (occu_cop.py:236-243)
    out.set_lazy("rate_detection" if spec.model == "occu_cop" else "prob_detection", prob_detection)   # (n, J, T, N, S)
    if spec.model in ("occu", "occu_fp", "occu_re"):
        def prob_detection_fp():  # occu.py:229-235
            z = np.stack(latent, axis=-1)[:, None].astype(np.float32)
            zp = prob_detection() * z
            if fp_site is None:   # both rates 0:  1 - (1 - z p) = z p
                return zp
            f = rate.astype(np.float32).reshape((n,) + (1,) * 4)
            f_c, f_u = (f, 0.0) if fp_site == "prob_fp_constant" else (0.0, f)
            return 1.0 - (1.0 - zp) * (1.0 - f_c) * (1.0 - (1.0 - z) * f_u)

        out.set_lazy("prob_detection_fp", prob_detection_fp)
    out.set_lazy("y", lambda: np.stack(y8, axis=-1).astype(np.int32))                 # (n, J, T, N, S)
    return rename_samples(out, site_names, obs_names)


def _predict_dyn(model_fn, posterior, site_covs, obs_covs, site_names, obs_names, num_samples, random_seed, kwargs):
    """Posterior predictive sites of the BUILDER-DEFINED dynamic occupancy model (models/occu_dyn.py; no reference counterpart): per
    posterior draw the latent path z_i1 ~ Bernoulli(psi_i), z_i,t+1 | z_it ~ Bernoulli(gamma_i) / Bernoulli(1 - epsilon_i) and
    y_itj ~ Bernoulli(z_it p_itj), drawn ancestrally on the host (predict is not on the hot path; the path has T dependent steps).
    Sites, species plate last as everywhere: ``psi``, ``gamma``, ``epsilon`` (n, N, 1); ``z`` (n, T, N, 1) int32; ``prob_detection`` and
    ``y`` (n, J, T, N, 1), materialised on first access.  The generator is keyed by ``random_seed``."""
    b = {k: np.asarray(posterior[k], dtype=np.float32)[:, 0, :] for k in ("beta", "beta_col", "beta_ext", "alpha")}
    n = b["beta"].shape[0]
    if num_samples is not None and num_samples != n:
        warnings.warn(f"Sample's batch dimension size {n} is different from the provided {num_samples} "
                      f"num_samples argument. Defaulting to {n}.", UserWarning, stacklevel=3)
    blank = np.full((1,) + np.shape(obs_covs)[:3], np.nan, dtype=np.float32)
    spec = model_fn(site_covs=site_covs, obs_covs=obs_covs, obs=blank, **kwargs)
    X = np.nan_to_num(np.asarray(spec.site_covs, dtype=np.float32))
    W = np.nan_to_num(np.asarray(spec.obs_covs, dtype=np.float32))
    if b["beta"].shape[1] != X.shape[1] + 1 or b["alpha"].shape[1] != W.shape[3] + 1:
        raise ValueError("predict(): covariate counts differ from the fitted model's coefficients")
    N, T, J = W.shape[:3]

    def prob(block):
        eta = block[:, :1] + block[:, 1:] @ X.T
        return (1.0 / (1.0 + np.exp(-eta))).astype(np.float32)   # (n, N)

    psi, gam, eps = prob(b["beta"]), prob(b["beta_col"]), prob(b["beta_ext"])
    rng = np.random.default_rng([int(random_seed) & 0x7FFFFFFF, 0xD1A])
    z = np.empty((n, T, N), dtype=np.int8)
    z[:, 0] = rng.random((n, N), dtype=np.float32) < psi
    for t in range(1, T):
        z[:, t] = rng.random((n, N), dtype=np.float32) < np.where(z[:, t - 1] == 1, 1.0 - eps, gam)

    def prob_detection():   # (n, J, T, N, 1)
        nu = b["alpha"][:, :1, None, None] + np.einsum("itjk,nk->nitj", W, b["alpha"][:, 1:])
        return (1.0 / (1.0 + np.exp(-nu))).astype(np.float32).transpose(0, 3, 2, 1)[..., None]

    def y():
        p = prob_detection()[..., 0]
        u = np.random.default_rng([int(random_seed) & 0x7FFFFFFF, 0xD1B]).random(p.shape, dtype=np.float32)
        return ((u < p) & (z[:, None] == 1)).astype(np.int32)[..., None]

    out = LazySamples()
    out["psi"], out["gamma"], out["epsilon"] = psi[..., None], gam[..., None], eps[..., None]
    out["z"] = z.astype(np.int32)[..., None]
    out.set_lazy("prob_detection", prob_detection)
    out.set_lazy("y", y)
    return rename_samples(out, site_names, obs_names)
