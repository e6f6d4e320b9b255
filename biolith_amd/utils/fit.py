"""``fit`` -- drop-in for biolith/utils/fit.py:16-135 on the HIP engine.

Same signature, defaults and return type.  Where the reference builds ``MCMC(NUTS(model_fn))`` and
calls ``mcmc.run`` (fit.py:92-130), this lowers the model onto the C-ABI in
``include/biolith_hip.h``: one dataset upload, one persistent-kernel launch for all chains
(warmup + sampling), one copy of the draws back.  There is no NumPyro/JAX and no CPU fallback.
"""
from __future__ import annotations

from collections import namedtuple
from typing import Callable, Optional

import numpy as np

from .data import prepare_data, rename_samples
from .mcmc import HipMCMC

FitResult = namedtuple("FitResult", ["samples", "mcmc"])

_OTHER_KERNELS = ("hmc", "mixed_hmc", "discrete_hmc_gibbs", "hmcecs")


def fit(
    model_fn: Callable,
    site_covs=None,
    obs_covs=None,
    obs=None,
    session_duration=None,
    num_samples: int = 1000,
    num_warmup: int = 1000,
    random_seed: int = 0,
    num_chains: int = 5,
    kernel: Optional[str] = None,
    init_strategy: Optional[Callable] = None,
    timeout: Optional[int] = None,
    **kwargs,
) -> FitResult:
    """Fit an occupancy model with NUTS on an MI355X.

    Parameters are those of the reference ``fit`` (fit.py:33-66).  ``model_fn`` must be
    :func:`biolith_amd.models.occu` or :func:`biolith_amd.models.occu_rn`.  ``kernel`` must be ``None`` or ``"nuts"``; ``init_strategy``
    must be ``None`` (= ``init_to_uniform``, fit.py:93).  Extra keyword arguments go to the model,
    plus two engine knobs that the reference does not have: ``device`` (GPU index, default 0) and
    ``chain_offset`` (global id of the first chain, for sharding chains over processes).

    Returns
    -------
    FitResult
        ``samples``: dict ``cov_state_*`` / ``cov_det_*`` of shape (chains*draws, n_species),
        ``psi`` (chains*draws, T, N, S), ``prob_detection`` (chains*draws, J, T, N, S; lazy);
        ``mcmc``: :class:`HipMCMC`.
    """
    if not callable(model_fn) or getattr(model_fn, "__biolith_amd_model__", None) is None:
        raise TypeError(
            "fit(): model_fn must be a biolith_amd model (biolith_amd.models.occu); NumPyro model "
            "functions cannot run on the HIP engine"
        )
    if kernel is None:
        kernel = "nuts"
    if kernel in _OTHER_KERNELS:
        raise NotImplementedError(f"kernel={kernel!r}: the HIP engine implements NUTS only (fit.py:92-104)")
    if kernel != "nuts":
        raise KeyError(kernel)
    if init_strategy is not None:
        raise NotImplementedError("init_strategy: only the default init_to_uniform (fit.py:93) is built")
    device = int(kwargs.pop("device", 0))
    chain_offset = int(kwargs.pop("chain_offset", 0))

    site_covs, obs_covs, obs, session_duration, site_names, obs_names = prepare_data(
        site_covs, obs_covs, obs, session_duration
    )
    arguments = dict(site_covs=site_covs, obs_covs=obs_covs, obs=obs, session_duration=session_duration)
    valid = {k: v for k, v in arguments.items() if v is not None}
    spec = model_fn(**valid, **kwargs)

    from ..engine import OccuDataset

    ds = OccuDataset(spec.site_covs, spec.obs_covs, spec.obs, spec.prior_beta, spec.prior_alpha, device=device,
                     model=spec.model, max_abundance=spec.extras.get("max_abundance", 100))
    try:
        run_kw = dict(num_warmup=num_warmup, num_samples=num_samples, num_chains=num_chains,
                      seed=random_seed, chain_offset=chain_offset)
        if timeout is not None:
            from .misc import time_limit

            with time_limit(timeout):
                res = ds.nuts(timeout=None if timeout is None else float(timeout) + 1.0, **run_kw)
        else:
            res = ds.nuts(**run_kw)
        mcmc = _assemble(ds, spec, res, num_warmup)
    finally:
        pass  # ds stays alive inside the lazy prob_detection thunk; freed with the FitResult
    samples = rename_samples(mcmc.get_samples(), site_names, obs_names)
    return FitResult(samples, mcmc)


def _assemble(ds, spec, res, num_warmup) -> HipMCMC:
    """Draws (C, S, D) -> the sample sites the reference's model emits (occu.py:185-228)."""
    C, S, D = res.draws.shape
    Ks, Ko = ds.Ks, ds.Ko
    nsp = 1
    # plate "species" is the last axis of every site (occu.py:182, dim=-1)
    beta = res.draws[:, :, : Ks + 1].reshape(C, S, nsp, Ks + 1)
    alpha = res.draws[:, :, Ks + 1:].reshape(C, S, nsp, Ko + 1)
    flat = res.draws.reshape(C * S, D)
    psi, _ = ds.deterministic(flat, psi=True, prob_detection=False) if S else (np.empty((0, ds.T, ds.N), np.float32), None)
    psi = psi.reshape(C, S, ds.T, ds.N, nsp)

    def prob_detection():
        _, pd = ds.deterministic(flat, psi=False, prob_detection=True)
        return pd.reshape(C, S, ds.J, ds.T, ds.N, nsp)

    # occu emits "psi" (occu.py:207); occu_rn emits "abundance" = exp(linear predictor) (occu_rn.py:192)
    first = "abundance" if spec.model == "occu_rn" else "psi"
    return HipMCMC(res, latent=dict(beta=beta, alpha=alpha),
                   deterministic={first: psi, "prob_detection": prob_detection},
                   num_warmup=num_warmup, spec_shape=spec.shape)
