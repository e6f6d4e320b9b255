"""``fit`` -- drop-in for biolith/utils/fit.py:16-135 on the HIP engine.

Same signature, defaults and return type.  Where the reference builds ``MCMC(NUTS(model_fn))`` and
calls ``mcmc.run`` (fit.py:92-130), this lowers the model onto the C-ABI in
``include/biolith_hip.h``: one dataset upload, one persistent-kernel launch for all chains
(warmup + sampling), one copy of the draws back.  There is no NumPyro/JAX and no CPU fallback.
"""
from __future__ import annotations

import os
import time
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

    Parameters are those of the reference ``fit`` (fit.py:33-66).  ``model_fn`` must be one of ``biolith_amd.models``' model functions
    (``occu``, ``occu_rn``, ``occu_cop``, ``nmixture``, ``occu_cs``, and the builder-defined ``occu_dyn``).  ``kernel`` must be ``None`` or ``"nuts"``; ``init_strategy``
    is ``None`` (= ``init_to_uniform``, fit.py:93) or one of :mod:`biolith_amd.utils.init`'s strategies (NumPyro's callables of the same
    names are recognised).  Extra keyword arguments go to the model,
    plus engine knobs that the reference does not have: ``device`` (GPU index, default 0), ``devices``
    (list of GPU indices: the chains are dealt over them in contiguous blocks and sampled concurrently,
    the in-process counterpart of ``chain_method="parallel"``, fit.py:109-113) and ``chain_offset``
    (global id of the first chain, for sharding chains over processes).

    Returns
    -------
    FitResult
        ``samples``: dict ``cov_state_*`` / ``cov_det_*`` of shape (chains*draws, n_species),
        ``psi`` (chains*draws, T, N, S), ``prob_detection`` (chains*draws, J, T, N, S; lazy), ``prob_detection_fp``
        (chains*draws, 2, J, T, N, S; lazy; the leading 2 is the enumerated z, occu.py:229-235);
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
    from .init import as_strategy, initial_positions

    strategy = as_strategy(init_strategy)   # None = init_to_uniform (fit.py:93); NotImplementedError for anything unknown
    device = int(kwargs.pop("device", 0))
    devices = kwargs.pop("devices", None)
    explicit_devices = devices is not None
    devices = [device] if devices is None else [int(d) for d in devices]
    if not devices:
        raise ValueError("devices must name at least one GPU")
    chain_offset = int(kwargs.pop("chain_offset", 0))
    joint_species = bool(kwargs.pop("joint_species", True))

    site_covs, obs_covs, obs, session_duration, site_names, obs_names = prepare_data(
        site_covs, obs_covs, obs, session_duration
    )
    arguments = dict(site_covs=site_covs, obs_covs=obs_covs, obs=obs, session_duration=session_duration)
    valid = {k: v for k, v in arguments.items() if v is not None}
    spec = model_fn(**valid, **kwargs)

    from ..distributed import shard_chains
    from ..engine import OccuDataset

    # The "species" plate (occu.py:182-186) carries its own beta/alpha per species over shared covariates,
    # so the joint posterior is a product over species: each species gets its own device dataset and
    # its own chains (distinct RNG streams); the draws are stacked on the trailing species axis.
    n_species = spec.obs.shape[0]
    # Several species: the reference samples ONE chain over all species' coefficients (the species plate sits inside one NUTS,
    # occu.py:182-186; a false-positive rate is shared across the plate, occu.py:146-157).  occu with or without false positives
    # does the same here -- one device dataset holding all species, theta = [species 0: beta, alpha | species 1: ... | (phi)], one
    # step size, one tree per transition -- as long as the joint vector fits the kernel's 60 lanes and the records fit LDS
    # (``joint_species=False`` forces the species-by-species form, which has the same marginals but its own adaptation per species).
    # Random effects: their sds sit outside the species plate (occu.py:170-173), so several species are one chain by definition.
    if n_species > 1 and spec.model == "occu_re" and not joint_species:
        raise NotImplementedError("random effects share their sds across species (occu.py:170-173): joint_species=False does not apply")
    joint = n_species > 1 and spec.model in ("occu", "occu_fp", "occu_re") and joint_species
    # chain_method="parallel" (fit.py:109-113) deals chains over the local devices; here ``devices``
    # names the GPUs, chains go to them in contiguous blocks, every shard is an asynchronous launch and
    # the draws are concatenated on the chain axis afterwards (no exchange while sampling).
    world = len(devices)

    def make_jobs(joint):
        jobs = []
        for sp in ([0] if joint else range(n_species)):
            for r, dev in enumerate(devices):
                count, first = shard_chains(num_chains, world, r)
                if count == 0:
                    continue
                ds = OccuDataset(spec.site_covs, spec.obs_covs, spec.obs if joint else spec.obs[sp:sp + 1], spec.prior_beta, spec.prior_alpha,
                                 device=dev, model=spec.model, **engine_options(spec))
                kw = dict(num_warmup=num_warmup, num_samples=num_samples, num_chains=count,
                          seed=random_seed, chain_offset=chain_offset + sp * num_chains + first)
                nsp_here = n_species if joint else 1
                init = initial_positions(strategy, D=ds.D, Ks=ds.Ks, Ko=ds.Ko, n_species=nsp_here, plain=ds.D == nsp_here * (ds.Ks + ds.Ko + 2) and spec.model != "occu_dyn", blocks_first=spec.model != "occu_dyn",
                                         prior_beta=spec.prior_beta, prior_alpha=spec.prior_alpha, num_chains=count,
                                         first_chain=kw["chain_offset"], seed=random_seed, species=sp)
                if init is not None:
                    kw["init_theta"] = init
                jobs.append((sp, ds, kw))
        return jobs

    try:
        jobs = make_jobs(joint)
    except NotImplementedError:
        if not joint or spec.model in ("occu_fp", "occu_re"):   # (a shared false-positive rate / shared sds cannot be sampled species by species)
            raise
        joint = False
        jobs = make_jobs(False)
    n_units = 1 if joint else n_species

    def run_all(wgs_per_chain=None):
        # (wgs_per_chain: one entry per job -- the engine-timeout retry below -- or None = the engine's own choice)
        t_end = None if timeout is None else time.monotonic() + float(timeout) + 1.0
        launched = []
        try:
            for j, (_, ds, kw) in enumerate(jobs):   # launches on one device queue behind each other, devices overlap
                ds.launch(**kw, wgs_per_chain=0 if wgs_per_chain is None else wgs_per_chain[j])
                launched.append(ds)
            for ds in launched:
                while t_end is not None and not ds.done():
                    if time.monotonic() > t_end:
                        raise TimeoutError("Timed out")
                    time.sleep(0.0005)
                ds.wait()
        except BaseException:
            for ds in launched:      # reference-style SIGALRM TimeoutException, Ctrl-C or an engine error
                ds.abort()
            for ds in launched:
                try:
                    ds.wait()
                except Exception:
                    pass
            raise
        if use_rccl:
            # the gather of fit.py:132, as ONE RCCL all-gather per species behind the C-ABI (bl_gather_draws): every
            # device contributes the result block of its chains; the host reads all of them from the first device
            out = {}
            for sp in range(n_units):
                mine = [(ds, kw["num_chains"]) for s, ds, kw in jobs if s == sp]
                out[sp] = gather_draws(comms, [ds for ds, _ in mine], [c for _, c in mine])
            return out
        return [ds.fetch() for _, ds, _ in jobs]

    # devices=[...] with distinct GPUs (one is enough): communicators from ncclCommInitAll (made before the clock of a
    # timeout starts; their cost is reported in mcmc.result.comm_init_ms).  A device named twice (tests on a one-GPU box)
    # cannot hold two RCCL ranks: the shards are then fetched one by one and concatenated on the host.
    used = [dev for r, dev in enumerate(devices) if shard_chains(num_chains, world, r)[0] > 0]
    # (BIOLITH_TEST_ALLOW_DUP_DEVICES=1: tests only -- with tests/fake_rccl, a double of the collective that lets the several-ranks
    # branch of bl_gather_draws run on a one-GPU box.  Not keyed on BIOLITH_RCCL_LIB, which may simply name a real librccl elsewhere.)
    use_rccl = explicit_devices and (len(set(used)) == len(used) or os.environ.get("BIOLITH_TEST_ALLOW_DUP_DEVICES") == "1")
    comms = []
    if use_rccl:
        from .._ffi import BL_ERR_COMM, EngineError
        from ..distributed import comms_for_devices, gather_draws

        try:
            comms = comms_for_devices(used)
        except EngineError as exc:
            # librccl missing or communicator creation refused: the gather is a convenience of the multi-GPU form, not a
            # requirement of the fit -- every shard is then fetched by itself and the chains are concatenated on the host
            if exc.code != BL_ERR_COMM:
                raise
            import warnings

            warnings.warn(f"fit(devices={devices}): no RCCL communicator ({exc}); gathering the chains on the host instead", RuntimeWarning)
            use_rccl = False

    def run_retrying():
        try:
            return run_all()
        except TimeoutError as exc:
            # BL_ERR_TIMEOUT from the ENGINE (not the caller's time limit, which says "Timed out"): a chain's workgroups spin on each
            # other's partial sums and must all be resident; on a shared or profiled device they may not be.  One fresh launch on half
            # the workgroups per chain -- every job's own count halved; nothing of the failed launch is reused -- before giving up.
            # (A halved count whose slices no longer fit LDS is raised again by the engine for the LDS-only models; should the second
            # launch be refused all the same, the caller sees the ORIGINAL timeout, with the refusal as its cause.)
            if not str(exc).startswith("biolith_hip:"):
                raise
            ks = [ds.wgs_per_chain() for _, ds, _ in jobs]
            if max(ks) <= 1:
                raise
            import warnings

            half = [max(1, k // 2) for k in ks]
            warnings.warn(f"fit(): the engine's exchange timed out with {max(ks)} workgroups per chain ({exc}); retrying once with {max(half)}", RuntimeWarning)
            try:
                return run_all(wgs_per_chain=half)
            except TimeoutError:
                raise
            except Exception as second:  # noqa: BLE001 -- the retry's own refusal must not mask the timeout
                raise exc from second

    def run_with_fallback():
        # Joint sampling of several species needs all species' records in LDS for the chain count asked for; that is decided by the
        # launch (geometry depends on num_chains), which refuses with BL_ERR_UNSUPPORTED before anything runs.  Plain occu then
        # falls back to one dataset and one sampler per species, as documented above (same marginals).
        nonlocal jobs, joint, n_units
        try:
            return run_retrying()
        except NotImplementedError:
            if not joint or spec.model != "occu":
                raise
            joint = False
            jobs = make_jobs(False)
            n_units = n_species
            return run_retrying()

    try:
        if timeout is not None:
            from .misc import time_limit

            with time_limit(timeout):
                results = run_with_fallback()
        else:
            results = run_with_fallback()
    finally:
        for c in comms:
            c.close()
    per_species = []
    for sp in range(n_units):
        if use_rccl:
            ds0 = next(ds for s, ds, _ in jobs if s == sp)
            results[sp].comm_init_ms = comms[0].init_ms
            per_species.append((ds0, results[sp]))
            continue
        shard = [(ds, res) for (s, ds, _), res in zip(jobs, results) if s == sp]
        per_species.append((shard[0][0], _concat_chains([r for _, r in shard])))
    joint_result = None
    if joint:
        # one launch holds every species: cut its draws into the per-species blocks the site assembly works on (deterministic
        # sites come from one plain handle per species; the shared false-positive coordinate rides along with every block)
        import copy

        ds_joint, joint_result = per_species[0]
        Dsp = ds_joint.Ks + ds_joint.Ko + 2
        per_species = []
        jd = joint_result.draws
        for sp in range(n_species):
            part = copy.copy(joint_result)
            if spec.model == "occu_re":
                # [species' beta, alpha | log sds | site_re_occ [S][N] | site_re_det [S][N] | obs_re [S][N][T][J]] -> the one-species layout
                N, V = ds_joint.N, ds_joint.T * ds_joint.J
                nsd = int(ds_joint.site_re) + int(ds_joint.obs_re)
                at = n_species * Dsp + nsd
                blocks = [jd[:, :, sp * Dsp:(sp + 1) * Dsp], jd[:, :, n_species * Dsp: at]]
                if ds_joint.site_re:
                    blocks += [jd[:, :, at + sp * N: at + (sp + 1) * N], jd[:, :, at + (n_species + sp) * N: at + (n_species + sp + 1) * N]]
                    at += 2 * n_species * N
                if ds_joint.obs_re:
                    blocks.append(jd[:, :, at + sp * N * V: at + (sp + 1) * N * V])
                part.draws = np.ascontiguousarray(np.concatenate(blocks, axis=2))
            else:
                part.draws = np.ascontiguousarray(np.concatenate([jd[:, :, sp * Dsp:(sp + 1) * Dsp], jd[:, :, n_species * Dsp:]], axis=2))
            per_species.append((OccuDataset(spec.site_covs, spec.obs_covs, spec.obs[sp:sp + 1], spec.prior_beta, spec.prior_alpha,
                                            device=devices[0], model=spec.model, **engine_options(spec)), part))
    if spec.model == "occu_dyn":
        mcmc = _assemble_dyn(per_species[0], spec, num_warmup)
        samples = rename_samples(mcmc.get_samples(), site_names, obs_names)
        for prefix, base in (("cov_col_", "beta_col"), ("cov_ext_", "beta_ext")):   # the two extra predictors share the site covariates' names
            block = samples.pop(base)
            for i, name in enumerate(site_names):
                samples[f"{prefix}{name}"] = block[..., i]
        return FitResult(samples, mcmc)
    mcmc = _assemble(per_species, spec, num_warmup, joint_result)
    samples = rename_samples(mcmc.get_samples(), site_names, obs_names)
    return FitResult(samples, mcmc)


def engine_options(spec) -> dict:
    """Model options of an ``OccuSpec`` as ``OccuDataset`` keyword arguments."""
    opts = dict(max_abundance=spec.extras.get("max_abundance", 100))
    if spec.model == "occu_fp":
        opts.update(fp_mode=spec.extras["fp_mode"], prior_fp=spec.extras["prior_fp"])
    if spec.model == "occu_cop":
        opts.update(fp_mode=spec.extras["fp_mode"], session_duration=spec.extras["session_duration"],
                    prior_fp_rate=spec.extras.get("prior_fp_rate", 1.0))
    if spec.model == "occu_re":
        opts.update({k: spec.extras[k] for k in ("site_random_effects", "obs_random_effects", "prior_site_re_sd", "prior_obs_re_sd")})
        if spec.extras.get("re_fp_mode") is not None:
            opts.update(re_fp_mode=spec.extras["re_fp_mode"], prior_fp=spec.extras["prior_fp"])
    if spec.model in ("nmixture", "occu_rn", "occu_cop") and "site_random_effects" in spec.extras:
        opts.update({k: spec.extras[k] for k in ("site_random_effects", "obs_random_effects", "prior_site_re_sd", "prior_obs_re_sd")})
        if spec.extras.get("re_fp_mode") is not None:   # occu_rn with a false-positive rate
            opts.update(re_fp_mode=spec.extras["re_fp_mode"], prior_fp=spec.extras["prior_fp"])
    if spec.model == "occu_cs":
        opts.update(prior_mu=spec.extras["prior_mu"], prior_sigma=spec.extras["prior_sigma"])
    return opts


def _concat_chains(parts):
    """Shards of one species' chains (one NutsResult per device) -> one NutsResult, chain axis first."""
    if len(parts) == 1:
        return parts[0]
    import copy

    res = copy.copy(parts[0])
    for name in ("draws", "diverging", "num_steps", "accept_prob", "potential_energy", "step_size", "inv_mass", "n_leapfrog"):
        setattr(res, name, np.concatenate([getattr(r, name) for r in parts], axis=0))
    res.kernel_ms = float(max(r.kernel_ms for r in parts))   # shards on different devices overlap
    res.chains_l2_local = int(sum(r.chains_l2_local for r in parts))
    return res


def _assemble_dyn(ds_res, spec, num_warmup) -> HipMCMC:
    """Dynamic occupancy (builder-defined, models/occu_dyn.py): theta = [b_psi | b_col | b_ext | alpha] -> sample sites, with the
    species plate last as everywhere (one species); psi / gamma / epsilon per site are formed lazily on the host."""
    ds0, res = ds_res
    C, S, D = res.draws.shape
    Ks, Ko, B = ds0.Ks, ds0.Ko, ds0.Ks + 1
    blocks = [res.draws[:, :, b * B:(b + 1) * B][:, :, None, :] for b in range(3)]
    latent = dict(beta=blocks[0], beta_col=blocks[1], beta_ext=blocks[2], alpha=res.draws[:, :, 3 * B:][:, :, None, :])
    X = np.nan_to_num(np.asarray(spec.site_covs, dtype=np.float32))

    def site(block):
        def get():
            b = block[:, :, 0, :].astype(np.float32)
            eta = b[..., :1] + b[..., 1:] @ X.T
            return (1.0 / (1.0 + np.exp(-eta)))[..., None].astype(np.float32)      # (C, S, N, species)
        return get

    return HipMCMC(res, latent=latent, deterministic=dict(psi=site(blocks[0]), gamma=site(blocks[1]), epsilon=site(blocks[2])),
                   num_warmup=num_warmup, spec_shape=spec.shape)


def _assemble(per_species, spec, num_warmup, joint_result=None) -> HipMCMC:
    """Draws (C, S, D) per species -> the sample sites the reference's model emits (occu.py:185-228)."""
    ds0, res0 = per_species[0]
    C, S, D = res0.draws.shape
    Ks, Ko = ds0.Ks, ds0.Ko
    nsp = len(per_species)
    # plate "species" is the last axis of every site (occu.py:182, dim=-1)
    beta = np.stack([r.draws[:, :, : Ks + 1] for _, r in per_species], axis=2)     # (C, S, nsp, Ks+1)
    alpha = np.stack([r.draws[:, :, Ks + 1: Ks + Ko + 2] for _, r in per_species], axis=2)  # (C, S, nsp, Ko+1)
    latent = dict(beta=beta, alpha=alpha)
    if spec.model == "occu_fp":
        # the engine samples phi = logit(rate); the model's site is the rate itself, shape (C, S) (occu.py:146-157)
        phi = res0.draws[:, :, Ks + Ko + 2].astype(np.float64)
        latent[f"prob_fp_{spec.extras['fp_mode']}"] = (1.0 / (1.0 + np.exp(-phi))).astype(np.float32)
    if spec.model == "occu_cop" and spec.extras["fp_mode"] is not None:
        # phi = log(rate); the model's site is the rate (occu_cop.py:158-170)
        latent[f"rate_fp_{spec.extras['fp_mode']}"] = np.exp(res0.draws[:, :, Ks + Ko + 2].astype(np.float64)).astype(np.float32)
    if spec.model == "occu_cs":
        # theta = [beta, alpha, mu0, log(mu1 - mu0), log sigma0, log sigma1]; the model's sites are mu0, mu1, sigma0, sigma1 (occu_cs.py:143-152)
        e = res0.draws[:, :, Ks + Ko + 2:].astype(np.float64)
        latent["mu0"] = e[..., 0].astype(np.float32)
        latent["mu1"] = (e[..., 0] + np.exp(e[..., 1])).astype(np.float32)
        latent["sigma0"], latent["sigma1"] = np.exp(e[..., 2]).astype(np.float32), np.exp(e[..., 3]).astype(np.float32)
    if spec.model == "occu_re" or (spec.model in ("nmixture", "occu_rn", "occu_cop") and "site_random_effects" in spec.extras):
        # theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_occ[N], site_re_det[N]), (obs_re[N][T][J])]; the
        # model's sites are the sds themselves, the effects with the species plate last (occu.py:170-173, 191-196, 215-218)
        N, T, J = ds0.N, ds0.T, ds0.J
        at = Ks + Ko + 2
        if spec.model == "occu_cop" and spec.extras["fp_mode"] is not None:   # [beta, alpha, phi = log(rate), log sds, effects]: the rate was read above
            at += 1
        if spec.extras.get("re_fp_mode") is not None:   # [beta, alpha, phi = logit(rate), log sds, effects]
            phi = res0.draws[:, :, at].astype(np.float64)
            latent[f"prob_fp_{spec.extras['re_fp_mode']}"] = (1.0 / (1.0 + np.exp(-phi))).astype(np.float32)
            at += 1
        if spec.extras["site_random_effects"]:
            latent["site_re_sd"] = np.exp(res0.draws[:, :, at].astype(np.float64)).astype(np.float32)
            at += 1
        if spec.extras["obs_random_effects"]:
            latent["obs_re_sd"] = np.exp(res0.draws[:, :, at].astype(np.float64)).astype(np.float32)
            at += 1
        if spec.extras["site_random_effects"]:
            # (the abundance models name the first one after their predictor: nmixture.py:166-169, occu_rn.py:172-176)
            latent["site_re_abu" if spec.model in ("nmixture", "occu_rn") else "site_re_occ"] = np.stack([r.draws[:, :, at: at + N] for _, r in per_species], axis=-1)   # (C, S, N, nsp)
            latent["site_re_det"] = np.stack([r.draws[:, :, at + N: at + 2 * N] for _, r in per_species], axis=-1)
            at += 2 * N
        if spec.extras["obs_random_effects"]:
            e = np.stack([r.draws[:, :, at: at + N * T * J].reshape(C, S, N, T, J) for _, r in per_species], axis=-1)
            latent["obs_re"] = np.ascontiguousarray(e.transpose(0, 1, 4, 3, 2, 5))                                   # (C, S, J, T, N, nsp)
    def memo(fn):   # a site is computed once, on its first access
        box = []

        def get():
            if not box:
                box.append(fn())
            return box[0]
        return get

    @memo
    def psi():
        # (as the reference's samples stay on the device until they are looked at, so does this site: 160 MB over PCIe at the
        # headline size would otherwise be a sixth of fit()'s wall time)
        if not S:
            return np.empty((C, 0, ds0.T, ds0.N, nsp), np.float32)
        parts = [d.deterministic(r.draws.reshape(C * S, D), psi=True, prob_detection=False)[0] for d, r in per_species]
        # one species (the common case): a view, not a 160 MB copy at the headline size
        out = parts[0][..., None] if nsp == 1 else np.stack(parts, axis=-1)
        return out.reshape(C, S, ds0.T, ds0.N, nsp)

    @memo
    def prob_detection():
        parts = [d.deterministic(r.draws.reshape(C * S, D), psi=False, prob_detection=True)[1] for d, r in per_species]
        pd = parts[0][..., None] if nsp == 1 else np.stack(parts, axis=-1)
        return pd.reshape(C, S, ds0.J, ds0.T, ds0.N, nsp)

    @memo
    def prob_detection_fp():
        # occu.py:229-235: 1 - (1 - z p)(1 - f_c)(1 - (1 - z) f_u) depends on the ENUMERATED z, so under numpyro's parallel
        # enumeration (occu.py:208-210) the recorded value carries z's enumeration axis in front of the plates
        # (first available dim = -5: [UPSTREAM] funsor's enum dims sit left of max_plate_nesting = 4; not verifiable in this
        # image): (C, S, 2, J, T, N, species), index 0 = unoccupied, 1 = occupied.  Nothing in the reference reads it.
        pd = prob_detection().astype(np.float32)                                  # (C, S, J, T, N, nsp)
        mode = spec.extras.get("fp_mode") if spec.model == "occu_fp" else spec.extras.get("re_fp_mode")
        if mode is None:   # both rates 0:  1 - (1 - z p) = z p
            return np.stack([np.zeros_like(pd), pd], axis=2)
        rate = latent[f"prob_fp_{mode}"].astype(np.float32).reshape(C, S, 1, 1, 1, 1)
        f_c, f_u = (rate, np.float32(0.0)) if mode == "constant" else (np.float32(0.0), rate)
        z0 = np.broadcast_to(1.0 - (1.0 - f_c) * (1.0 - f_u), pd.shape).astype(np.float32)
        z1 = (1.0 - (1.0 - pd) * (1.0 - f_c)).astype(np.float32)
        return np.stack([z0, z1], axis=2)

    # one result object for the mcmc shim: extras concatenated over species along the chain axis
    import copy

    res = copy.copy(res0)
    if joint_result is not None:
        res = copy.copy(joint_result)    # one chain over all species: its per-draw fields are the sampler's own
    elif nsp > 1:
        res.diverging = np.logical_or.reduce([r.diverging for _, r in per_species])
        res.num_steps = np.sum([r.num_steps for _, r in per_species], axis=0)
        res.accept_prob = np.mean([r.accept_prob for _, r in per_species], axis=0)
        res.potential_energy = np.sum([r.potential_energy for _, r in per_species], axis=0)
        res.n_leapfrog = np.sum([r.n_leapfrog for _, r in per_species], axis=0)
        res.kernel_ms = float(np.sum([r.kernel_ms for _, r in per_species]))
        res.inv_mass = np.concatenate([r.inv_mass for _, r in per_species], axis=1)
    if joint_result is None:
        res.draws = np.concatenate([r.draws for _, r in per_species], axis=2) if nsp > 1 else res0.draws
    # occu emits "psi" (occu.py:207); occu_rn emits "abundance" = exp(linear predictor) (occu_rn.py:192)
    first = "abundance" if spec.model in ("occu_rn", "nmixture") else "psi"
    # occu_cop's replicate-level site is the detection RATE exp(linear predictor) (occu_cop.py:236-243)
    second = "rate_detection" if spec.model == "occu_cop" else "prob_detection"
    sites = {first: psi, second: prob_detection}
    if spec.model in ("occu", "occu_fp", "occu_re"):
        sites["prob_detection_fp"] = prob_detection_fp
    return HipMCMC(res, latent=latent,
                   deterministic=sites,
                   num_warmup=num_warmup, spec_shape=spec.shape)
