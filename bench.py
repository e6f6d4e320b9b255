#!/usr/bin/env python3
"""Headline benchmark: effective samples / second of psi for occu NUTS on 10k sites x 5 visits.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete pass of the hot path over the workload BASELINE.json's metric is quoted
on (its configs[1]): simulate(n_sites=10000, 3 site + 3 obs covariates, 5 visits), 4 chains per
GPU x (1000 warmup + 1000 draws) of NUTS, i.e. exactly what ``fit(occu, **data, num_chains=4)``
runs behind the C-ABI.  The dataset is resident in HBM before the timed region.  With N > 1 GPUs
every rank runs its own 4 chains (weak scaling; chains are the unit the path shards over, no
data-path collective) and the draws are all-gathered over RCCL inside the timed region.
ESS is computed afterwards with the NumPyro estimator (mean over sites of per-site ESS of psi).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG2 = dict(n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7, random_seed=0)
CFG4 = dict(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7, random_seed=0)
CHAINS_PER_GPU = 4
# --workload: "occu" is the headline (BASELINE.json configs[1], what the driver runs); "occu_rn" is the
# secondary line for configs[3] (same JSON shape, metric on `abundance`), run by hand for profiles/.
WORKLOADS = {
    "occu": dict(model="occu", cfg=CFG2, num_warmup=1000, num_samples=1000, cpu_sample=(250, 250), site="psi",
                 metric="effective samples/sec (psi) for occu NUTS, 10k sites x 5 visits",
                 text="biolith simulate(n_sites=10000, n_site_covs=3, n_obs_covs=3, 5 visits, seed 0); fit(occu)"),
    "occu_rn": dict(model="occu_rn", cfg=CFG4, num_warmup=500, num_samples=500, cpu_sample=(40, 40), site="abundance",
                    metric="effective samples/sec (abundance) for occu_rn NUTS, 5k sites x 10 visits",
                    text="biolith simulate_rn(n_sites=5000, n_site_covs=3, n_obs_covs=3, 10 visits, seed 0); "
                         "fit(occu_rn, max_abundance=100)"),
    # SURVEY section 8 row f3 (no BASELINE config of its own): occu with site random effects, 20 009 coordinates.  (With
    # observation effects as well the Bernoulli data leave obs_re_sd and the detection coefficients nearly unidentified: NUTS
    # mixes in neither the reference's parameterisation nor here, R-hat > 2 -- not a throughput workload.)
    "occu_re": dict(model="occu_re", cfg=dict(n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7,
                                              site_random_effects=True, random_seed=0),
                    num_warmup=1000, num_samples=1000, cpu_sample=(20, 20), site="psi",
                    options=dict(site_random_effects=True),
                    metric="effective samples/sec (psi) for occu NUTS with site random effects, 10k sites x 10 visits",
                    text="biolith simulate(n_sites=10000, n_site_covs=3, n_obs_covs=3, 10 visits, site random effects, seed 0); "
                         "fit(occu, site_random_effects=True)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes_per_eval(N, T, J, Ks, Ko, S=1):
    """float32 bytes one potential+gradient evaluation of one chain must read."""
    return 4 * (N * Ks + N * T * J * Ko + S * N * T * J)


def psi_draws(draws, X, site="psi", effects=None):
    """draws (C, S, D) -> psi (C, S, N) float32 = sigmoid(beta0 + X beta)  (occu.py:198-207), or
    abundance = exp(beta0 + X beta) for occu_rn (occu_rn.py:192).  effects: (C, S, N) site_re_occ draws of these sites."""
    Ks = X.shape[1]
    eta = draws[..., :1] + draws[..., 1:Ks + 1] @ X.T
    if effects is not None:
        eta = eta + effects
    return (np.exp(eta) if site == "abundance" else 1.0 / (1.0 + np.exp(-eta))).astype(np.float32)


class _DevArray:
    """Expose an engine-owned device buffer to torch (for the RCCL gather) without a copy."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr="<f4", data=(int(ptr), False), version=2)


def ess_of_site_function(draws, X, site="psi", chunk=1000, o_u=None):
    """Mean over sites of ESS(psi_i) (or abundance_i), computed over site chunks so that the (chains, draws, sites)
    array of the deterministic site never exists as a whole (8 GPUs x 4 chains x 1000 draws x 10 000 sites)."""
    from biolith_amd.evaluation import effective_sample_size

    total, n = 0.0, X.shape[0]
    for s0 in range(0, n, chunk):
        eff = None if o_u is None else draws[..., o_u + s0: o_u + min(s0 + chunk, n)]   # random effects: psi_i carries site_re_occ_i
        total += float(effective_sample_size(psi_draws(draws, X[s0:s0 + chunk], site, eff)).sum())
    return total / n


def cpu_baseline(data, threads, wl):
    """Oracle (port of the same algorithm, float64 C) on host cores, bounded sample of the workload."""
    import oracle
    from biolith_amd.evaluation import effective_sample_size

    od = oracle.OracleData(data["site_covs"], data["obs_covs"], data["obs"], model=wl["model"], **wl.get("options", {}))
    w, s = wl["cpu_sample"]
    t0 = time.perf_counter()
    r = oracle.nuts_run(od, w, s, num_chains=CHAINS_PER_GPU, seed=0, threads=threads)
    wall = time.perf_counter() - t0
    X = np.asarray(data["site_covs"], dtype=np.float32).astype(np.float64)
    ess = ess_of_site_function(r["draws"], X, wl["site"], o_u=od.Ks + od.Ko + 3 if wl["model"] == "occu_re" else None)
    nleap = int(r["n_leapfrog"].sum())
    return dict(value=ess / wall, unit="ESS/s", cores=int(r["threads"]), kind="port",
                sample=f"oracle NUTS (float64 C restatement), same data, {CHAINS_PER_GPU} chains x ({w} warmup + {s} draws) "
                       f"on {int(r['threads'])} threads: {wall:.1f} s, {nleap} gradient evaluations, "
                       f"{1e3 * wall * int(r['threads']) / nleap:.2f} ms per evaluation per core, ESS({wl['site']}) {ess:.0f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true",
                    help="occu_rn / occu_re: also time the oracle (minutes on 4 cores; off by default)")
    ap.add_argument("--wgs-per-chain", type=int, default=0)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="occu")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    NUM_WARMUP, NUM_SAMPLES = wl["num_warmup"], wl["num_samples"]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch

    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":  # the env knob exercises the RCCL path on one GPU
        import torch.distributed as dist

        # NCCL_DEBUG=VERSION (set on the GPU boxes) makes RCCL print a five-line banner on STDOUT when the first
        # communicator is created; stdout is reserved for the one JSON line, so that banner is turned off
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            del os.environ["NCCL_DEBUG"]
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    from biolith_amd.distributed import gather_draws
    from biolith_amd.engine import OccuDataset
    from biolith_amd.evaluation import effective_sample_size, split_gelman_rubin
    from biolith_amd.models import simulate, simulate_rn

    with contextlib.redirect_stdout(io.StringIO()):
        data, truth = (simulate_rn if wl["model"] == "occu_rn" else simulate)(**wl["cfg"])
    X = np.asarray(data["site_covs"], dtype=np.float32).astype(np.float64)
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"], device=local_rank,
                     model=wl["model"], **wl.get("options", {}))  # resident in HBM from here on
    stream = torch.cuda.current_stream().cuda_stream

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(step_seed):
        ds.launch(num_warmup=NUM_WARMUP, num_samples=NUM_SAMPLES, num_chains=CHAINS_PER_GPU, seed=step_seed,
                  chain_offset=rank * CHAINS_PER_GPU, wgs_per_chain=args.wgs_per_chain, stream=stream)
        ds.wait()
        res = ds.fetch()
        draws_all = res.draws
        if dist is not None:
            ptr, _ = ds.device_draws()
            # clone: the collective then runs on torch-allocated memory, not on the engine's own hipMalloc block
            local = torch.as_tensor(_DevArray(ptr, res.draws.shape), device=f"cuda:{local_rank}").clone()
            draws_all = gather_draws(local).cpu().numpy()  # RCCL all-gather: the path's only collective
        return res, draws_all

    for w in range(args.warmup):
        one_step(10_000 + w)
    sync_all()
    t0 = time.perf_counter()
    steps = [one_step(s) for s in range(args.steps)]
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-rank kernel statistics (HIP events on the launch stream, recorded inside the timed region)
    kernel_ms = np.array([r.kernel_ms for r, _ in steps])
    leap = np.array([int(r.n_leapfrog.sum()) + CHAINS_PER_GPU for r, _ in steps])  # + the initial evaluation of each chain
    if dist is not None:
        agg = torch.tensor([kernel_ms.mean(), leap.mean()], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        kernel_ms_mean, leap_mean = float(agg[0].item()) / world, float(agg[1].item()) / world
    else:
        kernel_ms_mean, leap_mean = float(kernel_ms.mean()), float(leap.mean())

    if rank == 0:
        # ---- metric numerator: ESS(psi), NumPyro estimator, mean over sites (diagnostics.py:28-32) ----
        ess_psi, ess_coef, rhat = [], [], []
        for _, d in steps:
            G0 = ds.Ks + ds.Ko + 2
            fixed = d[..., :G0] if wl["model"] == "occu_re" else d   # (with random effects: min ESS / R-hat over the fixed effects)
            ess_psi.append(ess_of_site_function(d.astype(np.float64), X, wl["site"], o_u=G0 + 1 if wl["model"] == "occu_re" else None))
            ess_coef.append(effective_sample_size(fixed).min())
            rhat.append(float(split_gelman_rubin(fixed).max()))
        total_ess = float(np.sum(ess_psi))
        N, T, J, Ks, Ko = ds.N, ds.T, ds.J, ds.Ks, ds.Ko
        bytes_eval = algorithmic_bytes_per_eval(N, T, J, Ks, Ko)
        if wl["model"] == "occu_re":   # + the sampler's own vectors: about 14 of them are read or written per leapfrog
            bytes_eval += 56 * ds.D
        res0 = steps[0][0]
        achieved = leap_mean * bytes_eval / (kernel_ms_mean * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and args.workload == "occu":  # the PMC passes were taken on the headline workload
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": wl["metric"],
            "value": total_ess / elapsed,
            "unit": "ESS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{wl['text']}: NUTS {CHAINS_PER_GPU} chains per GPU x ({NUM_WARMUP} warmup + {NUM_SAMPLES} draws), "
                            "one step = one full fit",
                "chains_per_gpu": CHAINS_PER_GPU, "total_chains": CHAINS_PER_GPU * world,
                "num_warmup": NUM_WARMUP, "num_samples": NUM_SAMPLES, "parallelism": f"chains x{world} (1 process per GPU)",
                "wgs_per_chain": res0.wgs_per_chain, "lds_bytes_per_wg": res0.lds_bytes, "lds_staged": res0.lds_staged,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                # <KS, KO, LDS-staged, model id, compute waves>: occu = 0; occu_rn = 5 (104-entry table, max_abundance <= 103)
                "kernel": "bl_re_nuts_kernel(BlReRun const*)" if wl["model"] == "occu_re" else
                          f"bl_nuts_kernel<{ds.Ks}, {ds.Ko}, true, {5 if wl['model'] == 'occu_rn' else 0}, {res0.threads_per_wg // 64 - 1}>",
                "kernel_ms": kernel_ms_mean,
                "algorithmic_bytes_per_launch": leap_mean * bytes_eval, "bytes_per_gradient_evaluation": bytes_eval,
                "gradient_evaluations_per_launch": leap_mean,
                "us_per_leapfrog_per_chain": 1e3 * kernel_ms_mean / (leap_mean / CHAINS_PER_GPU),
                "note": "data is LDS-resident after one HBM read, so this is an effective (algorithmic) bandwidth; "
                        "the sequential-leapfrog latency is the real bound",
            },
            "sampler": {  # rank 0's chains, last timed step (SURVEY.md section 8d "also report")
                "mean_num_steps": float(steps[-1][0].num_steps.mean()), "divergences": int(steps[-1][0].diverging.sum()),
                "step_size": [float(x) for x in steps[-1][0].step_size], "mean_accept_prob": float(steps[-1][0].accept_prob.mean()),
                "leapfrogs_per_s_per_chain": 1e3 * (leap_mean / CHAINS_PER_GPU) / kernel_ms_mean,
            },
            "ess": {f"{wl['site']}_mean_over_sites_per_step": ess_psi, "min_coef_per_step": [float(x) for x in ess_coef],
                    "max_split_rhat": max(rhat), "draws_per_step": CHAINS_PER_GPU * world * NUM_SAMPLES},
        }
        want_cpu = not args.no_cpu_baseline and (args.workload == "occu" or args.cpu_baseline)
        if world == 1 and want_cpu:
            out["cpu_baseline"] = cpu_baseline(data, threads=min(CHAINS_PER_GPU, os.cpu_count() or 1), wl=wl)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
