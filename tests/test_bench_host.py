"""Host-only pieces of bench.py and of the multi-rank test double (no GPU): which chains a rank runs (BASELINE.json configs[2] is
`--gpus 8 --chains-per-gpu 1`), the per-workload chain counts, and that the collective's double exports the ten symbols
csrc/comm_rccl.hpp resolves (reference: chain_method="parallel", biolith/utils/fit.py:109-113)."""
import ctypes
import os
import subprocess

import numpy as np

import bench
from biolith_amd.distributed import shard_chains

HERE = os.path.dirname(os.path.abspath(__file__))


def test_rank_shard_deals_consecutive_global_chain_ids():
    for chains in (1, 2, 4):
        seen = []
        for r in range(8):
            s = bench.rank_shard(r, chains)
            assert s["num_chains"] == chains
            seen += list(range(s["chain_offset"], s["chain_offset"] + chains))
            # the same dealing as fit(devices=[...])'s shard_chains when the chains divide evenly
            assert shard_chains(8 * chains, 8, r) == (chains, s["chain_offset"])
        assert seen == list(range(8 * chains))


def test_chains_per_gpu_flag_and_workload_defaults():
    a = bench.parse_args(["--gpus", "8", "--chains-per-gpu", "1", "--steps", "3"])
    assert (a.gpus, a.chains_per_gpu, a.steps) == (8, 1, 3)
    assert bench.parse_args([]).chains_per_gpu == 0
    assert bench.workload_chains(bench.WORKLOADS["occu"]) == 4            # BASELINE configs[1]: 4 chains on 1 GPU
    assert bench.workload_chains(bench.WORKLOADS["occu_cfg1"]) == 2       # configs[0]: 2 chains
    for wl in bench.WORKLOADS.values():
        assert bench.workload_chains(wl, 1) == 1                          # the flag overrides every workload
    assert {w for w, _, _ in bench.SECONDARY} == {"occu_rn", "occu_re", "occu_stacked", "occu_dyn", "occu_cfg1"}


def test_algorithmic_bytes_are_surveys_figures():
    assert bench.algorithmic_bytes_per_eval(10000, 1, 5, 3, 3) == 920000   # configs[1] / [2]
    assert bench.algorithmic_bytes_per_eval(100, 1, 52, 1, 1) == 42000     # configs[0]
    assert bench.algorithmic_bytes_per_eval(2000, 8, 4, 3, 3) == 1048000   # the stacked stand-in of configs[4]
    assert np.isclose(bench.HBM_PEAK_GBS, 8000.0)


def test_collective_double_builds_and_exports_what_the_engine_resolves():
    lib = os.path.join(HERE, "fake_rccl", "libfakerccl.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(HERE, "fake_rccl")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    h = ctypes.CDLL(lib)
    for sym in ("ncclGetVersion", "ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclAllGather",
                "ncclBroadcast", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):
        assert hasattr(h, sym), sym
    v = ctypes.c_int(0)
    assert h.ncclGetVersion(ctypes.byref(v)) == 0 and v.value == 99999    # (how the tests tell the double from librccl)
    # group bookkeeping without any device: an empty group closes cleanly, a stray ncclGroupEnd is refused
    assert h.ncclGroupStart() == 0 and h.ncclGroupEnd() == 0
    assert h.ncclGroupEnd() != 0


def test_last_line_is_compact():
    """the driver keeps an 8 KB tail of stdout and BENCH_r05's 19.5 KB line was lost to it.  The line builder on the
    verbose record of round 5 (profiles/r05/v_bench_final.json, a real default run) and on hostile inputs: under 4 KB, strict JSON, and it
    carries what the contract asks for (benchmarks/occu_spoccupancy.py:104-113 prints the number; so does this)."""
    import json

    with open(os.path.join(os.path.dirname(HERE), "profiles", "r05", "v_bench_final.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 8192                       # (the record that did not parse)
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert len(line) < bench.COMPACT_LIMIT == 4096 and "\n" not in line
    got = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))   # NaN / Infinity refused
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "secondary_summary"):
        assert k in got, k
    assert got["value"] == float(f"{full['value']:.7g}") and got["unit"] == "ESS/s" and got["vs_baseline"] is None
    assert got["config"]["workload"].startswith("biolith simulate(n_sites=10000") and "model" not in got["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "bytes_per_gradient_evaluation",
              "gradient_evaluations_per_launch", "us_per_leapfrog_per_chain", "latency_floor_us"):
        assert k in got["roofline"], k
    assert abs(got["roofline"]["frac"] - got["roofline"]["achieved"] / got["roofline"]["peak"]) < 1e-6
    assert set(got["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and len(got["cpu_baseline"]["sample"]) <= 120
    assert set(got["secondary_summary"]) == {w for w, _, _ in bench.SECONDARY}
    # hostile: paragraph-long strings everywhere, a non-finite figure, no cpu_baseline
    bad = json.loads(json.dumps(full))
    bad["metric"] = "m" * 5000
    bad["config"]["workload"] = "w" * 5000
    bad["config"]["gather"] = "g" * 5000
    bad["roofline"]["kernel"] = "k" * 5000
    bad["roofline"]["traffic"] = float("nan")
    bad["ms_per_step"] = float("inf")
    del bad["cpu_baseline"]
    bad["cpu_baseline_error"] = "e" * 5000
    line = bench.compact_line(bad, None)
    assert len(line) < 4096
    got = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert got["roofline"]["traffic"] is None and got["ms_per_step"] is None and len(got["cpu_baseline_error"]) <= 120


def test_rn_roofline_peak_and_executed_terms():
    """the transcendental peak is 8 lanes per SIMD per clock (8 issue cycles per wave64 v_exp / v_log / v_rcp,
    MI355X_MICROARCH.md:489), and the line reports what the kernel executes beside SURVEY section 8d's algorithmic 5.05 M terms."""
    assert bench.TRANS_PER_CU_PER_S == 32 * 2.4e9
    rng = np.random.default_rng(0)
    N, J = 40, 10
    data = dict(site_covs=rng.normal(size=(N, 3)), obs_covs=rng.normal(size=(N, 1, J, 3)),
                obs=(rng.random((1, N, 1, J)) < 0.4).astype(np.float64))
    ex, items = bench.rn_executed_terms(data, np.array([0.5, 0.1, -0.1, 0.2, 0.0, 0.3, -0.2, 0.1]), 3)
    assert ex == items * 8 * J and N <= items <= N * 13 and ex < N * J * 101
    cb = bench.committed_cpu_baseline(bench.RN_CPU_MEASURED)
    assert cb["scaled"] is False and cb["kind"] == "port" and cb["value"] > 0 and "NOT timed in this run" in cb["sample"]
