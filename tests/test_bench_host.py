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
