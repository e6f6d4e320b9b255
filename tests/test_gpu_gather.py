"""The gather of the draws over RCCL behind the C-ABI (bl_comm_*, bl_gather_draws; include/biolith_hip.h).

Reference counterpart: chain_method="parallel" + the implicit gather of mcmc.get_samples() (biolith/utils/fit.py:109-113,
132).  On a one-GPU box the communicator has one rank; what is checked is that the result that went through librccl is
bit-equal to the plain device->host copy, for the single launch, for `fit(devices=[0])`, and -- config 3's analogue --
for 8 chains at 10k x 5 dealt as one launch, as chain_offset shards and through the gather."""
import json
import os

import numpy as np
import pytest

from biolith_amd import _ffi
from biolith_amd.distributed import RcclComm, comm_from_env, comms_for_devices, gather_draws, rccl_version, shard_chains
from biolith_amd.engine import OccuDataset
from biolith_amd.evaluation import effective_sample_size, split_gelman_rubin
from biolith_amd.models import occu
from biolith_amd.utils import fit
from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

FIELDS = ("draws", "diverging", "num_steps", "accept_prob", "potential_energy", "step_size", "inv_mass", "n_leapfrog")


def _same(a, b):
    return all(np.array_equal(getattr(a, f), getattr(b, f)) for f in FIELDS)


def test_one_rank_communicator_returns_the_fetch_bit_for_bit():
    assert rccl_version() >= 21000
    g = load_golden("small_3x3")
    ds = OccuDataset(g["site_covs"], g["obs_covs"], g["obs"])
    comm = comm_from_env(0, rank=0, world=1)
    assert (comm.world, comm.rank, comm.device) == (1, 0, 0) and comm.init_ms > 0
    for chains in (1, 3):
        ds.launch(num_warmup=40, num_samples=30, num_chains=chains, seed=2)
        ds.wait()
        plain = ds.fetch()
        via = gather_draws([comm], [ds], [chains])
        assert via.draws.shape == (chains, 30, ds.D) and _same(plain, via)
    # argument checks of the C-ABI: wrong chain count, launch not finished
    with pytest.raises(ValueError, match="chains_per_rank says"):
        gather_draws([comm], [ds], [2])
    ds.launch(num_warmup=400, num_samples=400, num_chains=2, seed=2)
    with pytest.raises(_ffi.EngineError, match="no finished NUTS launch"):
        gather_draws([comm], [ds], [2])
    ds.wait()
    comm.close()


def test_init_all_rejects_a_device_named_twice_and_out_of_range():
    with pytest.raises(ValueError, match="named twice"):
        comms_for_devices([0, 0])
    with pytest.raises(ValueError, match="out of range"):
        comms_for_devices([63])
    (c,) = comms_for_devices([0])
    assert c.world == 1
    c.close()


def test_fit_devices_goes_through_the_communicator_and_equals_host_concat():
    g = load_golden("missing")
    data = dict(site_covs=g["site_covs"], obs_covs=g["obs_covs"], obs=g["obs"])
    kw = dict(num_chains=3, num_warmup=60, num_samples=40, random_seed=4)
    plain = fit(occu, **data, **kw)                       # one launch, bl_nuts_fetch
    via = fit(occu, **data, **kw, devices=[0])            # bl_comm_init_all + bl_gather_draws
    concat = fit(occu, **data, **kw, devices=[0, 0])      # a device named twice: shards fetched and concatenated on the host
    assert via.mcmc.result.comm_init_ms > 0 and plain.mcmc.result.comm_init_ms == 0
    for other in (via, concat):
        assert _same(plain.mcmc.result, other.mcmc.result)
        for k in plain.samples:
            assert np.array_equal(plain.samples[k], other.samples[k]), k


def test_config3_analogue_eight_chains_one_launch_vs_shards_vs_gather(cfg2_data):
    """BASELINE.json configs[2] (8 chains, one per GPU, RCCL gather) needs an 8-GPU node; its single-GPU analogue: the same 8
    chains at 10 000 x 5 as ONE launch, as eight chain_offset shards (what rank r of 8 runs), and the shard that went through
    bl_gather_draws -- all bit-equal -- and the posterior against the oracle's captured one."""
    data, truth = cfg2_data
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])
    one = ds.nuts(num_warmup=1000, num_samples=1000, num_chains=8, seed=0)
    assert one.lds_staged and one.diverging.sum() == 0 and one.chains_l2_local == 8
    comm = comm_from_env(0, rank=0, world=1)
    for r in range(8):
        count, first = shard_chains(8, 8, r)
        ds.launch(num_warmup=1000, num_samples=1000, num_chains=count, seed=0, chain_offset=first)
        ds.wait()
        shard = gather_draws([comm], [ds], [count]) if r % 2 else ds.fetch()
        assert np.array_equal(shard.draws, one.draws[first: first + count]), r
        assert np.array_equal(shard.num_steps, one.num_steps[first: first + count])
    comm.close()
    via_fit = fit(occu, **data, num_chains=8, devices=[0] * 8)        # eight launches of one chain, host concat
    assert np.array_equal(via_fit.mcmc.result.draws, one.draws)
    fx = json.load(open(os.path.join(GOLDEN, "oracle_posterior_cfg2.json")))
    flat = one.draws.reshape(-1, 8).astype(np.float64)
    mcse = np.sqrt(flat.var(0) / effective_sample_size(one.draws) + np.array(fx["sd"]) ** 2 / np.array(fx["ess"]))
    assert np.all(np.abs(flat.mean(0) - fx["mean"]) <= 4 * mcse)
    assert np.all(np.abs(flat.std(0) / fx["sd"] - 1) < 0.1)
    assert split_gelman_rubin(one.draws).max() < 1.01
    assert abs(float(via_fit.samples["psi"].mean()) - truth["z"].mean()) < 0.1     # occu.py:440


def test_config3_as_worded_one_chain_per_rank_through_the_bench_shard(cfg2_data):
    """BASELINE.json configs[2] verbatim is `bench.py --gpus 8 --chains-per-gpu 1`: rank r runs ONE chain, global id r.  On one GPU:
    the shard bench.py's `rank_shard` hands rank r of 8 (what its `one_step` launches with), for r = 0..7, against the 8-chain
    launch -- bit-equal -- and the flag's plumbing (`workload_chains`)."""
    import bench

    wl = bench.WORKLOADS["occu"]
    assert bench.workload_chains(wl) == 4 and bench.workload_chains(wl, 1) == 1 and bench.workload_chains(bench.WORKLOADS["occu_cfg1"]) == 2
    assert bench.parse_args(["--gpus", "8", "--chains-per-gpu", "1"]).chains_per_gpu == 1
    data, _ = cfg2_data
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])
    W = S = 300
    one = ds.nuts(num_warmup=W, num_samples=S, num_chains=8, seed=0)
    for r in range(8):
        shard = bench.rank_shard(r, bench.workload_chains(wl, 1))
        assert shard == dict(num_chains=1, chain_offset=r)
        got = ds.nuts(num_warmup=W, num_samples=S, seed=0, **shard)
        assert np.array_equal(got.draws[0], one.draws[r]) and np.array_equal(got.num_steps[0], one.num_steps[r]), r
    # (4 chains per GPU, the default: rank 1 of 2 runs chains 4..7)
    got = ds.nuts(num_warmup=W, num_samples=S, seed=0, **bench.rank_shard(1, 4))
    assert np.array_equal(got.draws, one.draws[4:8])
