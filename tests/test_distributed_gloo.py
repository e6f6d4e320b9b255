"""N>1 path on CPU: world_size-2 gloo.  Each rank runs its shard of chains (selected by
chain_offset, as one-process-per-GPU does with the HIP engine; here the oracle stands in for the
sampler because no GPU exists in this container) and the draws are all-gathered.  The gathered
posterior must equal a single-process run of all chains, in chain order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from biolith_amd.distributed import gather_host_arrays, shard_chains
from conftest import ROOT, load_golden


def test_shard_chains_partitions_exactly():
    for n in (1, 4, 5, 8, 13):
        for w in (1, 2, 3, 8):
            got = [shard_chains(n, w, r) for r in range(w)]
            assert sum(c for c, _ in got) == n
            assert [o for _, o in got] == list(np.cumsum([0] + [c for c, _ in got[:-1]]))
    with pytest.raises(ValueError):
        shard_chains(4, 2, 2)


def _worker(rank, world, port, total_chains, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle

    g = load_golden("seed7_2x1")
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"])
    count, offset = shard_chains(total_chains, world, rank)
    r = oracle.nuts_run(od, 30, 25, num_chains=count, seed=3, chain_offset=offset, threads=1)
    local = torch.from_numpy(r["draws"].astype(np.float32))
    full = gather_host_arrays(local)
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_equals_single_process():
    import oracle

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total = 3  # uneven shards: 2 + 1
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = load_golden("seed7_2x1")
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"])
    single = oracle.nuts_run(od, 30, 25, num_chains=total, seed=3)["draws"].astype(np.float32)
    assert gathered.shape == (3, 25, od.D)
    assert np.array_equal(gathered, single)


def test_gather_is_identity_without_process_group():
    t = torch.arange(6.0).reshape(1, 2, 3)
    assert gather_host_arrays(t) is t
