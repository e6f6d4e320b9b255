#!/usr/bin/env python3
"""The several-ranks branch of bl_gather_draws on ONE GPU, with tests/fake_rccl/libfakerccl.so standing in for librccl.

Runs in a process of its own (the engine resolves its collective library once per process): tests/test_gpu_fake_rccl_world.py starts it
with BIOLITH_RCCL_LIB set and reads the JSON it prints.  A DOUBLE of the collective, not RCCL: what is exercised is the engine's side --
block offsets, the all-gather and its "v" form (grouped broadcasts), stream / event ordering behind the launches, want_result=False,
the group closed on an error.  Reference: chain_method="parallel" + mcmc.get_samples() (biolith/utils/fit.py:109-113, 132).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from biolith_amd import _ffi  # noqa: E402
from biolith_amd.distributed import comms_for_devices, gather_draws, rccl_version, shard_chains  # noqa: E402
from biolith_amd.engine import OccuDataset  # noqa: E402
from biolith_amd.models import occu  # noqa: E402
from biolith_amd.utils import fit  # noqa: E402

FIELDS = ("draws", "diverging", "num_steps", "accept_prob", "potential_energy", "step_size", "inv_mass", "n_leapfrog")


def same(a, b):
    return all(np.array_equal(getattr(a, f), getattr(b, f)) for f in FIELDS)


def main():
    out = {"version": rccl_version(), "worlds": [], "fit": []}
    z = np.load(os.path.join(ROOT, "tests", "golden", "simulate_small_3x3.npz"))
    data = dict(site_covs=z["site_covs"], obs_covs=z["obs_covs"], obs=z["obs"])
    W, S = 40, 30
    make = lambda: OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])  # noqa: E731
    ds_all = make()
    for world, chains in [(2, 4), (2, 3), (3, 3), (3, 7), (8, 8), (8, 11), (8, 16)]:
        full = ds_all.nuts(num_warmup=W, num_samples=S, num_chains=chains, seed=5)
        deal = [shard_chains(chains, world, r) for r in range(world)]
        dss = [make() for _ in range(world)]
        comms = comms_for_devices([0] * world)
        for ds, (c, first) in zip(dss, deal):
            ds.launch(num_warmup=W, num_samples=S, num_chains=c, seed=5, chain_offset=first)
        # (no wait on the host for the later ranks' kernels before the gather is queued: bl_gather_draws orders each rank's
        # contribution behind that rank's launch with an event; only "finished" is required of the handle)
        for ds in dss:
            ds.wait()
        counts = [c for c, _ in deal]
        via = gather_draws(comms, dss, counts)
        none = gather_draws(comms, dss, counts, want_result=False)       # what a non-root rank asks for
        again = gather_draws(comms[::-1], dss[::-1], counts)              # the local ranks in another order: offsets come from the RANK
        rec = dict(world=world, chains=chains, counts=counts, equal_blocks=len(set(counts)) == 1, bit_equal=bool(same(full, via)),
                   no_result_is_none=none is None, reversed_bit_equal=bool(same(full, again)),
                   shape=list(via.draws.shape))
        if world == 3 and chains == 7:
            # an error inside the group: the call fails, the thread stays usable, the next gather is right
            os.environ["FAKE_RCCL_FAIL_CALL"] = "2"
            try:
                gather_draws(comms, dss, counts)
                rec["injected"] = "no error raised"
            except _ffi.EngineError as exc:
                rec["injected"] = str(exc)
            os.environ.pop("FAKE_RCCL_FAIL_CALL", None)
            rec["after_injected_bit_equal"] = bool(same(full, gather_draws(comms, dss, counts)))
            # a rank whose launch is still in flight is refused before anything is posted
            dss[1].launch(num_warmup=400, num_samples=400, num_chains=counts[1], seed=5, chain_offset=deal[1][1])
            try:
                gather_draws(comms, dss, counts)
                rec["in_flight"] = "no error raised"
            except _ffi.EngineError as exc:
                rec["in_flight"] = str(exc)
            dss[1].wait()
        out["worlds"].append(rec)
        for c in comms:
            c.close()
        for ds in dss:
            ds.close()
    # fit(devices=[0, 0, ...]): the same through the reference's entry point
    for world, chains in [(2, 4), (3, 5), (8, 8), (4, 3)]:
        kw = dict(num_chains=chains, num_warmup=W, num_samples=S, random_seed=4)
        plain = fit(occu, **data, **kw)
        via = fit(occu, **data, **kw, devices=[0] * world)
        ok = same(plain.mcmc.result, via.mcmc.result) and all(np.array_equal(plain.samples[k], via.samples[k]) for k in plain.samples)
        out["fit"].append(dict(world=world, chains=chains, bit_equal=bool(ok), comm_init_ms=float(via.mcmc.result.comm_init_ms)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
