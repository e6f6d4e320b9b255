"""The `world > 1` branch of bl_gather_draws (csrc/comm_rccl.hpp) on a one-GPU box, with a TEST DOUBLE of the collective.

RCCL refuses two ranks on one device and the pool's boxes have one GPU, so that branch -- the grouped ncclAllGather over several
communicators, its "v" form as grouped ncclBroadcasts when the chains do not divide evenly, the block offsets, the ordering of a rank's
contribution behind its own launch, want_result=False -- had never executed anywhere before the driver's 8-GPU run.  Here
tests/fake_rccl/libfakerccl.so (ten nccl* symbols; device-to-device copies between the ranks' buffers at ncclGroupEnd) is loaded through
the engine's BIOLITH_RCCL_LIB override in a process of its own.  It is a double, NOT RCCL: nothing here says anything about rings, xGMI
or multi-process rendezvous.  Reference: chain_method="parallel" and the gather of mcmc.get_samples() (biolith/utils/fit.py:109-113, 132)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FAKE = os.path.join(HERE, "fake_rccl", "libfakerccl.so")


@pytest.fixture(scope="module")
def world_run():
    if not os.path.exists(FAKE):
        subprocess.run(["make", "-C", os.path.join(HERE, "fake_rccl")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, BIOLITH_RCCL_LIB=FAKE, BIOLITH_TEST_ALLOW_DUP_DEVICES="1")
    env.pop("FAKE_RCCL_FAIL_CALL", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "fake_rccl", "run_world.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_the_double_is_what_was_loaded(world_run):
    assert world_run["version"] == 99999


def test_gathered_chains_equal_the_single_launch_for_every_world(world_run):
    seen = set()
    for rec in world_run["worlds"]:
        assert rec["bit_equal"] and rec["reversed_bit_equal"] and rec["no_result_is_none"], rec
        assert rec["shape"][0] == rec["chains"]
        seen.add((rec["world"], rec["equal_blocks"]))
    # worlds 2, 3, 8, each with equal blocks (one ncclAllGather per rank) and with unequal ones (world ncclBroadcasts per rank)
    assert {(2, True), (2, False), (3, True), (3, False), (8, True), (8, False)} <= seen


def test_an_error_inside_the_group_leaves_the_thread_usable(world_run):
    rec = next(r for r in world_run["worlds"] if "injected" in r)
    assert "injected failure" in rec["injected"] and "ncclBroadcast" in rec["injected"], rec["injected"]
    assert rec["after_injected_bit_equal"]
    assert "no finished NUTS launch" in rec["in_flight"]


def test_fit_with_devices_goes_through_the_gather(world_run):
    for rec in world_run["fit"]:
        assert rec["bit_equal"] and rec["comm_init_ms"] > 0, rec
