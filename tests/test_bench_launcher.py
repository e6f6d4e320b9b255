"""bench.py --gpus N must really run N ranks, or fail loudly (VERDICT r01 "Missing #1"; the reference's counterpart is
chain_method="parallel", biolith/utils/fit.py:109-113).  CPU-only: the launcher and the rendezvous are driven on gloo with
BENCH_SELFTEST=1 (no GPU work, the printed line is not a bench line)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=280):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(300)
def test_launcher_starts_two_ranks_on_gloo():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"BENCH_SELFTEST": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["selftest"] is True and line["n_gpus"] == 2 and line["world"] == 2
    assert line["rank_sum"] == 1.0          # ranks 0 and 1 both took part in the all-reduce
    assert line["launcher"] == "bench.py"


@pytest.mark.timeout(300)
def test_more_ranks_than_gpus_is_a_loud_failure():
    """No GPU in this container: asking for 2 must not quietly run 1 (or 0)."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr and "failed" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_match_gpus_flag():
    r = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "BENCH_SELFTEST": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_ranks_on_a_one_gpu_box_fail_loudly():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], timeout=560)
    assert r.returncode != 0 and "only 1 GPU(s) visible" in r.stderr
