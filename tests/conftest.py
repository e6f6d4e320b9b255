import contextlib
import io
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The engine library is a build product (git-ignored): build it once if a fresh checkout lacks it, so that the
    ABI tests test the library rather than its absence.  hipcc cross-compiles gfx950 without a GPU (about two minutes)."""
    lib = os.path.join(ROOT, "biolith_amd", "lib", "libbiolith_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess

        subprocess.run(["make", "-C", os.path.join(ROOT, "biolith_amd", "csrc"), "-j8"], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1800)


@pytest.fixture(scope="session")
def golden_index():
    with open(os.path.join(GOLDEN, "simulate_index.json")) as f:
        return json.load(f)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, f"simulate_{name}.npz"))
    return {k: z[k] for k in z.files}


def quiet_simulate(**kw):
    from biolith_amd.models import simulate

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        data, truth = simulate(**kw)
    return data, truth, buf.getvalue()


CFG2 = dict(n_sites=10000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7)


@pytest.fixture(scope="session")
def cfg2_data():
    data, truth, _ = quiet_simulate(**CFG2)
    return data, truth


# ---- HIP NUTS vs CPU-oracle NUTS on independent streams, >= 4 chains x 1000 draws ----
PARITY_W, PARITY_S = 500, 2000          # warmup / draws per chain of the posterior-parity tests (4 chains each side)


def load_oracle_draws(name, D, warmup, samples):
    """(4, samples, kept) float64 draws of the oracle's NUTS from tests/golden/oracle_draws_<name>.npz (made by
    tests/golden/make_oracle_posterior_draws.py with the same data set, model arguments, warm-up and draw counts -- checked here)."""
    z = np.load(os.path.join(GOLDEN, f"oracle_draws_{name}.npz"))
    assert int(z["D"]) == D and int(z["warmup"]) == warmup and int(z["samples"]) == samples and int(z["divergences"]) == 0, name
    return z["draws"].astype(np.float64)


def posterior_parity(draws_gpu, draws_orc, ess_gpu=None):
    """|mean_gpu - mean_oracle| <= 4 MCSE, 0.9 <= sd ratio <= 1.1, split R-hat < 1.01 (both sides) -- the tolerances SURVEY.md
    section 8c states; the runs are long enough (4 x 2000 draws) for them to hold with room."""
    import oracle
    from biolith_amd.evaluation import effective_sample_size, split_gelman_rubin

    D = draws_gpu.shape[-1]
    fg, fo = draws_gpu.reshape(-1, D).astype(np.float64), np.asarray(draws_orc).reshape(-1, D)
    ess_g = effective_sample_size(draws_gpu) if ess_gpu is None else ess_gpu
    mcse = np.sqrt(fg.var(0) / ess_g + fo.var(0) / oracle.effective_sample_size(draws_orc))
    assert np.all(np.abs(fg.mean(0) - fo.mean(0)) <= 4 * mcse), (fg.mean(0) - fo.mean(0), mcse)
    ratio = fg.std(0) / fo.std(0)
    assert np.all((ratio > 0.9) & (ratio < 1.1)), ratio
    assert split_gelman_rubin(draws_gpu).max() < 1.01, split_gelman_rubin(draws_gpu)
    assert oracle.split_gelman_rubin(np.asarray(draws_orc)).max() < 1.01
