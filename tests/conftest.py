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
