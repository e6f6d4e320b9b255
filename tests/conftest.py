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
