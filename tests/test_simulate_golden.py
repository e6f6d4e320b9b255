"""biolith_amd.models.simulate must be bit-identical to the reference's simulate()
(biolith/models/occu.py:245-430): fixtures in tests/golden were produced by the reference itself
(tests/golden/make_golden.py)."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden, quiet_simulate


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


CASES = ["default", "missing", "missing_3periods", "small_3x3", "seed7_2x1", "cfg2", "stacked", "bench_i3", "fp_constant",
         "fp_unoccupied"]


@pytest.mark.parametrize("name", CASES)
def test_simulate_matches_reference(golden_index, name):
    e = golden_index[name]
    data, truth, out = quiet_simulate(**e["kwargs"])
    for k in ("site_covs", "obs_covs", "obs"):
        assert list(data[k].shape) == e["shapes"][k]
        assert _sha(data[k]) == e["sha256"][k], k
    assert data["coords"] is None and data["ell"] == e["ell"] == 0.0
    assert _sha(truth["z"]) == e["sha256_z"]
    assert np.array_equal(truth["beta"], np.array(e["beta"])) and np.array_equal(truth["alpha"], np.array(e["alpha"]))
    assert out == e["stdout"]  # the two progress lines, occu.py:388-391
    if e["stored"]:
        g = load_golden(name)
        for k in ("site_covs", "obs_covs", "obs"):
            assert np.array_equal(data[k], g[k], equal_nan=True)


def test_simulate_survey_hashes(golden_index):
    """SURVEY.md Appendix C lists the first 24 hex digits independently."""
    assert golden_index["default"]["sha256"]["obs"].startswith("5a59a82c3a2d8d65e2968a26")
    assert golden_index["missing"]["sha256"]["site_covs"].startswith("59bf28765fc589ce2f307fd7")
    assert golden_index["cfg2"]["sha256"]["obs_covs"].startswith("e9beb9dc3d2abe6881e1f1d4")
    assert golden_index["stacked"]["sha256"]["obs"].startswith("dc28492b6fdddb9035e82726")


def test_simulate_defaults_are_52_visits():
    data, _, _ = quiet_simulate()
    assert data["obs"].shape == (1, 100, 1, 52)  # 365/7, occu.py:251-252,336


def test_simulate_spatial_not_built():
    with pytest.raises(NotImplementedError):
        quiet_simulate(spatial=True)
