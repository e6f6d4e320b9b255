"""False-positive occupancy model (biolith/models/occu.py:146-157, 229-241) -- oracle side, no GPU:
the C oracle's branch-by-branch potential against the literal NumPy model statement and against
finite differences, for both options and non-default Beta priors."""
import numpy as np
import pytest

import oracle
from conftest import load_golden


@pytest.mark.parametrize("name,mode,prior", [("fp_constant", "constant", (2.0, 5.0)), ("fp_unoccupied", "unoccupied", (2.0, 5.0)),
                                             ("fp_unoccupied", "constant", (1.5, 9.0)), ("missing", "unoccupied", (0.7, 3.0))])
def test_fp_potential_equals_literal_model_and_fd(name, mode, prior):
    g = load_golden(name)
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_fp", fp_mode=mode, prior_fp=prior)
    assert od.D == g["site_covs"].shape[1] + g["obs_covs"].shape[3] + 3
    rng = np.random.default_rng(5)
    for _ in range(3):
        th = rng.uniform(-1.5, 1.5, size=od.D)
        U, G = od.potential_grad(th)
        lit = oracle.literal_log_joint_fp(th, g["site_covs"], g["obs_covs"], g["obs"], fp_mode=mode, prior_fp=prior)
        assert U == pytest.approx(-lit, rel=1e-12, abs=1e-9)
        h = 1e-6
        fd = np.array([(od.potential_grad(th + h * e)[0] - od.potential_grad(th - h * e)[0]) / (2 * h) for e in np.eye(od.D)])
        assert np.max(np.abs(fd - G)) <= 1e-6 * max(1.0, np.max(np.abs(G)))


def test_fp_rate_to_zero_recovers_the_plain_model():
    # f -> 0: the z=1 branch tends to the plain model's and the z=0 branch forbids detections, so for a
    # dataset WITHOUT detections at unoccupied-looking sites the likelihood parts agree up to O(f)
    g = load_golden("small_3x3")
    od0 = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"])
    odf = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_fp", fp_mode="constant", prior_fp=(1.0, 1.0))
    th = np.random.default_rng(0).uniform(-1, 1, size=od0.D)
    phi = -30.0                                    # f = 9.4e-14
    Uf, _ = odf.potential_grad(np.append(th, phi))
    U0, _ = od0.potential_grad(th)
    # Beta(1,1) log-density is 0, Jacobian log f + log(1-f) = phi - 2 softplus(phi) ~ phi; the z=0 branch
    # of a site-period with n_det detections is f^n_det instead of tiny^n_det: negligible either way
    assert Uf + phi == pytest.approx(U0, abs=1e-6 * abs(U0))


def test_oracle_nuts_samples_the_rate():
    g = load_golden("fp_constant")
    od = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], model="occu_fp", fp_mode="constant")
    o = oracle.nuts_run(od, 150, 150, num_chains=2, seed=1)
    assert o["draws"].shape == (2, 150, 5)
    rate = 1 / (1 + np.exp(-o["draws"][..., -1]))
    assert abs(rate.mean() - 0.1) < 0.05            # simulated with prob_fp_constant = 0.1 (occu.py:495-508)
