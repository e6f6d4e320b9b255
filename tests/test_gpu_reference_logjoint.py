"""``bl_logp_grad`` (the C-ABI parity hook of K1) against potentials that the REFERENCE'S OWN model functions produced:
tests/golden/reference_logjoint_*.json (biolith/models/occu.py:136-242, occu_rn.py:123-222, occu_cop.py:150-255,
nmixture.py:150-220 executed under a functional NumPy shim; tests/golden/make_reference_logjoint.py).  No oracle in the
comparison of U: HIP float32 per-term arithmetic against the reference's model text in float64, at DESIGN.md section 3's
tolerances -- occu, false positives 1e-6 / 1e-5; occu_cop, nmixture, random effects 2e-6 / 2e-5; occu_rn 1e-5 / 1e-4.  Gradients
against the fixture's central differences where it has them (D <= 24), else against the oracle's analytic gradient (which
tests/test_reference_logjoint.py checks against those central differences on the small cases)."""
import numpy as np
import pytest

import oracle
import reference_logjoint as R
from biolith_amd.engine import OccuDataset

pytestmark = pytest.mark.gpu


def _tol(e, kw):
    if e["model"] == "occu_rn":
        return 1e-5, 1e-4
    if e["model"] in ("occu_cop", "nmixture") or kw["site_random_effects"] or kw["obs_random_effects"]:
        return 2e-6, 2e-5
    return 1e-6, 1e-5


@pytest.mark.parametrize("case", R.case_names())
def test_engine_potential_equals_the_reference_models(case):
    e = R.load(case)
    X, W, Y, kw = R.build(e)
    ds = OccuDataset(X, W, Y, **R.engine_kwargs(kw))
    assert ds.D == e["dims"]["D"]
    u_tol, g_tol = _tol(e, kw)
    pts = e["points"][:4]
    th = np.stack([R.flat_theta(e, p["unconstrained"]) for p in pts])
    U, G = ds.logp_grad(th)
    Uref = np.array([p["U"] for p in pts])
    assert np.max(np.abs(U - Uref) / np.abs(Uref)) <= u_tol, (case, U, Uref)
    if "grad_U_central_difference" in pts[0]:
        Gref = np.stack([R.flat_theta(e, p["grad_U_central_difference"]) for p in pts])
    else:
        Gref = oracle.OracleData(X, W, Y, **kw).potential_grad(th)[1]
    assert np.max(np.abs(G - Gref) / np.max(np.abs(Gref), axis=1, keepdims=True)) <= g_tol, (case, np.abs(G - Gref).max(1))


@pytest.mark.parametrize("case", ["default", "small_3x3", "rn_default", "nmix_ref_test"])
def test_engine_in_the_clamp_regime(case):
    """The fifth point of a fixture: detection probabilities within eps_f32 of one.  Royle-Nichols and N-mixture follow the
    reference there; occu states log sigma exactly where numpyro clamps (the documented deviation: the engine equals the
    ORACLE there, which tests/test_reference_logjoint.py relates to the reference)."""
    e = R.load(case)
    X, W, Y, kw = R.build(e)
    ds, p = OccuDataset(X, W, Y, **R.engine_kwargs(kw)), e["points"][4]
    th = R.flat_theta(e, p["unconstrained"])[None]
    U = ds.logp_grad(th)[0][0]
    want = p["U"] if e["model"] in ("occu_rn", "nmixture") else oracle.OracleData(X, W, Y, **kw).potential_grad(th[0])[0]
    assert abs(U - want) <= _tol(e, kw)[0] * abs(want), (U, want)
