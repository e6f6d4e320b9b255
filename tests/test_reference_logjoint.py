"""The oracle against potentials that the REFERENCE'S OWN model functions produced (tests/golden/reference_logjoint_*.json,
made by tests/golden/make_reference_logjoint.py: biolith/models/occu.py:136-242, occu_rn.py:123-222 + utils/distributions.py:6-40,
occu_cop.py:150-255, nmixture.py:150-220 executed under a functional NumPy shim of numpyro / jax).  This is what pins the oracle's
log-densities to the reference's model text rather than to the builder's reading of it.

Tolerance: |U_oracle - U_ref| <= 1e-10 |U_ref| (both float64 on the same float32-exact data and theta; measured <= 2e-15), the
analytic gradient within 1e-7 of the fixture's central differences (measured <= 8e-9).

Deliberate deviation, asserted rather than hidden: numpyro clamps a Bernoulli's probs into
[tiny_f32, 1 - eps_f32].  The oracle and the engine keep the clamp where it decides a value -- the z = 0 branch, where a detection
costs log tiny_f32 (every point of every occu fixture exercises that) -- and NOT at the top end, |logit| > 15.9 (p or psi within
1.2e-7 of one), where they state log sigma exactly.  The fixtures' fifth point ("clamp_regime") sits there on purpose: the oracle's
LITERAL statement with every clamp switched on reproduces the reference there, the closed form is larger (exact logs are more
negative than clamped ones), and the Royle-Nichols and N-mixture forms -- which follow both clamps -- still agree.
"""
import numpy as np
import pytest

import oracle
import reference_logjoint as R

CASES = R.case_names()


def test_the_fixture_set_covers_the_models_and_options_of_the_path():
    models = {R.load(c)["model"] for c in CASES}
    assert models == {"occu", "occu_rn", "occu_cop", "nmixture"}
    want = {"default", "missing", "missing_3periods", "small_3x3", "two_species", "fp_constant", "fp_unoccupied", "re_site", "re_obs",
            "re_both", "rn_default", "rn_missing", "cop_default", "nmix_ref_test", "priors_normal", "priors_laplace", "priors_fp", "priors_re",
            "priors_rn", "priors_cop", "priors_nmix"}
    assert want <= set(CASES)
    for c in CASES:
        e = R.load(c)
        assert [p["label"] for p in e["points"]] == ["uniform2_a", "uniform2_b", "uniform1", "near_truth", "clamp_regime"]
        assert e["clamp_dtype"] == "float32"


@pytest.mark.parametrize("case", CASES)
def test_oracle_potential_equals_the_reference_models(case):
    e = R.load(case)
    X, W, Y, kw = R.build(e)
    od = oracle.OracleData(X, W, Y, **kw)
    assert od.D == e["dims"]["D"]
    for p in e["points"][:4]:
        th = R.flat_theta(e, p["unconstrained"])
        U, g = od.potential_grad(th)
        assert abs(U - p["U"]) <= 1e-10 * abs(p["U"]), (case, p["label"], U, p["U"])
        if "grad_U_central_difference" in p:
            gf = R.flat_theta(e, p["grad_U_central_difference"])
            assert np.max(np.abs(g - gf)) <= 1e-7 * np.max(np.abs(gf)), (case, p["label"])


@pytest.mark.parametrize("case", CASES)
def test_clamp_regime_is_the_one_documented_deviation(case):
    e = R.load(case)
    X, W, Y, kw = R.build(e)
    od = oracle.OracleData(X, W, Y, **kw)
    p = e["points"][4]
    th = R.flat_theta(e, p["unconstrained"])
    U, _ = od.potential_grad(th)
    if e["model"] in ("occu_rn", "nmixture"):      # both clamps followed (rn) / no probability clamp in the model (Binomial)
        assert abs(U - p["U"]) <= 1e-10 * abs(p["U"]), (U, p["U"])
        return
    if e["model"] == "occu_cop":                   # only z ~ Bernoulli(psi) is clamped there: log(1 - psi) stops at log eps_f32 = -15.94
        d = e["dims"]
        assert abs(U - p["U"]) <= 3.0 * d["N"] * d["T"], (U, p["U"])      # psi's logit is 17.5 + O(1) at this point
        return
    assert U > p["U"] * 1.01, (U, p["U"])          # exact logs at the top end are more negative than clamped ones; far inside the regime
    if e["model"] == "occu" and e["dims"]["S"] == 1 and not kw["site_random_effects"] and not kw["obs_random_effects"]:
        # the literal statement with numpyro's clamp in both branches IS the reference there
        if kw["model"] == "occu_fp":
            lj = oracle.literal_log_joint_fp(th, X, W, Y, fp_mode=kw["fp_mode"], prior_fp=kw.get("prior_fp", (2.0, 5.0)),
                                             prior_beta=kw["prior_beta"], prior_alpha=kw["prior_alpha"], clamp_z1=True)
        else:
            lj = oracle.literal_log_joint(th, X, W, Y, kw["prior_beta"], kw["prior_alpha"], clamp_z1=True, prior_family=kw["prior_family"])
        assert abs(-lj - p["U"]) <= 1e-10 * abs(p["U"]), (-lj, p["U"])


def test_enumerated_deterministic_site_carries_the_enumeration_axis():
    """DESIGN.md section 3, upstream assumption (ii): under parallel enumeration ``prob_detection_fp`` (occu.py:229-235) is
    computed from z's enumerated value, so it has z's axis in front of the plates; psi and prob_detection do not."""
    e = R.load("fp_constant")
    d, sh = e["dims"], e["points"][0]["deterministic_shapes"]
    assert sh["prob_detection_fp"] == [2, d["J"], d["T"], d["N"], d["S"]]
    assert sh["prob_detection"] == [d["J"], d["T"], d["N"], d["S"]] and sh["psi"] == [d["N"], d["S"]]
