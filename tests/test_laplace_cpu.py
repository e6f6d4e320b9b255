"""Laplace coefficient priors in the oracle (the family biolith/utils/grid_search.py:366-371 tries besides Normal) and
grid_search_priors' argument handling -- no GPU."""
import numpy as np
import pytest

import oracle
from conftest import load_golden


@pytest.mark.parametrize("model,kw", [("occu", {}), ("occu_rn", {}), ("occu_fp", dict(fp_mode="unoccupied")),
                                      ("occu_re", dict(site_random_effects=True, obs_random_effects=True))])
@pytest.mark.parametrize("families", [("laplace", "laplace"), ("laplace", "normal")])
def test_laplace_prior_is_the_normal_model_with_the_prior_swapped(model, kw, families):
    g = load_golden("rn_small_2x2" if model == "occu_rn" else "small_3x3")
    pri = ((0.2, 0.7), (-0.1, 1.5))
    od_n = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], *pri, model=model, **kw)
    od_l = oracle.OracleData(g["site_covs"], g["obs_covs"], g["obs"], *pri, model=model, prior_family=families, **kw)
    th = np.random.default_rng(2).uniform(-1, 1, size=od_n.D)
    Un, Gn = od_n.potential_grad(th)
    Ul, Gl = od_l.potential_grad(th)
    Ks, Ko = od_n.Ks, od_n.Ko

    def logpdf(v, loc, scale, fam):
        return (-np.abs(v - loc) / scale - np.log(2 * scale)) if fam == "laplace" else (-0.5 * ((v - loc) / scale) ** 2 - np.log(scale) - 0.5 * np.log(2 * np.pi))

    def dlogpdf(v, loc, scale, fam):
        return -np.sign(v - loc) / scale if fam == "laplace" else -(v - loc) / scale ** 2

    want, dwant = Un, Gn.copy()
    for sl, p, fam in ((slice(0, Ks + 1), pri[0], families[0]), (slice(Ks + 1, Ks + Ko + 2), pri[1], families[1])):
        want += logpdf(th[sl], *p, "normal").sum() - logpdf(th[sl], *p, fam).sum()
        dwant[sl] += dlogpdf(th[sl], *p, "normal") - dlogpdf(th[sl], *p, fam)
    assert Ul == pytest.approx(want, rel=1e-12) and np.allclose(Gl, dwant, rtol=1e-10, atol=1e-10)
    if model == "occu":
        lit = oracle.literal_log_joint(th, g["site_covs"], g["obs_covs"], g["obs"], *pri, prior_family=families)
        assert Ul == pytest.approx(-lit, rel=1e-12)


def test_grid_search_rejects_unknown_families_before_any_fit():
    from biolith_amd.models import occu
    from biolith_amd.regression import LinearRegression
    from biolith_amd.utils import GridSearchResult, grid_search_priors

    g = load_golden("small_3x3")
    with pytest.raises(ValueError, match="Unsupported prior type"):
        grid_search_priors(occu, g["site_covs"], g["obs_covs"], g["obs"], LinearRegression, LinearRegression, prior_types=["cauchy"])
    assert GridSearchResult._fields == ("best_result", "best_params", "best_score", "cv_results")
