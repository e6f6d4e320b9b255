"""The sampler's per-form instantiations (nuts_kernel.hpp: JSEL = one visits-per-period form, LEAN = one species / one period / one
pair per lane / one-batch poll, the lean lane-group and Royle-Nichols forms; kernels_inst.hip picks by the launch's geometry) against
the GENERAL kernel, which carries every form at run time (BIOLITH_HIP_GENERAL=1): the same arithmetic in the same order, so draws, trees,
step sizes and metrics must be equal bit for bit (the parity gates against the float64 oracle are statistical past the first
transitions; this one is exact).  Likewise nmixture's table in LDS against the table in L2 (BIOLITH_HIP_NMIX_LDS=0)."""
import numpy as np
import pytest

from biolith_amd.engine import OccuDataset
from conftest import load_golden, quiet_simulate

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert np.array_equal(a.draws, b.draws) and np.array_equal(a.num_steps, b.num_steps)
    assert np.array_equal(a.step_size, b.step_size) and np.array_equal(a.inv_mass, b.inv_mass)
    assert np.array_equal(a.accept_prob, b.accept_prob) and np.array_equal(a.diverging, b.diverging)


def _ab(ds, monkeypatch, var, value, **kw):
    monkeypatch.delenv(var, raising=False)
    a = ds.nuts(**kw)
    monkeypatch.setenv(var, value)
    b = ds.nuts(**kw)
    monkeypatch.delenv(var, raising=False)
    return a, b, a.kernel_name, b.kernel_name


def _dyn():
    import contextlib
    import io

    from biolith_amd.models import simulate_dyn

    with contextlib.redirect_stdout(io.StringIO()):
        return simulate_dyn(n_sites=400, n_periods=8, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=28, session_duration=7, random_seed=4)[0]


CASES = {
    # one pair per lane, J = 5 form, lean (the headline's kernel at a tenth of its size)
    "headline_like": lambda: (quiet_simulate(n_sites=1000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=35, session_duration=7)[0], {}),
    # run-time J (7 visits: no unrolled form), lean
    "seven_visits": lambda: (quiet_simulate(n_sites=150, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=49, random_seed=11)[0], {}),
    # lane groups, several workgroups, one period (the lean one-period group form)
    "grid_row_4": lambda: (quiet_simulate(n_site_covs=2, n_obs_covs=1, n_sites=1600, deployment_days_per_site=7 * 32, session_duration=7, random_seed=46)[0], {}),
    # lane groups over periods (the lean group form)
    "stacked": lambda: (quiet_simulate(n_sites=600, n_periods=4, n_site_covs=2, n_obs_covs=3, deployment_days_per_site=42, session_duration=7, random_seed=3)[0], {}),
    # one period per lane at four visits each (the lean own-period group form: BASELINE configs[4]'s stand-in shape)
    "stacked_8x4": lambda: (quiet_simulate(n_sites=500, n_periods=8, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=28, session_duration=7, random_seed=2)[0], {}),
    # dynamic occupancy, eight periods on eight lanes at four visits (compile-time) against the same two-scans form with run-time counts
    "dyn_8x4": lambda: (_dyn(), dict(model="occu_dyn")),
    # Royle-Nichols: the lean J <= 10 form
    "rn": lambda: (load_golden("rn_small_2x2"), dict(model="occu_rn")),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_per_form_kernels_equal_the_general_kernel_bit_for_bit(case, monkeypatch):
    d, kw = CASES[case]()
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], **kw)
    a, b, na, nb = _ab(ds, monkeypatch, "BIOLITH_HIP_GENERAL", "1", num_warmup=150, num_samples=100, num_chains=2, seed=5)
    assert na != nb and (nb.rstrip().endswith(", -1, false>") or (case == "dyn_8x4" and nb.rstrip().endswith(", 1, false>"))), (na, nb)   # a per-form instantiation, then the general one
    _same(a, b)
    ds.close()


def test_nmixture_table_in_lds_or_in_l2_changes_no_bit(monkeypatch):
    d = load_golden("nmix_ref_test")
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="nmixture")
    a, b, _, _ = _ab(ds, monkeypatch, "BIOLITH_HIP_NMIX_LDS", "0", num_warmup=100, num_samples=60, num_chains=2, seed=2)
    _same(a, b)
    ds.close()
