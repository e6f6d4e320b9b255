#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by IMPORTING the reference (build container only).

The reference's model/inference half needs jax/numpyro/funsor, which are not installed and cannot
be.  Its simulators are pure NumPy, so this script serves MagicMock modules
for those three roots at import time, imports ``biolith.models`` from /root/reference and records
the outputs of ``simulate()`` (biolith/models/occu.py:245-430).  Only DATA is written: inputs
(kwargs) and outputs (arrays or their SHA-256).  Nothing of the reference's source travels.

Run:  python tests/golden/make_golden.py        (needs /root/reference; not run on the GPU box)
"""
import contextlib
import hashlib
import importlib.abc
import importlib.machinery
import io
import json
import os
import sys
from unittest import mock

import numpy as np

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    roots = ("jax", "numpyro", "funsor")

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__path__ = []
        m.__spec__ = spec
        return m

    def exec_module(self, module):
        pass


def load_reference_simulate():
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REFERENCE)
    import biolith.models  # noqa: F401
    return sys.modules["biolith.models.occu"].simulate


def load_reference_simulate_rn():
    return sys.modules["biolith.models.occu_rn"].simulate_rn


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


# name -> kwargs; "store" = keep full arrays (small), else hashes + moments only
CASES = {
    "default": dict(store=True, kw=dict()),
    "missing": dict(store=True, kw=dict(simulate_missing=True)),
    "missing_3periods": dict(store=True, kw=dict(simulate_missing=True, n_periods=3)),
    "small_3x3": dict(store=True, kw=dict(n_sites=300, n_site_covs=3, n_obs_covs=3,
                                          deployment_days_per_site=35, session_duration=7)),
    "seed7_2x1": dict(store=True, kw=dict(n_sites=64, n_site_covs=2, n_obs_covs=1, random_seed=7,
                                          deployment_days_per_site=56)),
    "cfg2": dict(store=False, kw=dict(n_sites=10000, n_site_covs=3, n_obs_covs=3,
                                      deployment_days_per_site=35, session_duration=7)),
    "stacked": dict(store=False, kw=dict(n_sites=2000, n_periods=8, n_site_covs=3, n_obs_covs=3,
                                         deployment_days_per_site=28, session_duration=7)),
    "fp_constant": dict(store=True, kw=dict(simulate_missing=True, prob_fp_constant=0.1)),   # occu.py:495-499
    "fp_unoccupied": dict(store=True, kw=dict(n_sites=150, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=49,
                                              prob_fp_unoccupied=0.08, random_seed=11)),
    "bench_i3": dict(store=False, kw=dict(n_sites=800, n_site_covs=2, n_obs_covs=1, random_seed=45,
                                          deployment_days_per_site=23 * 7, session_duration=7)),
}


# Royle-Nichols generator (biolith/models/occu_rn.py:225-358)
RN_CASES = {
    "rn_default": dict(store=True, kw=dict()),
    "rn_small_2x2": dict(store=True, kw=dict(n_sites=60, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=42, random_seed=3)),
    "rn_missing": dict(store=True, kw=dict(n_sites=50, simulate_missing=True, deployment_days_per_site=56, n_periods=2, random_seed=1)),
    "rn_cfg4": dict(store=False, kw=dict(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7)),
}


# count occupancy generator (biolith/models/occu_cop.py:258-396)
COP_CASES = {
    "cop_default": dict(store=True, kw=dict()),
    "cop_missing": dict(store=True, kw=dict(simulate_missing=True)),                       # occu_cop.py:399-400
    "cop_small_2x2": dict(store=True, kw=dict(n_sites=80, n_site_covs=2, n_obs_covs=2, n_periods=2,
                                              deployment_days_per_site=42, random_seed=5)),
}


def main_cop():
    simulate_cop = sys.modules["biolith.models.occu_cop"].simulate_cop
    index = {}
    for name, case in COP_CASES.items():
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            data, truth = simulate_cop(**case["kw"])
        keys = ("site_covs", "obs_covs", "obs", "session_duration")
        entry = dict(
            kwargs=case["kw"], stdout=buf.getvalue(),
            shapes={k: list(np.shape(data[k])) for k in keys}, sha256={k: sha(data[k]) for k in keys},
            coords=data["coords"], ell=float(data["ell"]), false_positives_constant=bool(data["false_positives_constant"]),
            beta=np.asarray(truth["beta"]).tolist(), alpha=np.asarray(truth["alpha"]).tolist(),
            mean_z=float(np.mean(truth["z"])), sha256_z=sha(truth["z"]), mean_obs=float(np.nanmean(data["obs"])),
            stored=bool(case["store"]),
        )
        np.savez_compressed(os.path.join(HERE, f"simulate_{name}.npz"), site_covs=data["site_covs"], obs_covs=data["obs_covs"],
                            obs=data["obs"], session_duration=data["session_duration"], z=truth["z"], beta=truth["beta"],
                            alpha=truth["alpha"])
        index[name] = entry
        print(name, entry["shapes"], entry["sha256"]["obs"][:24])
    with open(os.path.join(HERE, "simulate_cop_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


# N-mixture generator (biolith/models/nmixture.py:223-369)
NMIX_REF_TEST = dict(simulate_missing=True, deployment_days_per_site=70, session_duration=7, min_abundance=1.0,
                     min_observation_rate=1.0, max_observation_rate=6.0)          # nmixture.py:372-380
NMIX_CASES = {
    "nmix_default": dict(kw=dict()),
    "nmix_ref_test": dict(kw=NMIX_REF_TEST),
    "nmix_ref_test_3periods": dict(kw=dict(NMIX_REF_TEST, n_periods=3)),           # nmixture.py:423-431
    "nmix_small_2x2": dict(kw=dict(n_sites=60, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=42, random_seed=4)),
    # random effects (nmixture.py:285-310; the reference's own tests :516-600)
    "nmix_site_re": dict(kw=dict(NMIX_REF_TEST, site_random_effects=True, obs_random_effects=False, deployment_days_per_site=140)),
    "nmix_both_re": dict(kw=dict(n_sites=40, n_site_covs=2, n_obs_covs=1, deployment_days_per_site=42, site_random_effects=True,
                                 obs_random_effects=True, site_re_sd=0.4, obs_re_sd=0.6, random_seed=3)),
}


def main_nmix():
    simulate_nmixture = sys.modules["biolith.models.nmixture"].simulate_nmixture
    index = {}
    for name, case in NMIX_CASES.items():
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            data, truth = simulate_nmixture(**case["kw"])
        keys = ("site_covs", "obs_covs", "obs")
        index[name] = dict(
            kwargs=case["kw"], stdout=buf.getvalue(),
            shapes={k: list(np.shape(data[k])) for k in keys}, sha256={k: sha(data[k]) for k in keys},
            coords=data["coords"], ell=float(data["ell"]),
            beta=np.asarray(truth["beta"]).tolist(), alpha=np.asarray(truth["alpha"]).tolist(),
            mean_N=float(np.mean(truth["N_i"])), sha256_N=sha(truth["N_i"]), mean_obs=float(np.nanmean(data["obs"])),
            max_obs=float(np.nanmax(data["obs"])),
        )
        extra = {k: np.asarray(truth[k]) for k in ("site_re_abu", "site_re_det", "obs_re") if k in truth}
        np.savez_compressed(os.path.join(HERE, f"simulate_{name}.npz"), site_covs=data["site_covs"], obs_covs=data["obs_covs"],
                            obs=data["obs"], N_i=truth["N_i"], abundance=truth["abundance"], beta=truth["beta"], alpha=truth["alpha"], **extra)
        print(name, index[name]["shapes"], index[name]["sha256"]["obs"][:24])
    with open(os.path.join(HERE, "simulate_nmix_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


# continuous-score generator (biolith/models/occu_cs.py:222-361)
CS_CASES = {
    "cs_default": dict(kw=dict()),
    "cs_missing": dict(kw=dict(simulate_missing=True)),                                   # occu_cs.py:364-365
    "cs_small_2x2": dict(kw=dict(n_sites=70, n_site_covs=2, n_obs_covs=2, n_periods=2, deployment_days_per_site=42, random_seed=6)),
}


def main_cs():
    simulate_cs = sys.modules["biolith.models.occu_cs"].simulate_cs
    index = {}
    for name, case in CS_CASES.items():
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            data, truth = simulate_cs(**case["kw"])
        keys = ("site_covs", "obs_covs", "obs")
        index[name] = dict(kwargs=case["kw"], stdout=buf.getvalue(), shapes={k: list(np.shape(data[k])) for k in keys},
                           sha256={k: sha(data[k]) for k in keys}, coords=data["coords"], ell=float(data["ell"]),
                           truth={k: float(truth[k]) for k in ("mu0", "sigma0", "mu1", "sigma1")}, mean_z=float(np.mean(truth["z"])))
        np.savez_compressed(os.path.join(HERE, f"simulate_{name}.npz"), site_covs=data["site_covs"], obs_covs=data["obs_covs"],
                            obs=data["obs"], z=truth["z"], beta=truth["beta"], alpha=truth["alpha"])
        print(name, index[name]["shapes"], index[name]["sha256"]["obs"][:24])
    with open(os.path.join(HERE, "simulate_cs_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


def main_rn():
    simulate_rn = load_reference_simulate_rn()
    index = {}
    for name, case in RN_CASES.items():
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            data, truth = simulate_rn(**case["kw"])
        entry = dict(
            kwargs=case["kw"], stdout=buf.getvalue(),
            shapes={k: list(np.shape(data[k])) for k in ("site_covs", "obs_covs", "obs")},
            sha256={k: sha(data[k]) for k in ("site_covs", "obs_covs", "obs")},
            coords=data["coords"], ell=float(data["ell"]),
            beta=np.asarray(truth["beta"]).tolist(), alpha=np.asarray(truth["alpha"]).tolist(),
            mean_abundance=float(np.mean(truth["abundance"])), sha256_abundance=sha(truth["abundance"]),
            mean_obs=float(np.nanmean(data["obs"])), stored=bool(case["store"]),
        )
        if case["store"]:
            np.savez_compressed(os.path.join(HERE, f"simulate_{name}.npz"), site_covs=data["site_covs"],
                                obs_covs=data["obs_covs"], obs=data["obs"], abundance=truth["abundance"],
                                beta=truth["beta"], alpha=truth["alpha"])
        index[name] = entry
        print(name, entry["shapes"], entry["sha256"]["obs"][:24])
    with open(os.path.join(HERE, "simulate_rn_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


def main():
    simulate = load_reference_simulate()
    index = {}
    for name, case in CASES.items():
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            data, truth = simulate(**case["kw"])
        entry = dict(
            kwargs=case["kw"],
            stdout=buf.getvalue(),
            shapes={k: list(np.shape(data[k])) for k in ("site_covs", "obs_covs", "obs")},
            sha256={k: sha(data[k]) for k in ("site_covs", "obs_covs", "obs")},
            coords=data["coords"],
            ell=float(data["ell"]),
            beta=np.asarray(truth["beta"]).tolist(),
            alpha=np.asarray(truth["alpha"]).tolist(),
            mean_z=float(np.mean(truth["z"])),
            sha256_z=sha(truth["z"]),
            mean_obs=float(np.nanmean(data["obs"])),
            nan_frac_obs=float(np.isnan(data["obs"]).mean()),
            stored=bool(case["store"]),
        )
        if case["store"]:
            np.savez_compressed(
                os.path.join(HERE, f"simulate_{name}.npz"),
                site_covs=data["site_covs"], obs_covs=data["obs_covs"], obs=data["obs"],
                z=truth["z"], beta=truth["beta"], alpha=truth["alpha"],
            )
        index[name] = entry
        print(name, entry["shapes"], entry["sha256"]["obs"][:24])
    with open(os.path.join(HERE, "simulate_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
    main_rn()
    main_cop()
    main_nmix()
    main_cs()
