"""CPU sanitizer run of the C oracle (SURVEY.md section 5 "race detection / sanitizers").

The 1.1 k-line ``oracle/occu_oracle.c`` is the ground truth of every GPU parity test; it carves a heap pool by hand and
keeps stack arrays of ``ORC_MAX_D``.  Here its ASan + UBSan build (``make -C oracle liboccu_oracle_asan.so``) evaluates the
potential + gradient and runs a short NUTS for EVERY model id (occu, occu_rn, false positives x2, occu_cop x2, nmixture,
random effects x3, occu_cs, three species under one chain x3; Normal and Laplace priors; missing data; several periods) in a fresh child process with libasan
preloaded.  Any report aborts the child (``halt_on_error``).  CPU only: GPU sanitizers are not available on this pool."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

CHILD = textwrap.dedent("""
    import contextlib, io, os, sys
    sys.path.insert(0, %r)
    os.environ["OCCU_ORACLE_FLAVOR"] = "asan"
    import numpy as np
    import oracle
    from biolith_amd.models import simulate, simulate_rn, simulate_cop, simulate_nmixture, simulate_cs
    assert oracle.lib()._name.endswith("liboccu_oracle_asan.so")

    def quiet(fn, **kw):
        with contextlib.redirect_stdout(io.StringIO()):
            return fn(**kw)[0]

    small = dict(n_sites=40, n_site_covs=2, n_obs_covs=2, deployment_days_per_site=42, session_duration=7, random_seed=1)
    occ = quiet(simulate, simulate_missing=True, n_periods=2, **small)
    rn = quiet(simulate_rn, **small)
    cop = quiet(simulate_cop, **small)
    nmx = quiet(simulate_nmixture, **small)
    cs = quiet(simulate_cs, **small)
    base = lambda d: (d["site_covs"], d["obs_covs"], d["obs"])
    cases = [
        ("occu", oracle.OracleData(*base(occ))),
        ("occu laplace", oracle.OracleData(*base(occ), prior_family=("laplace", "normal"))),
        ("occu_rn", oracle.OracleData(*base(rn), model="occu_rn", max_abundance=30)),
        ("occu_fp constant", oracle.OracleData(*base(occ), model="occu_fp", fp_mode="constant")),
        ("occu_fp unoccupied", oracle.OracleData(*base(occ), model="occu_fp", fp_mode="unoccupied")),
        ("occu_cop", oracle.OracleData(*base(cop), model="occu_cop", fp_mode=None, session_duration=cop["session_duration"])),
        ("occu_cop fp", oracle.OracleData(*base(cop), model="occu_cop", fp_mode="constant", session_duration=cop["session_duration"])),
        ("nmixture", oracle.OracleData(*base(nmx), model="nmixture", max_abundance=60)),
        ("occu_re site", oracle.OracleData(*base(occ), model="occu_re", site_random_effects=True)),
        ("occu_re obs", oracle.OracleData(*base(occ), model="occu_re", obs_random_effects=True)),
        ("occu_re both", oracle.OracleData(*base(occ), model="occu_re", site_random_effects=True, obs_random_effects=True)),
        ("occu_cs", oracle.OracleData(cs["site_covs"], cs["obs_covs"], cs["scores"] if "scores" in cs else cs["obs"], model="occu_cs")),
    ]
    # several species under one chain (shared false-positive rate / shared random-effect sds)
    sp = quiet(simulate, n_species=3, simulate_missing=True, **small)
    cases += [
        ("occu 3 species", oracle.OracleData(*base(sp))),
        ("occu_fp 3 species", oracle.OracleData(*base(sp), model="occu_fp", fp_mode="constant")),
        ("occu_re 3 species", oracle.OracleData(*base(sp), model="occu_re", site_random_effects=True, obs_random_effects=True)),
    ]
    rng = np.random.default_rng(0)
    for name, od in cases:
        th = rng.uniform(-1.0, 1.0, size=(3, od.D))
        U, G = od.potential_grad(th)
        assert np.all(np.isfinite(U)) and np.all(np.isfinite(G)), name
        r = oracle.nuts_run(od, num_warmup=25, num_samples=10, num_chains=2, seed=5, threads=2)
        assert r["draws"].shape == (2, 10, od.D) and np.all(np.isfinite(r["draws"])), name
        print("ok", name, od.D, int(r["n_leapfrog"].sum()), flush=True)
    # the RNG helpers and the schedule restatement
    oracle.rng_streams(3, 2, 64); oracle.adaptation_schedule(1000); oracle.adaptation_schedule(7)
    print("SANITIZER RUN COMPLETE", flush=True)
""")


@pytest.mark.timeout(900)
def test_oracle_is_clean_under_asan_and_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not installed with this gcc")
    env = dict(os.environ, LD_PRELOAD=os.path.realpath(asan),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=23",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=24", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=850)
    tail = (r.stdout[-1500:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "SANITIZER RUN COMPLETE" in r.stdout, tail
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
    assert r.stdout.count("ok ") == 15
