import contextlib, io, os, subprocess, sys
import numpy as np
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    from biolith_amd.engine import OccuDataset
    from biolith_amd.models import simulate_rn
    with contextlib.redirect_stdout(io.StringIO()):
        d, _ = simulate_rn(n_sites=5000, n_site_covs=3, n_obs_covs=3, deployment_days_per_site=70, session_duration=7)
    ds = OccuDataset(d["site_covs"], d["obs_covs"], d["obs"], model="occu_rn")
    r = ds.nuts(num_warmup=300, num_samples=300, num_chains=4, seed=11)
    np.savez(sys.argv[1], draws=r.draws, steps=r.num_steps, eps=r.step_size, minv=r.inv_mass, acc=r.accept_prob)
else:
    for n in (os.environ.get("AB_A", "h1"), os.environ.get("AB_B", "h2")):
        subprocess.run([sys.executable, __file__, f"/tmp/{n}.npz"], env=dict(os.environ, BIOLITH_HIP_LIB=f"{ROOT}/biolith_amd/lib/libbiolith_hip_{n}.so"), check=True)
    a, b = np.load(f"/tmp/{os.environ.get('AB_A', 'h1')}.npz"), np.load(f"/tmp/{os.environ.get('AB_B', 'h2')}.npz")
    print({k: bool(np.array_equal(a[k], b[k])) for k in a.files})
