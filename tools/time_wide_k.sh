python - <<'PY'
import os, sys, io, contextlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import time_wide as T
from biolith_amd.engine import OccuDataset
from biolith_amd.models import simulate
for i, ks in ((7, [66, 80, 96, 100, 112, 128]), (6, [])):
    n_sites, visits = int(100 * 2 ** i), int(8 * 2 ** (i / 2))
    with contextlib.redirect_stdout(io.StringIO()):
        data, _ = simulate(n_site_covs=2, n_obs_covs=1, n_sites=n_sites, deployment_days_per_site=visits * 7, session_duration=7, simulate_missing=False, random_seed=42 + i)
    ds = OccuDataset(data["site_covs"], data["obs_covs"], data["obs"])
    print(f"row {i}: {n_sites} x {visits}")
    for v in ("BIOLITH_HIP_WIDE_K", "BIOLITH_HIP_OCCU_G"): os.environ.pop(v, None)
    T.run(ds, "host's choice")
    for k in ks:
        for G in (2, 4):
            os.environ["BIOLITH_HIP_WIDE_K"], os.environ["BIOLITH_HIP_OCCU_G"] = str(k), str(G)
            T.run(ds, f"forced k={k} G={G}")
PY
