"""CPU oracle for the occupancy NUTS hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``biolith_amd``) never does; it fails loudly when its HIP
library is missing instead of falling back to anything in here.

Parity status (see ``occu_oracle.c`` header and DESIGN.md section 3): the log-densities are PINNED to values the reference's own
model functions produced (``tests/golden/reference_logjoint_*.json``, made by ``tests/golden/make_reference_logjoint.py``) and the
simulators bit for bit (``tests/golden/make_golden.py``); the sampler is "parity unpinned" against numpyro's trees (numpyro / jax cannot
be installed) and is held to the distribution it must sample instead (``tests/quadrature.py``, ``tests/sbc.py``).
"""
from .oracle import (  # noqa: F401
    OracleData,
    adaptation_schedule,
    build,
    effective_sample_size,
    lib,
    literal_log_joint,
    literal_log_joint_cop,
    literal_log_joint_cs,
    literal_log_joint_dyn,
    literal_log_joint_fp,
    literal_log_joint_nmix,
    literal_log_joint_re,
    literal_log_joint_rn,
    nuts_run,
    rng_streams,
    split_gelman_rubin,
)
