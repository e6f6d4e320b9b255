"""CPU oracle for the occupancy NUTS hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``biolith_amd``) never does; it fails loudly when its HIP
library is missing instead of falling back to anything in here.

Parity status (see ``occu_oracle.c`` header and DESIGN.md): the log-density and the sampler are
"parity unpinned" (the reference pins no numbers for them and numpyro/jax cannot be installed);
the simulator is pinned bit-exactly by ``tests/golden``.
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
