"""ctypes binding of ``include/biolith_hip.h`` (the C-ABI of the HIP engine).

There is deliberately no fallback: if ``libbiolith_hip.so`` is missing or no MI355X is visible the
calls raise.  Build the library with ``make -C biolith_amd/csrc -j8`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BIOLITH_HIP_LIB") or os.path.join(_HERE, "lib", "libbiolith_hip.so")

BL_OK, BL_ERR_INVALID, BL_ERR_NO_DEVICE, BL_ERR_UNSUPPORTED, BL_ERR_TIMEOUT, BL_ERR_ABORTED, BL_ERR_BUSY, BL_ERR_COMM = range(8)
COMM_ID_BYTES = 128
RNG_STREAMS_PER_CHAIN = 64
MAX_COVS = 16


class EngineError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"biolith_hip error {code}: {message}")
        self.code = code


class bl_dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_species", "n_sites", "n_periods", "n_replicates", "n_site_covs", "n_obs_covs")]


class bl_beta_prior(C.Structure):
    _fields_ = [("a", C.c_double), ("b", C.c_double)]


FP_CONSTANT, FP_UNOCCUPIED = 1, 2


class bl_normal_prior(C.Structure):
    _fields_ = [("loc", C.c_double), ("scale", C.c_double)]


class bl_nuts_config(C.Structure):
    _fields_ = [
        ("num_warmup", C.c_int32), ("num_samples", C.c_int32), ("num_chains", C.c_int32),
        ("chain_offset", C.c_int32), ("seed", C.c_uint64), ("max_tree_depth", C.c_int32),
        ("wgs_per_chain", C.c_int32), ("target_accept", C.c_double),
        ("init_theta", C.POINTER(C.c_double)),
    ]


class bl_nuts_output(C.Structure):
    _fields_ = [
        ("draws", C.POINTER(C.c_float)), ("diverging", C.POINTER(C.c_uint8)),
        ("num_steps", C.POINTER(C.c_int32)), ("accept_prob", C.POINTER(C.c_float)),
        ("potential_energy", C.POINTER(C.c_float)), ("step_size", C.POINTER(C.c_float)),
        ("inv_mass", C.POINTER(C.c_float)), ("n_leapfrog", C.POINTER(C.c_int64)),
    ]


# every symbol include/biolith_hip.h declares (tests check the library exports each one)
EXPORTS = (
    "bl_abi_version", "bl_last_error", "bl_device_count", "bl_dataset_create", "bl_dataset_create_rn", "bl_dataset_create_dyn", "bl_dataset_destroy",
    "bl_dataset_param_dim", "bl_logp_grad", "bl_nuts_run", "bl_nuts_launch", "bl_nuts_poll",
    "bl_nuts_abort", "bl_nuts_wait", "bl_nuts_fetch", "bl_nuts_elapsed_ms", "bl_nuts_device_draws",
    "bl_nuts_geometry", "bl_nuts_lane_group", "bl_nuts_kernel_name", "bl_nuts_debug_counters", "bl_deterministic", "bl_predict", "bl_predict_counts", "bl_predict_scores", "bl_dataset_create_fp", "bl_dataset_create_cop", "bl_dataset_create_nmix", "bl_dataset_create_re", "bl_dataset_create_re_fp", "bl_dataset_create_nmix_re", "bl_dataset_create_rn_re", "bl_dataset_create_rn_fp", "bl_dataset_create_cop_re", "bl_dataset_create_cs", "bl_dataset_set_prior_family", "bl_rng_streams", "bl_adaptation_schedule",
    "bl_comm_rccl_version", "bl_comm_unique_id", "bl_comm_init_rank", "bl_comm_init_all", "bl_comm_info", "bl_comm_destroy",
    "bl_gather_draws", "bl_result_block_layout", "bl_gather_unpack", "bl_host_alloc", "bl_host_free",
    "bl_nuts_env_overrides", "bl_env_overrides",
)

_lib = None
_lock = threading.Lock()


def load():
    """dlopen the engine (needs libamdhip64, not a GPU)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP engine with `make -C biolith_amd/csrc -j8` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "biolith_amd has no CPU fallback for fit()."
            )
        L = C.CDLL(LIB_PATH)
        vp, ip = C.c_void_p, C.POINTER(C.c_int)
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.bl_abi_version.restype = C.c_int
        L.bl_last_error.restype = C.c_char_p
        L.bl_device_count.argtypes = [ip]
        L.bl_dataset_create.argtypes = [C.POINTER(bl_dims), fp, fp, fp, C.POINTER(bl_normal_prior),
                                        C.POINTER(bl_normal_prior), C.c_int, C.POINTER(vp)]
        L.bl_dataset_create_rn.argtypes = [C.POINTER(bl_dims), fp, fp, fp, C.c_int, C.POINTER(bl_normal_prior),
                                           C.POINTER(bl_normal_prior), C.c_int, C.POINTER(vp)]
        L.bl_dataset_create_dyn.argtypes = [C.POINTER(bl_dims), fp, fp, fp, C.POINTER(bl_normal_prior),
                                            C.POINTER(bl_normal_prior), C.c_int, C.POINTER(vp)]
        L.bl_dataset_create_fp.argtypes = [C.POINTER(bl_dims), fp, fp, fp, C.c_int, C.POINTER(bl_beta_prior),
                                           C.POINTER(bl_normal_prior), C.POINTER(bl_normal_prior), C.c_int, C.POINTER(vp)]
        L.bl_dataset_create_cop.argtypes = [C.POINTER(bl_dims), fp, fp, fp, fp, C.c_int, C.c_double,
                                            C.POINTER(bl_normal_prior), C.POINTER(bl_normal_prior), C.c_int, C.POINTER(vp)]
        L.bl_dataset_create_nmix.argtypes = [C.POINTER(bl_dims), fp, fp, fp, C.c_int, C.POINTER(bl_normal_prior),
                                             C.POINTER(bl_normal_prior), C.c_int, C.POINTER(vp)]
        L.bl_dataset_create_re.argtypes = [C.POINTER(bl_dims), fp, fp, fp, C.c_int, C.c_int, C.c_double, C.c_double,
                                           C.POINTER(bl_normal_prior), C.POINTER(bl_normal_prior), C.c_int, C.POINTER(vp)]
        L.bl_dataset_create_re_fp.argtypes = [C.POINTER(bl_dims), fp, fp, fp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                              C.POINTER(bl_beta_prior), C.POINTER(bl_normal_prior), C.POINTER(bl_normal_prior), C.c_int, C.POINTER(vp)]
        L.bl_dataset_create_nmix_re.argtypes = [C.POINTER(bl_dims), fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                                C.POINTER(bl_normal_prior), C.POINTER(bl_normal_prior), C.c_int, C.POINTER(vp)]
        L.bl_dataset_create_rn_re.argtypes = L.bl_dataset_create_nmix_re.argtypes
        L.bl_dataset_create_rn_fp.argtypes = [C.POINTER(bl_dims), fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                              C.POINTER(bl_beta_prior), C.POINTER(bl_normal_prior), C.POINTER(bl_normal_prior), C.c_int, C.POINTER(vp)]
        L.bl_dataset_create_cop_re.argtypes = [C.POINTER(bl_dims), fp, fp, fp, fp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double,
                                               C.POINTER(bl_normal_prior), C.POINTER(bl_normal_prior), C.c_int, C.POINTER(vp)]
        L.bl_dataset_create_cs.argtypes = [C.POINTER(bl_dims), fp, fp, fp, dp, dp, C.POINTER(bl_normal_prior),
                                           C.POINTER(bl_normal_prior), C.c_int, C.POINTER(vp)]
        L.bl_dataset_set_prior_family.argtypes = [vp, C.c_int, C.c_int]
        L.bl_dataset_destroy.argtypes = [vp]
        L.bl_dataset_param_dim.argtypes = [vp, ip]
        L.bl_logp_grad.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int]
        L.bl_nuts_run.argtypes = [vp, C.POINTER(bl_nuts_config), C.POINTER(bl_nuts_output)]
        L.bl_nuts_launch.argtypes = [vp, C.POINTER(bl_nuts_config), vp]
        L.bl_nuts_poll.argtypes = [vp, ip]
        L.bl_nuts_abort.argtypes = [vp]
        L.bl_nuts_wait.argtypes = [vp]
        L.bl_nuts_fetch.argtypes = [vp, C.POINTER(bl_nuts_output)]
        L.bl_nuts_elapsed_ms.argtypes = [vp, fp]
        L.bl_nuts_device_draws.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.bl_nuts_geometry.argtypes = [vp, ip, ip, ip, ip, ip]
        L.bl_nuts_lane_group.argtypes = [vp, ip, ip]
        L.bl_nuts_kernel_name.argtypes = [vp, C.c_char_p, C.c_int]
        if hasattr(L, "bl_nuts_env_overrides"):   # (absent from libraries built before round 6: tools/ A/B them by name, BIOLITH_HIP_LIB)
            L.bl_nuts_env_overrides.argtypes = [vp, C.c_char_p, C.c_int]
            L.bl_env_overrides.argtypes = [C.c_char_p, C.c_int]
        L.bl_nuts_debug_counters.argtypes = [vp, C.POINTER(C.c_int64), C.c_int]
        L.bl_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        L.bl_host_free.argtypes = [C.c_void_p]
        L.bl_deterministic.argtypes = [vp, C.c_int, fp, fp, fp]
        L.bl_predict.argtypes = [vp, C.c_int, fp, C.c_uint64, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
        L.bl_predict_counts.argtypes = [vp, C.c_int, fp, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.bl_predict_scores.argtypes = [vp, C.c_int, fp, C.c_uint64, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), fp]
        L.bl_rng_streams.argtypes = [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
        L.bl_adaptation_schedule.argtypes = [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]
        u8p = C.POINTER(C.c_uint8)
        L.bl_comm_rccl_version.argtypes = [ip]
        L.bl_comm_unique_id.argtypes = [u8p]
        L.bl_comm_init_rank.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.bl_comm_init_all.argtypes = [C.c_int, ip, C.POINTER(vp)]
        L.bl_comm_info.argtypes = [vp, ip, ip, ip, dp]
        L.bl_comm_destroy.argtypes = [vp]
        L.bl_gather_draws.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int, C.POINTER(C.c_int32), C.POINTER(bl_nuts_output)]
        L.bl_result_block_layout.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.bl_gather_unpack.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(bl_nuts_output)]
        for name in EXPORTS:
            if name != "bl_last_error" and (hasattr(L, name) or name not in ("bl_nuts_env_overrides", "bl_env_overrides")):
                getattr(L, name).restype = C.c_int
        if L.bl_abi_version() != 1:
            raise RuntimeError("libbiolith_hip.so ABI version mismatch")
        _lib = L
        return _lib


def check(rc: int):
    if rc != BL_OK:
        msg = load().bl_last_error().decode("utf-8", "replace")
        if rc == BL_ERR_INVALID:
            raise ValueError(f"biolith_hip: {msg}")
        if rc == BL_ERR_UNSUPPORTED:
            raise NotImplementedError(f"biolith_hip: {msg}")
        if rc in (BL_ERR_TIMEOUT,):
            raise TimeoutError(f"biolith_hip: {msg}")
        raise EngineError(rc, msg)


def device_count() -> int:
    n = C.c_int(0)
    rc = load().bl_device_count(C.byref(n))
    return n.value if rc == BL_OK else 0
