"""The C-ABI shared library loads without a GPU and exports every symbol include/biolith_hip.h
declares; host-only entry points agree with the oracle; compute entry points fail loudly here."""
import os
import re

import numpy as np
import pytest

import oracle
from biolith_amd import _ffi
from biolith_amd.engine import OccuDataset, adaptation_schedule
from conftest import ROOT, load_golden


def _declared():
    text = open(os.path.join(ROOT, "include", "biolith_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bl_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _ffi.load()
    declared = _declared()
    assert declared == sorted(_ffi.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.bl_abi_version() == 1


def test_adaptation_schedule_through_abi():
    for w in (0, 5, 19, 20, 100, 150, 151, 300, 1000, 2000):
        assert adaptation_schedule(w) == (oracle.adaptation_schedule(w) if w > 0 else [])


@pytest.mark.skipif(_ffi.device_count() > 0, reason="only meaningful on a box without a GPU")
def test_no_gpu_means_loud_failure_not_fallback():
    g = load_golden("seed7_2x1")
    with pytest.raises(_ffi.EngineError, match="no HIP device"):
        OccuDataset(g["site_covs"], g["obs_covs"], g["obs"])
    from biolith_amd.models import occu
    from biolith_amd.utils import fit

    with pytest.raises(_ffi.EngineError, match="no CPU fallback"):
        fit(occu, site_covs=g["site_covs"], obs_covs=g["obs_covs"], obs=g["obs"], num_chains=1,
            num_samples=10, num_warmup=10)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "biolith_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "liboccu_oracle" not in src, f


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C99 on its own (no C++ or HIP types in the signatures)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "biolith_hip.h")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_library_in_tree_was_linked_from_the_sources_in_tree():
    """``__graft_entry__.build()`` skips ``make`` for a library whose stamp holds the hash of the current sources (the object files do not
    travel to a GPU box): a stamp that is present must therefore BE that hash -- an edited kernel with a stale library fails here."""
    import __graft_entry__ as entry

    stamp = os.path.join(ROOT, "biolith_amd", "lib", "build_stamp.txt")
    if not os.path.exists(stamp):
        pytest.skip("no stamp: the library was built by `make` directly; build() will run make")
    assert open(stamp).read().strip() == entry._sources_digest()


def test_environment_knobs_are_read_in_one_place_and_reported():
    """every BIOLITH_HIP_* variable of the launch path sits in ONE table (csrc/biolith_hip.hip BL_ENV_TABLE), is read by
    ONE function, and the set that is active is reported (bl_env_overrides here, without a GPU; bl_nuts_env_overrides per launch:
    tests/test_gpu_lane_groups.py).  INTEGRATION.md lists every name of the table."""
    import ctypes as C

    src = open(os.path.join(ROOT, "biolith_amd", "csrc", "biolith_hip.hip")).read()
    table = re.search(r"#define BL_ENV_TABLE\(X\)(.*?)\nenum bl_env_knob", src, flags=re.S).group(1)
    names = re.findall(r"X\(([A-Z_]+)\)", table)
    assert len(names) == len(set(names)) >= 20
    for f in os.listdir(os.path.join(ROOT, "biolith_amd", "csrc")):         # no getenv on the launch path outside the snapshot
        if f.endswith((".hip", ".hpp")):
            text = open(os.path.join(ROOT, "biolith_amd", "csrc", f)).read()
            calls = re.findall(r"getenv\(([^)]*)\)", text)
            assert all(c in ("BL_ENV_NAMES[i]", '"BIOLITH_RCCL_LIB"', '"BIOLITH_TEST_ALLOW_DUP_DEVICES"') for c in calls), (f, calls)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for n in names:
        assert re.search(r"`%s`|`[A-Z_, `/=0-9a-zµ…]*\b%s\b" % (n, n), doc), n
    lib = _ffi.load()
    buf = C.create_string_buffer(512)
    saved = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith("BIOLITH_HIP_") and k != "BIOLITH_HIP_LIB"}
    try:
        assert lib.bl_env_overrides(buf, 512) == 0 and buf.value == b""
        os.environ["BIOLITH_HIP_OCCU_G"] = "4"
        os.environ["BIOLITH_HIP_NO_WIDE"] = "1"
        os.environ["BIOLITH_HIP_NOT_A_KNOB"] = "1"
        assert lib.bl_env_overrides(buf, 512) == 0 and buf.value == b"BIOLITH_HIP_OCCU_G=4,BIOLITH_HIP_NO_WIDE=1"
    finally:
        for k in ("BIOLITH_HIP_OCCU_G", "BIOLITH_HIP_NO_WIDE", "BIOLITH_HIP_NOT_A_KNOB"):
            os.environ.pop(k, None)
        os.environ.update(saved)
