"""Resource usage of the compiled gfx950 kernels, read from the metadata notes of the code objects inside the built library
(`make` cross-compiles without a GPU; `__graft_entry__.build()` has normally done it already, so this takes seconds).

The persistent kernels are written to live in registers: occu_rn's item state (80 kept reciprocals + the 8-term table, at its
256-register budget), the control wave's loop-carried state, and two site records per lane.  A change that tips one of them into scratch does
not fail to build, it just gets slower -- or worse (occu_rn) -- so the budget is asserted here
for the headline capacity pair and for the fullest one."""
import functools
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
LLVM = "/opt/rocm/lib/llvm/bin"
CSRC = os.path.join(ROOT, "biolith_amd", "csrc")
LIB = os.path.join(ROOT, "biolith_amd", "lib", "libbiolith_hip.so")
needs_toolchain = pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))), reason="no ROCm toolchain")


@functools.lru_cache(maxsize=None)
def kernel_metadata(tmp):
    """{kernel symbol: (scratch bytes, VGPRs, static LDS bytes)} of every kernel in the library (built here if it is not up to date)."""
    r = subprocess.run(["make", "-C", CSRC, "-j6"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    work = os.path.join(tmp, "objs")
    os.makedirs(work, exist_ok=True)
    shutil.copy(LIB, work)   # (the bundles are extracted next to the file they come from)
    r = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "libbiolith_hip.so"], capture_output=True, text=True, cwd=work)
    assert r.returncode == 0, r.stderr[-2000:]
    meta = {}
    for f in sorted(os.listdir(work)):
        if not f.endswith("gfx950"):
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(work, f)], capture_output=True, text=True).stdout
        for block in notes.split("  - .agpr_count:")[1:]:
            get = lambda key: re.search(rf"\.{key}:\s+(\S+)", block).group(1)
            meta[get("name")] = (int(get("private_segment_fixed_size")), int(get("vgpr_count")), int(get("group_segment_fixed_size")))
    return meta


@needs_toolchain
@pytest.mark.parametrize("ks,ko,max_scratch", [(3, 3, 0), (4, 4, 64)])
def test_kernels_stay_in_registers(tmp_path_factory, ks, ko, max_scratch):
    meta = kernel_metadata(str(tmp_path_factory.getbasetemp()))
    mine = {n: v for n, v in meta.items() if re.match(rf"_Z\d+bl_(?:nuts|logp)_kernelILi{ks}ELi{ko}E", n)}
    assert len(mine) >= 16, sorted(mine)
    # the samplers (the hot path) hold the budget exactly; the parity hooks (one launch per bl_logp_grad call) may spill a few dwords
    for name, (scratch, vgprs, _) in mine.items():
        assert scratch <= (max_scratch if "nuts" in name else max(max_scratch, 16)), (name, scratch)
        assert vgprs <= 256, (name, vgprs)


@needs_toolchain
def test_vector_kernels_leave_the_lds_budget(tmp_path_factory):
    """The random-effects / occu_cs kernels are launched with up to 148 KB of dynamic LDS (rows + sampler vectors, re_geometry in
    biolith_hip.hip) next to their static arrays (reduction scratch, the exchange's staging, per-species sums): together they must
    stay inside the 160 KB of a workgroup -- a launch that asks for more fails at run time, not at build time.  And the capacity-4
    instantiations keep their arrays out of scratch memory."""
    meta = kernel_metadata(str(tmp_path_factory.getbasetemp()))
    src = os.path.join(CSRC, "biolith_hip.hip")
    host = open(src).read()
    budget_kb = int(re.search(r"budget = \(size_t\)(\d+) \* 1024", host).group(1))
    seen = 0
    for name, (scratch, _, static) in meta.items():
        if not re.match(r"_Z\d+bl_re_(?:nuts|logp)_kernel", name):
            continue
        assert static + budget_kb * 1024 <= 160 * 1024, (name, static)
        if "ILi4ELi4E" in name or "ILi4ELi5E" in name or ("ILi4E" in name and "logp" in name):   # (kind 5 = kind 4 with a false-positive rate)   # (the parity hook dispatches on the kind at run time: it holds kind 4's column too) kind 4 (Royle-Nichols with random effects): a thread's K + 1 terms of a (site, period) are a private column BY DESIGN
            assert scratch <= 512 + 192, (name, scratch)
        elif "ILi4E" in name:   # (a few spilled dwords in the rarer forms -- rows in device memory -- are tolerated, an array in scratch is not)
            # kind 2 (random effects + false positives: ILi4ELi2E) and the parity hook carry more live state: a few dozen dwords
            assert scratch <= (192 if ("ILi4ELi2E" in name or "logp" in name) else 64), (name, scratch)
            if "Lb1ELi2EE" in name and "ILi4ELi2E" not in name:   # the form the bench sizes run: rows and every per-leapfrog vector in LDS
                assert scratch == 0, (name, scratch)
        seen += 1
    assert seen >= 98   # 2 capacities x 8 kinds x 2 x 3 LDS forms of the sampler + 2 parity kernels
