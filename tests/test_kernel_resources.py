"""Resource usage of the compiled gfx950 kernels (hipcc cross-compiles without a GPU).

The persistent kernels are written to live in registers: occu_rn's item state (80 kept reciprocals + the 8-term table, at its
256-register budget), the control wave's loop-carried state, and two site records per lane.  A change that tips one of them into scratch does
not fail to build, it just gets slower -- or worse (occu_rn) -- so the budget is asserted here
for the headline capacity pair and for the fullest one."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
         "-fgpu-flush-denormals-to-zero", "--cuda-device-only", "-S"]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("ks,ko,max_scratch", [(3, 3, 0), (4, 4, 64)])
def test_kernels_stay_in_registers(tmp_path, ks, ko, max_scratch):
    out = tmp_path / "inst.s"
    src = os.path.join(ROOT, "biolith_amd", "csrc", "kernels_inst.hip")
    r = subprocess.run([HIPCC, *FLAGS, f"-DBL_KS={ks}", f"-DBL_KO={ko}", "-o", str(out), src], capture_output=True, text=True,
                       cwd=os.path.join(ROOT, "biolith_amd", "csrc"))
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    kernels = re.findall(r"^(_Z\w*bl_(?:nuts|logp)_kernel\w*):", text, flags=re.M)
    scratch = [int(x) for x in re.findall(r"; ScratchSize: (\d+)", text)]
    vgprs = [int(x) for x in re.findall(r"; NumVgprs: (\d+)", text)]
    assert len(kernels) >= 16 and len(scratch) >= len(kernels)
    # the samplers (the hot path) hold the budget exactly; the parity hooks (one launch per bl_logp_grad call) may spill a few dwords
    for name, sc in zip(kernels, scratch):
        assert sc <= (max_scratch if "nuts" in name else max(max_scratch, 16)), (name, sc)
    assert max(vgprs) <= 256


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_vector_kernels_leave_the_lds_budget(tmp_path):
    """The random-effects / occu_cs kernels are launched with up to 148 KB of dynamic LDS (rows + sampler vectors, re_geometry in
    biolith_hip.hip) next to their static arrays (reduction scratch, the exchange's staging, per-species sums): together they must
    stay inside the 160 KB of a workgroup -- a launch that asks for more fails at run time, not at build time.  And the capacity-4
    instantiations keep their arrays out of scratch memory."""
    out = tmp_path / "main.s"
    src = os.path.join(ROOT, "biolith_amd", "csrc", "biolith_hip.hip")
    r = subprocess.run([HIPCC, *FLAGS, "-o", str(out), src], capture_output=True, text=True, cwd=os.path.join(ROOT, "biolith_amd", "csrc"))
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    host = open(src).read()
    budget_kb = int(re.search(r"budget = \(size_t\)(\d+) \* 1024", host).group(1))
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (_Z\d+bl_re_(?:nuts|logp)_kernel\w+)\n(.*?)\.end_amdhsa_kernel", text, flags=re.S):
        name, body = m.group(1), m.group(2)
        static = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1))
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        assert static + budget_kb * 1024 <= 160 * 1024, (name, static)
        if "ILi4E" in name:   # (a few spilled dwords in the rarer forms -- rows in device memory -- are tolerated, an array in scratch is not)
            # kind 2 (random effects + false positives: ILi4ELi2E) and the parity hook carry more live state: a few dozen dwords
            assert scratch <= (192 if ("ILi4ELi2E" in name or "logp" in name) else 64), (name, scratch)
            if "Lb1ELi2EE" in name and "ILi4ELi2E" not in name:   # the form the bench sizes run: rows and every per-leapfrog vector in LDS
                assert scratch == 0, (name, scratch)
        seen += 1
    assert seen >= 50   # 2 capacities x 4 kinds x 2 x 3 LDS forms of the sampler + 2 parity kernels
