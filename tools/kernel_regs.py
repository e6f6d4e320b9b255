#!/usr/bin/env python3
"""Registers, scratch and spills of the kernels in a built library whose demangled names contain a pattern.
    python tools/kernel_regs.py biolith_amd/lib/libbiolith_hip.so 'bl_nuts_kernel<3, 3, true, 1,'"""
import re, struct, subprocess, sys, tempfile, os
lib, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
b = open(lib, "rb").read()
txt = ""
i = b.find(b"__CLANG_OFFLOAD_BUNDLE__")
while i >= 0:  # one bundle per translation unit that holds device code
    n = struct.unpack_from("<Q", b, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, s, ts = struct.unpack_from("<QQQ", b, off)
        off += 24
        name = b[off:off + ts]
        off += ts
        if b"gfx950" in name and s:
            with tempfile.TemporaryDirectory() as t:
                p = os.path.join(t, "dev.co")
                open(p, "wb").write(b[i + o:i + o + s])
                txt += subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", p], capture_output=True, text=True).stdout
    i = b.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 1)
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    if pat in name:
        print(name[:120], "| vgpr", g("vgpr_count"), "agpr", blk.split()[0], "sgpr", g("sgpr_count"), "scratch", g("private_segment_fixed_size"), "vgpr spills", g("vgpr_spill_count"), "sgpr spills", g("sgpr_spill_count"))
