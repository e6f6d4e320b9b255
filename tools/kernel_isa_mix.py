#!/usr/bin/env python3
"""Instruction mix of a kernel in a built library (static count over its code, loops counted once):
    python tools/kernel_isa_mix.py lib.so 'bl_nuts_kernel<3, 3, true, 1, 5, false, 10, true>' [dump.s]"""
import os, re, struct, subprocess, sys, tempfile
from collections import Counter
lib, pat = sys.argv[1], sys.argv[2]
b = open(lib, "rb").read()
i = b.find(b"__CLANG_OFFLOAD_BUNDLE__")
while i >= 0:
    n = struct.unpack_from("<Q", b, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, s, ts = struct.unpack_from("<QQQ", b, off); off += 24
        name = b[off:off + ts]; off += ts
        if b"gfx950" in name and s:
            with tempfile.TemporaryDirectory() as t:
                p = os.path.join(t, "dev.co")
                open(p, "wb").write(b[i + o:i + o + s])
                txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", "-C", p], capture_output=True, text=True).stdout
            for blk in re.split(r"\n(?=[0-9a-f]+ <)", txt):
                head = blk.split("\n", 1)[0]
                if pat in head:
                    ins = [ln.split()[0] for ln in blk.split("\n")[1:] if ln.strip() and not ln.strip().startswith(("//", ";")) and re.match(r"\s+[a-z]", ln)]
                    c = Counter()
                    for m in ins:
                        k = ("trans" if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)", m) else "v_pk" if m.startswith("v_pk") else "dpp/permlane/readlane" if re.search(r"dpp|permlane|readlane|readfirstlane|writelane", m)
                             else "valu" if m.startswith("v_") else "lds" if m.startswith("ds_") else "branch" if m.startswith(("s_cbranch", "s_branch")) else "waitcnt" if m.startswith("s_waitcnt") else "salu" if m.startswith("s_")
                             else "vmem" if m.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
                        c[k] += 1
                    print(head.strip()[:140]); print("  instructions", len(ins), dict(c.most_common()))
                    ds = Counter(m for m in ins if m.startswith("ds_")); print("  lds:", dict(ds.most_common()))
                    if len(sys.argv) > 3: open(sys.argv[3], "w").write(blk)
    i = b.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 1)
