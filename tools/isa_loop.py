#!/usr/bin/env python3
"""Print the instruction mix of the rotation loop (the basic block with DPP moves that branches to itself) of a kernel
in csrc/nb_engine.gfx950.s (`make asm`).  usage: tools/isa_loop.py <mangled-name-prefix> [-v]"""
import os
import re
import sys
from collections import Counter

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
s = open(os.path.join(ROOT, "nbody3d-webgpu_amd", "csrc", "nb_engine.gfx950.s")).read()
name = sys.argv[1]
a = s.index("\n" + name)
a = s.index(":\n", a)
body = s[a:s.index(".Lfunc_end", a)]
labels = list(re.finditer(r"^(\.LBB\d+_\d+):", body, re.M))
for i, x in enumerate(labels):
    seg = body[x.end():labels[i + 1].start() if i + 1 < len(labels) else len(body)]
    ins = [l.strip() for l in seg.split("\n") if l.strip() and not l.strip().startswith(";")]
    if not any("dpp" in l for l in ins) or not any("s_cbranch" in l and x.group(1) in l for l in ins):
        continue
    print(x.group(1), len(ins), dict(Counter(l.split()[0] for l in ins)))
    if "-v" in sys.argv:
        for n, l in enumerate(ins):
            if any(k in l for k in ("ds_", "waitcnt", "v_mov_b32_e32", "s_c", "subrev", "global_", "scratch_", "s_nop")):
                print("  ", n, l)
