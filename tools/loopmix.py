#!/usr/bin/env python3
"""Instruction mix of the hottest loop (the basic block with the most v_pk_* / v_fma_f64) of each kernel in a gfx950 .s file."""
import collections, re, subprocess, sys
path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
txt = open(path).read().split("\n")
kern, blocks, name, cur = {}, None, None, None
for line in txt:
    m = re.match(r"^(_Z[\w]+):", line)
    if m:
        name = m.group(1); blocks = kern.setdefault(name, []); cur = []; blocks.append(cur); continue
    if name is None: continue
    if re.match(r"^\.LBB\d+_\d+:", line): cur = []; blocks.append(cur); continue
    if "s_endpgm" in line: name = None; continue
    t = line.strip()
    if t and not t.startswith((";", ".")): cur.append(t.split()[0])
for k, bl in kern.items():
    dn = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void nb::", "")
    if pat and pat not in dn: continue
    best = max(bl, key=lambda b: sum(1 for i in b if i.startswith(("v_pk_", "v_fma_f64", "v_fmac_f64"))))
    c = collections.Counter()
    for i in best:
        key = ("v_pk" if i.startswith("v_pk_") else "v_rsq" if i.startswith("v_rsq") else "ds_read" if i.startswith("ds_read") else
               "s_nop" if i == "s_nop" else "v_mov" if i.startswith("v_mov") else "s_waitcnt" if i == "s_waitcnt" else
               "dp" if re.match(r"v_(fma|fmac|mul|add)_f64", i) else "scratch" if i.startswith("scratch") else "salu" if i.startswith("s_") else "other_v")
        c[key] += 1
    print("%-34s %s" % (dn, dict(c)))
