#!/usr/bin/env python3
"""Tabulates hipcc -Rpass-analysis=kernel-resource-usage output (csrc/nb_engine.resources.txt)."""
import re, subprocess, sys
path = sys.argv[1] if len(sys.argv) > 1 else "nbody3d-webgpu_amd/csrc/nb_engine.resources.txt"
rows, cur = [], None
for line in open(path):
    m = re.search(r"remark:\s+([A-Za-z ]+):\s+(\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    try:
        r["name"] = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void nb::", "")
    except Exception:
        pass
print("%-44s %5s %5s %4s %7s %7s %6s" % ("kernel", "VGPR", "SGPR", "occ", "LDS", "scratch", "spill"))
for r in rows:
    print("%-44s %5s %5s %4s %7s %7s %6s" % (r["name"][:44], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("Occupancy [waves/SIMD]", r.get("Occupancy")),
                                          r.get("LDS Size [bytes/block]", r.get("LDS Size")), r.get("ScratchSize [bytes/lane]", r.get("ScratchSize")), r.get("VGPRs Spill")))
