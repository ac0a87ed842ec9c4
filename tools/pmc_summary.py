#!/usr/bin/env python3
"""Summarises the rocprofv3 CSVs written by tools/gpu_prof.sh into
profiles/<round>/: per-kernel duration stats, PMC averages, derived clock /
VALU-busy figures, and profiles/k1_hbm_traffic.json (HBM bytes per K1 launch:
FETCH_SIZE and WRITE_SIZE from separate passes, KiB units, FETCH doubled per
MI355X_MICROARCH.md 'HBM' -- gfx950 reports half the bytes of wide reads)."""
import collections
import csv
import json
import os
import sys

src, dst, n, dtype = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
variant = sys.argv[5] if len(sys.argv) > 5 else None      # nb_variant_name() of the profiled run (bench.py keys on it)
os.makedirs(dst, exist_ok=True)
out = {"source": src, "n": n, "dtype": dtype, "kernel_variant": variant, "kernels": {}}


def short(name):
    return name.split("(")[0].replace("void ", "")


stats = os.path.join(src, "trace", "trace_kernel_stats.csv")
if os.path.exists(stats):
    for r in csv.DictReader(open(stats)):
        out["kernels"].setdefault(short(r["Name"]), {})["trace"] = {
            "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6,
            "max_ms": float(r["MaxNs"]) / 1e6, "pct": float(r["Percentage"])}
    open(os.path.join(dst, "kernel_stats.csv"), "w").write(open(stats).read())
for p in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    f = os.path.join(src, p, p + "_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, cs in agg.items():
        d = out["kernels"].setdefault(k, {}).setdefault("pmc", {})
        for c, v in cs.items():
            d[c] = sum(v) / len(v)
        out["kernels"][k].setdefault("pmc_pass_ms", {})[p] = sum(dur[k]) / len(dur[k])
for k, d in out["kernels"].items():
    pm = d.get("pmc", {})
    if "GRBM_GUI_ACTIVE" in pm and "pmc_sq2" in d.get("pmc_pass_ms", {}):
        d["eff_clock_GHz"] = pm["GRBM_GUI_ACTIVE"] / 8 / (d["pmc_pass_ms"]["pmc_sq2"] * 1e-3) / 1e9
    if "SQ_ACTIVE_INST_VALU" in pm and "SQ_BUSY_CYCLES" in pm and "pmc_sq1" in d.get("pmc_pass_ms", {}):
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over SIMDs; 1024 SIMDs
        clk = d.get("eff_clock_GHz", 2.3) * 1e9
        d["valu_busy_frac"] = pm["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (d["pmc_pass_ms"]["pmc_sq1"] * 1e-3 * clk)
        d["valu_cycles_per_inst"] = pm["SQ_ACTIVE_INST_VALU"] * 4 / pm["SQ_INSTS_VALU"] if pm.get("SQ_INSTS_VALU") else None
    if "FETCH_SIZE" in pm or "WRITE_SIZE" in pm:
        d["hbm_bytes_per_launch"] = {"read": 2 * 1024 * pm.get("FETCH_SIZE", 0), "written": 1024 * pm.get("WRITE_SIZE", 0)}
json.dump(out, open(os.path.join(dst, "summary.json"), "w"), indent=1)
k1 = [k for k in out["kernels"] if "nb_force" in k or "nb_step_fused" in k]
if k1 and "hbm_bytes_per_launch" in out["kernels"][k1[0]] and len(sys.argv) > 6 and sys.argv[6] == "--write-traffic":
    h = out["kernels"][k1[0]]["hbm_bytes_per_launch"]
    json.dump({"n": n, "n_gpus": 1, "dtype": dtype, "kernel": k1[0], "kernel_variant": variant, "source": dst,
               "bytes_per_launch": h["read"] + h["written"], "read": h["read"], "written": h["written"],
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; KiB units; FETCH_SIZE x2 on gfx950 "
                         "(MI355X_MICROARCH.md 'HBM')"},
              open("profiles/k1_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
