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
# The stats file averages EVERY launch of the process: the gate step, the warm-up steps (lower clock), the timed steps and
# the extra energy-drift step.  The per-dispatch trace gives the timed launches on their own -- what bench.py's
# roofline.avg_launch_ms measures with HIP events: launches [warmup, warmup + steps) of the force kernel (gpu_prof.sh runs
# the trace pass with --steps 20 --warmup 5: 1 gate + 4 warm-up + 20 timed + 1 extra).
trace = os.path.join(src, "trace", "trace_kernel_trace.csv")
TRACE_WARMUP, TRACE_STEPS = 5, 20
if os.path.exists(trace):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    for k, v in per.items():
        v.sort()
        if len(v) >= TRACE_WARMUP + TRACE_STEPS and ("nb_force" in k or "nb_step_" in k or "nb_integrate" in k):
            timed = [d for _, d in v[TRACE_WARMUP:TRACE_WARMUP + TRACE_STEPS]]
            out["kernels"].setdefault(k, {})["trace_timed_launches"] = {
                "launches": len(timed), "avg_ms": sum(timed) / len(timed), "min_ms": min(timed), "max_ms": max(timed),
                "note": "launches %d..%d of %d: the ones bench.py times" % (TRACE_WARMUP, TRACE_WARMUP + TRACE_STEPS - 1, len(v))}
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
bl = os.path.join(src, "bench_line.json")
if k1 and os.path.exists(bl):
    try:       # how well the tracked summary reproduces the bench line it corroborates
        line = json.load(open(bl))
        ev = line["roofline"]["avg_launch_ms"]
        tt = out["kernels"][k1[0]].get("trace_timed_launches", {}).get("avg_ms")
        out["bench_line_check"] = {"bench_avg_launch_ms_hip_events": ev, "rocprof_timed_launches_avg_ms": tt,
                                   "rel_diff": None if not tt else (tt - ev) / ev, "roofline_frac_in_line": line["roofline"]["frac"],
                                   "roofline_frac_from_rocprof": None if not tt else line["roofline"]["frac"] * ev / tt}
        json.dump(out, open(os.path.join(dst, "summary.json"), "w"), indent=1)
    except Exception as e:
        out["bench_line_check"] = {"error": str(e)}
if k1 and "hbm_bytes_per_launch" in out["kernels"][k1[0]] and len(sys.argv) > 6 and sys.argv[6] == "--write-traffic":
    h = out["kernels"][k1[0]]["hbm_bytes_per_launch"]
    json.dump({"n": n, "n_gpus": 1, "dtype": dtype, "kernel": k1[0], "kernel_variant": variant, "source": dst,
               "bytes_per_launch": h["read"] + h["written"], "read": h["read"], "written": h["written"],
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; KiB units; FETCH_SIZE x2 on gfx950 "
                         "(MI355X_MICROARCH.md 'HBM')"},
              open("profiles/k1_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
