#!/usr/bin/env python3
"""Calibration of the planner's cost model (plan_launch, csrc/nb_plan.cpp) against scanned timings.

    python tools/fit_model.py picks  > gpurun_out/model_picks.json      (on the GPU box: hipOccupancy needs a device)
    python tools/fit_model.py regret gpurun_out/model_picks.json profiles/r02/shape_scan_dma_*.txt

`picks` creates a default handle at every scanned size for a grid of model constants (NB_MODEL_* environment
variables, read by nb_create in the -DNB_TUNING build of the library) and records which launch shape the model takes.  `regret` looks every pick up
in the scanned (variant -> us/step) tables and prints, per constant set, the worst and mean regret."""
import itertools
import json
import os
import re
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
# the release library compiles the model constants in; the calibration build (make -C nbody3d-webgpu_amd/csrc tuning) reads NB_MODEL_*
os.environ.setdefault("NB_ENGINE_LIB", os.path.join(ROOT, "nbody3d-webgpu_amd", "csrc", "libnbody3d_hip_tuning.so"))
SIZES = [int(x) for x in os.environ.get("NB_FIT_SIZES", "1024,2048,3000,4096,5000,6000,7000,8192,10000,12000,14000,16384,20000,32768,40002,65536").split(",")]
GRID = {"NB_MODEL_TILE_LATENCY": [3000, 2200], "NB_MODEL_HANDOVER": [350, 250, 150, 50, 0],
        "NB_MODEL_LANES_SCALE": [1.0, 1.03, 1.06, 1.09, 1.12], "NB_MODEL_BOUNDARY": [3e-6, 4e-6, 5e-6]}


def picks():
    sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
    from nbody3d_amd import Simulation
    out = []
    keys = sorted(GRID)
    for combo in itertools.product(*(GRID[k] for k in keys)):
        for k, v in zip(keys, combo):
            os.environ[k] = repr(v)
        row = {"knobs": dict(zip(keys, combo)), "picks": {}}
        for n in SIZES:
            with Simulation(n) as s:
                row["picks"][str(n)] = s.variant
        out.append(row)
    json.dump(out, sys.stdout)


def regret(picks_path, scans):
    table = {}
    for path in scans:
        n = None
        for line in open(path):
            m = re.match(r"== N=(\d+)", line)
            if m:
                n = int(m.group(1))
                continue
            m = re.match(r"(\S+)\s+(\S+)\s+([\d.]+) us/step", line)
            if m and n:
                t = table.setdefault(n, {})
                t[m.group(2)] = min(t.get(m.group(2), 1e9), float(m.group(3)))
    rows = []
    for row in json.load(open(picks_path)):
        worst, tot, cnt, missing, detail = 0.0, 0.0, 0, [], []
        for n, var in row["picks"].items():
            t = table.get(int(n), {})
            # the j-packed arm is not offered by the model: compare against what the model can reach
            best = min(v for k, v in t.items() if "jpairs" not in k)
            if var not in t:
                missing.append((n, var))
                continue
            r = t[var] / best - 1.0
            worst, tot, cnt = max(worst, r), tot + r, cnt + 1
            detail.append((int(n), var, round(100 * r, 1)))
        rows.append((worst, tot / max(cnt, 1), len(missing), row["knobs"], detail, missing))
    rows.sort(key=lambda r: (r[2], r[0] + r[1]))
    for w, mean, miss, knobs, detail, missing in rows[:8]:
        print("worst %.1f %%  mean %.2f %%  unscanned picks %d  %s" % (100 * w, 100 * mean, miss, knobs))
    w, mean, miss, knobs, detail, missing = rows[0]
    print("best set, per size:", detail)
    print("unscanned:", missing)
    base = [r for r in rows if r[3] == {k: GRID[k][0] for k in GRID}]
    if base:
        print("current constants: worst %.1f %%  mean %.2f %%  unscanned %d" % (100 * base[0][0], 100 * base[0][1], base[0][2]), base[0][4], base[0][5])


if __name__ == "__main__":
    if sys.argv[1] == "picks":
        picks()
    else:
        regret(sys.argv[2], sys.argv[3:])
