#!/usr/bin/env python3
"""Sustained throughput: the headline system stepped back to back for minutes, ms/step per ~5 s window (bench.py times 0.2 s).
Shows what the clock does once the device is warm.  usage: sustained.py [seconds] [N]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
b, v = ic.plummer(n, seed=1)
rows = []
with Simulation(n) as sim:
    sim.init(b, v)
    sim.set_params(1e-3, 1.0)
    ke0, pe0, _ = sim.diagnostics()
    per = max(16, int(5.0 / (n * float(n) / 6.5e12)) // 16 * 16)
    t_start = time.perf_counter()
    steps = 0
    while time.perf_counter() - t_start < seconds:
        t0 = time.perf_counter()
        sim.simulate(per)
        sim.sync()
        dt = time.perf_counter() - t0
        steps += per
        rows.append({"t": time.perf_counter() - t_start, "ms_per_step": 1e3 * dt / per, "frac": n * (n - 1.0) * per / dt / 7.865e12})
        print("t=%6.1f s  %8.3f ms/step  %5.1f %%" % (rows[-1]["t"], rows[-1]["ms_per_step"], 100 * rows[-1]["frac"]), flush=True)
    bb = sim.read(vel=False, accel=False)[0]
    name = sim.variant
ms = np.array([r["ms_per_step"] for r in rows])
out = {"n": n, "variant": name, "seconds": rows[-1]["t"], "steps": steps, "windows": len(rows), "ms_per_step_first": float(ms[0]),
       "ms_per_step_median": float(np.median(ms)), "ms_per_step_min": float(ms.min()), "ms_per_step_max": float(ms.max()),
       "ms_per_step_last_minute": float(ms[-12:].mean()), "frac_median": float(np.median([r["frac"] for r in rows])),
       "finite": bool(np.isfinite(bb).all()), "rows": rows}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "sustained.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "rows"}))
