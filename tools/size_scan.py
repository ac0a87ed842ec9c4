#!/usr/bin/env python3
"""pairs/s and fraction of the fp32 roofline for N = 2^10 .. 2^20 Plummer spheres on one GPU
(north_star: "throughput on synthetic N=2^k particle clouds ... as absolute pair-interactions/s
and as fraction of fp32 roofline").  Wall time around sim.simulate(k) + sync, default shapes;
>= 0.25 s of warm-up (the clocks ramp over milliseconds) and the better of two >= 0.3 s windows."""
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1 << k for k in range(10, 21)]
rows = []
for n in sizes:
    b, v = ic.plummer(n, seed=1)
    est = max(n * n / 4.5e12, 3.5e-6)                 # seconds per step, rough
    steps = max(3, min(100000, int(0.3 / est) // 16 * 16 or 3))
    with Simulation(n) as sim:
        sim.init(b, v)
        sim.simulate(max(2, int(0.25 / est)), 1e-3, 1.0)
        sim.sync()
        best = 1e30
        for _ in range(2):
            t0 = time.perf_counter()
            sim.simulate(steps)
            sim.sync()
            best = min(best, time.perf_counter() - t0)
        rate = n * (n - 1) * steps / best
        rows.append({"n": n, "steps": steps, "us_per_step": 1e6 * best / steps, "pairs_per_s": rate,
                     "frac_fp32_roofline": rate / 7.865e12, "variant": sim.variant})
        print("N=%8d  %-34s %10.2f us/step  %.3e pairs/s  %5.1f %%" % (n, sim.variant, 1e6 * best / steps, rate,
                                                                         100 * rate / 7.865e12), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "size_scan.json"), "w"), indent=1)
