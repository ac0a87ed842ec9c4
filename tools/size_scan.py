#!/usr/bin/env python3
"""pairs/s and fraction of the fp32 roofline for N = 2^10 .. 2^20 Plummer spheres on one GPU
(north_star: "throughput on synthetic N=2^k particle clouds ... as absolute pair-interactions/s
and as fraction of fp32 roofline").  Wall time around sim.simulate(k) + sync, default shapes."""
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

rows = []
for k in range(10, 21):
    n = 1 << k
    b, v = ic.plummer(n, seed=1)
    steps = max(3, min(2000, int(2e11 / (n * n))))
    with Simulation(n) as sim:
        sim.init(b, v)
        sim.simulate(max(2, steps // 10), 1e-3, 1.0)
        sim.sync()
        t0 = time.perf_counter()
        sim.simulate(steps)
        sim.sync()
        dt = time.perf_counter() - t0
        rate = n * (n - 1) * steps / dt
        rows.append({"n": n, "steps": steps, "us_per_step": 1e6 * dt / steps, "pairs_per_s": rate,
                     "frac_fp32_roofline": rate / 7.865e12, "variant": sim.variant})
        print("N=2^%-2d %8d  %-28s %10.1f us/step  %.3e pairs/s  %5.1f %%" % (k, n, sim.variant, 1e6 * dt / steps, rate,
                                                                               100 * rate / 7.865e12), flush=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "size_scan.json"), "w"), indent=1)
