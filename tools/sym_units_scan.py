#!/usr/bin/env python3
"""The symmetric pass with wave ranges cut in whole / quarter / eighth sweeps (force_variant 7 II LL 3, LL = units per sweep),
1 or 2 waves per SIMD: wall time per step under graph replay, and the accelerations against the whole-sweep arm.
    python tools/sym_units_scan.py [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [13000, 16384, 20000, 24000, 32768, 40002, 65536]
for n in sizes:
    b, v = ic.plummer(n, seed=1)
    est = max(n * n / 4.5e12, 3.5e-6)
    steps = max(16, int(0.25 / est) // 16 * 16)
    ref = None
    arms = [("auto", dict()), ("auto whole", dict(flags=256))]
    for ipl in (4, 8, 16):
        if n <= 64 * ipl * 4:
            continue
        for k in (1, 2):
            for ups in (1, 8, 32):
                arms.append(("ipl%d k%d u%d" % (ipl, k, ups), dict(force_variant=700003 + ipl * 1000 + ups * 10, jsplit=k)))
    rows = []
    for name, kw in arms:
        with Simulation(n, **kw) as sim:
            sim.init(b, v)
            sim.simulate(1, 1e-3, 1.0)
            acc = sim.read(bodies=False, vel=False)[2]
            if ref is None:
                ref = acc
            err = float(np.abs(acc[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max())
            sim.simulate(max(16, int(0.2 / est)))
            sim.sync()
            best = 1e30
            for _ in range(2):
                t0 = time.perf_counter(); sim.simulate(steps); sim.sync(); best = min(best, time.perf_counter() - t0)
            us = 1e6 * best / steps
            rows.append((us, name, sim.variant, err))
            print("N=%7d %-14s %-40s %9.2f us/step %5.1f %%  acc vs first arm %.1e" % (n, name, sim.variant, us, 100 * n * (n - 1) / (us * 1e-6) / 7.865e12, err), flush=True)
    rows.sort()
    print("N=%7d BEST %s %s %.2f us   (auto: %.2f us)" % (n, rows[0][1], rows[0][2], rows[0][0], [r for r in rows if r[1] == "auto"][0][0]), flush=True)
