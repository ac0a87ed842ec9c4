#!/usr/bin/env python3
"""The symmetric force pass (force_variant 7 08 01 WS) against the number of chunk-list segments per super-block
(jsplit): wall time per step (graph replay) and the force / integrate kernel times (HIP events), beside the default shape.

    python tools/sym_sweep.py 40002 2,4,8 24,32,44,64,88
"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

n = int(sys.argv[1])
wss = [int(x) for x in sys.argv[2].split(",")]
qs = [int(x) for x in sys.argv[3].split(",")]
b, v = ic.plummer(n, seed=1)
est = max(n * n / 5.5e12, 4e-6)
steps = max(16, int(0.15 / est) // 16 * 16)
cfgs = [(0, 0)] + [(708010 + ws, q) for ws in wss for q in qs]
sims = []
for var, q in cfgs:
    s = Simulation(n, force_variant=var, jsplit=q)
    s.init(b, v)
    s.simulate(steps, 1e-3, 1.0)
    s.sync()
    sims.append((var, q, s, []))
for r in range(3):
    for var, q, s, t in sims:
        t0 = time.perf_counter()
        s.simulate(steps)
        s.sync()
        t.append((time.perf_counter() - t0) / steps)
print("N=%d steps/timing=%d" % (n, steps))
for var, q, s, t in sims:
    s.enable_timing(True)
    s.simulate(min(steps, 40))
    f, g, c = s.kernel_times()
    s.enable_timing(False)
    best = min(t)
    print("%7d q=%3d %-34s %9.2f us %5.1f %%   K1 %8.2f us  K2 %7.2f us" % (var, q, s.variant, 1e6 * best, 100 * n * (n - 1) / best / 7.865e12,
                                                                            1e3 * f, 1e3 * g), flush=True)
    s.close()
