#!/usr/bin/env python3
"""Where a step's time goes at mid sizes: wall time per step under graph replay (what the user gets) next to the force kernel,
the integrate kernel and the span of a timed step (nb_step_times2), and the planner's ideal (sweeps x sweep time).
    python tools/step_parts.py [N ...] [--variant V] [--flags F]"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

args = sys.argv[1:]
kw = {}
if "--variant" in args:
    kw["force_variant"] = int(args[args.index("--variant") + 1])
if "--flags" in args:
    kw["flags"] = int(args[args.index("--flags") + 1])
if "--jsplit" in args:
    kw["jsplit"] = int(args[args.index("--jsplit") + 1])
if "--precision" in args:
    kw["precision"] = args[args.index("--precision") + 1]
sizes = [int(a) for a in args if a.isdigit() and args[max(0, args.index(a) - 1)] not in ("--variant", "--flags", "--jsplit", "--precision")] or [13000, 16384, 20000, 24000, 32768, 40002, 65536]
for n in sizes:
    b, v = ic.plummer(n, seed=1)
    if kw.get("precision") == "f64":
        b, v = b.astype("float64"), v.astype("float64")
    est = max(n * n / 4.5e12, 3.5e-6)
    steps = max(16, int(0.3 / est) // 16 * 16)
    with Simulation(n, **kw) as sim:
        sim.init(b, v)
        sim.simulate(max(16, int(0.25 / est)), 1e-3, 1.0)
        sim.sync()
        best = 1e30
        for _ in range(2):
            t0 = time.perf_counter(); sim.simulate(steps); sim.sync(); best = min(best, time.perf_counter() - t0)
        wall = 1e6 * best / steps
        sim.enable_timing(True)
        sim.simulate(64)
        t = sim.step_breakdown()
        sim.enable_timing(False)
        print("N=%7d %-36s wall %8.2f us/step (%5.1f %%) | timed steps: force %8.2f  integrate %6.2f  span %8.2f us" % (
            n, sim.variant, wall, 100 * n * (n - 1) / (wall * 1e-6) / (7.865e12 * (0.5 if kw.get('precision') == 'f64' else 1.0)), 1e3 * t["force_ms"], 1e3 * t["integrate_ms"], 1e3 * t["span_ms"]), flush=True)
