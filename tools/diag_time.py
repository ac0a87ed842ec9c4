#!/usr/bin/env python3
"""nb_diagnostics: wall time per call and agreement with the oracle's fp64 host energy (one call must not
cost more than one force step).    python tools/diag_time.py [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402
from oracle import oracle  # noqa: E402  (the checker)

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [3000, 40002, 65536, 262144]
for n in sizes:
    b, v = ic.plummer(n, seed=5)
    rke, rpe, rmom = oracle.energy(b, v, 1.0) if n <= 70000 else (None, None, None)
    for prec in ("f32", "f64"):
        dt = np.float64 if prec == "f64" else np.float32
        with Simulation(n, precision=prec) as sim:
            sim.init(b.astype(dt), v.astype(dt))
            sim.set_params(1e-3, 1.0)
            sim.simulate(2)
            sim.sync()
            t0 = time.perf_counter(); sim.simulate(3); sim.sync(); step_ms = 1e3 * (time.perf_counter() - t0) / 3
            sim.init(b.astype(dt), v.astype(dt))
            ke, pe, mom = sim.diagnostics()
            best = 1e30
            for _ in range(3):
                t0 = time.perf_counter(); sim.diagnostics(); best = min(best, time.perf_counter() - t0)
            line = "N=%7d %s  diagnostics %8.3f ms per call (a step: %8.3f ms)  KE %.12g PE %.12g" % (n, prec, 1e3 * best, step_ms, ke, pe)
            if rke is not None:
                line += "  vs host fp64: dKE %.2e dPE %.2e" % (abs(ke - rke) / abs(rke), abs(pe - rpe) / abs(rpe))
            print(line, flush=True)
