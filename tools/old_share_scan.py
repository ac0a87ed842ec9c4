#!/usr/bin/env python3
"""Two waves per SIMD, unequal ranges: the OLDER wave of every SIMD (first half of the workgroups: dispatched first, wins the issue
arbitration -- tools/stamps_symw.py) given a share `a` of a pair of ranges, the younger wave 1 - a.  Tuning build (NB_MODEL_OLD_SHARE,
read at nb_create; 0.5 = equal ranges, waves in list order).  Wall time per step under graph replay; the arms stay alive and are timed in
turns (NB_ROUNDS rounds, best of each), so a drift of the chip's clock hits all of them alike.
    NB_SHARES=0.5,0.8,0.9 python tools/old_share_scan.py [f64] [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("NB_ENGINE_LIB", os.path.join(ROOT, "nbody3d-webgpu_amd", "csrc", "libnbody3d_hip_tuning.so"))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [40002, 65536, 131072, 262144]
shares = [float(x) for x in os.environ.get("NB_SHARES", "0.5,0.7,0.8,0.9").split(",")]
rounds = int(os.environ.get("NB_ROUNDS", "4"))
prec = "f64" if "f64" in sys.argv[1:] else "f32"
roof = 7.865e12 if prec == "f32" else 3.93e12
for n in sizes:
    b, v = ic.plummer(n, seed=1)
    est = n * n / (0.8 * roof)
    steps = max(4, int(0.25 / est))
    sims, ref = [], None
    for a in shares:
        os.environ["NB_MODEL_OLD_SHARE"] = "%g" % a
        sim = Simulation(n, precision=prec)
        sim.init(b, v)
        sim.simulate(1, 1e-3, 1.0)
        acc = sim.read(bodies=False, vel=False)[2]
        ref = acc if ref is None else ref
        err = float(np.abs(acc[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max())
        sim.simulate(steps)
        sim.sync()
        sims.append((a, sim, err))
    best = {a: 1e30 for a in shares}
    for _ in range(rounds):
        for a, sim, err in sims:
            t0 = time.perf_counter(); sim.simulate(steps); sim.sync()
            best[a] = min(best[a], time.perf_counter() - t0)
    for a, sim, err in sims:
        us = 1e6 * best[a] / steps
        print("N=%7d share %.2f %-36s %10.2f us/step %5.1f %%  %+5.2f %% vs the first arm  acc vs first arm %.1e" % (
            n, a, sim.variant, us, 100 * n * (n - 1) / (us * 1e-6) / roof, 100 * (best[shares[0]] / best[a] - 1), err), flush=True)
        sim.close()
