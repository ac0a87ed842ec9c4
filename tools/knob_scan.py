#!/usr/bin/env python3
"""Arms of the launch planner's model constants (tuning build: NB_MODEL_* read at nb_create), kept alive together and timed in turns
(NB_ROUNDS rounds, best of each: a drift of the chip's clock hits all of them alike).  Wall time per step under graph replay.
An arm is a comma-separated list of NAME=value (NB_MODEL_ prefix implied), arms separated by ';':
    NB_ARMS='ODD_XCD=1,XCD_START=0;ODD_XCD=0.978,XCD_START=1' python tools/knob_scan.py [f64] [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("NB_ENGINE_LIB", os.path.join(ROOT, "nbody3d-webgpu_amd", "csrc", "libnbody3d_hip_tuning.so"))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8192, 16384, 40002, 262144]
arms = [a.strip() for a in os.environ.get("NB_ARMS", "OLD_SHARE=0.5;OLD_SHARE=0").split(";") if a.strip()]
rounds = int(os.environ.get("NB_ROUNDS", "5"))
prec = "f64" if "f64" in sys.argv[1:] else "f32"
roof = 7.865e12 if prec == "f32" else 3.93e12
for n in sizes:
    b, v = ic.plummer(n, seed=1)
    steps = max(4, int(0.25 / max(n * n / (0.8 * roof), 4e-6)) // 16 * 16)
    sims, ref, touched = [], None, set()
    for arm in arms:
        for name in touched:
            os.environ.pop(name, None)
        for kv in arm.split(","):
            k, val = kv.split("=")
            os.environ["NB_MODEL_" + k] = val
            touched.add("NB_MODEL_" + k)
        sim = Simulation(n, precision=prec)
        sim.init(b, v)
        sim.simulate(1, 1e-3, 1.0)
        acc = sim.read(bodies=False, vel=False)[2]
        ref = acc if ref is None else ref
        err = float(np.abs(acc[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max())
        sim.simulate(steps)
        sim.sync()
        sims.append((arm, sim, err))
    best = {a: 1e30 for a in arms}
    for _ in range(rounds):
        for a, sim, err in sims:
            t0 = time.perf_counter(); sim.simulate(steps); sim.sync()
            best[a] = min(best[a], time.perf_counter() - t0)
    for a, sim, err in sims:
        us = 1e6 * best[a] / steps
        print("N=%7d %-28s %-38s %10.2f us/step %5.1f %%  %+5.2f %% vs the first arm  acc vs first arm %.1e" % (
            n, a, sim.variant, us, 100 * n * (n - 1) / (us * 1e-6) / roof, 100 * (best[arms[0]] / best[a] - 1), err), flush=True)
        sim.close()
