#!/usr/bin/env python3
"""BASELINE.json config 5: long-horizon energy conservation, fp32 vs fp64 engine,
same Plummer initial conditions.  Energy is computed on the device in fp64
(nb_diagnostics): KE of the velocity after call n paired with PE of the
positions before call n.  Writes one JSON document."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=262144)
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--every", type=int, default=100)
ap.add_argument("--dt", type=float, default=1e-3)
ap.add_argument("--workload", default="plummer", choices=["plummer", "galaxy"],
                help="galaxy: the reference's default UI state (index.html:68-74): 2 x 20,000 + 2 bodies, G = dt = 1e-4 (--n, --dt ignored)")
ap.add_argument("--frames", type=int, default=0, help="also request a viewer frame every this many steps (f4 frame feed under load)")
ap.add_argument("--out", default="gpurun_out/energy_horizon.json")
args = ap.parse_args()

if args.workload == "galaxy":
    b, v, gp = ic.reference_galaxies(os.path.join(ROOT, "tests", "golden", "galaxy40002_params.json"))
    G, args.dt, args.n = float(gp["G"]), 1e-4, b.shape[0]
    label = "reference default: 2 galaxies x 20,000 + central masses 1e7 (index.html:68-74)"
else:
    b, v = ic.plummer(args.n, seed=1)
    G, label = 1.0, "Plummer sphere seed=1"
res = {"n": args.n, "dt": args.dt, "G": G, "steps": args.steps, "workload": label, "runs": {}}
final = {}
for prec in ("f32", "f64"):
    dt_np = np.float64 if prec == "f64" else np.float32
    with Simulation(args.n, precision=prec) as sim:
        sim.init(b.astype(dt_np), v.astype(dt_np))
        sim.set_params(args.dt, G)
        ke0, pe0, _ = sim.diagnostics()
        e0 = ke0 + pe0
        samples = [{"step": 0, "E": e0, "dE_rel": 0.0}]
        t0 = time.perf_counter()
        done = 0
        frames = 0
        while done < args.steps:
            k = min(args.every, args.steps - done)
            todo = k - 1
            while todo > 0:
                m = min(args.frames or todo, todo)
                sim.simulate(m)
                todo -= m
                if args.frames:
                    sim.request_frame()
                    frames += 1
            _, pe_prev, _ = sim.diagnostics()
            sim.step()
            ke, _, mom = sim.diagnostics()
            done += k
            e = ke + pe_prev
            samples.append({"step": done, "E": e, "dE_rel": abs((e - e0) / e0), "momentum_abs": float(np.abs(mom).max())})
            print(prec, done, "dE/E0 = %.3e" % samples[-1]["dE_rel"], flush=True)
        sim.sync()
        wall = time.perf_counter() - t0
        final[prec] = sim.read(vel=False, accel=False)[0]
        res["runs"][prec] = {"variant": sim.variant, "E0": e0, "wall_s_incl_diagnostics": wall, "samples": samples,
                             "max_dE_rel": max(s["dE_rel"] for s in samples), "frames_requested": frames,
                             "finite": bool(np.isfinite(final[prec]).all())}
d = np.abs(final["f32"][:, :3].astype(np.float64) - final["f64"][:, :3]).max(1)
r = np.sqrt((final["f64"][:, :3] ** 2).sum(1))
res["f32_vs_f64_engine_max_rel_pos_err"] = float((d / np.maximum(r, 1.0)).max())
res["f32_vs_f64_engine_median_rel_pos_err"] = float(np.median(d / np.maximum(r, 1.0)))
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(res, open(args.out, "w"), indent=1)
print(json.dumps({k: (v if k != "runs" else {p: {"max_dE_rel": q["max_dE_rel"], "wall_s": q["wall_s_incl_diagnostics"]} for p, q in v.items()}) for k, v in res.items()}))
