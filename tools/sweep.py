#!/usr/bin/env python3
"""A/B sweep of force-kernel launch shapes on one GPU, interleaved rounds in ONE
process (cdna_hip_programming.md rule 24).  Prints force-kernel time per step
(HIP events on the engine stream) and pairs/s for each (variant, jsplit)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=262144)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--configs", default="2:0,22:0,24:0,28:0")
ap.add_argument("--shard", type=int, default=1, help="own 1/shard of the rows (multi-GPU rank shape)")
ap.add_argument("--precision", default="f32")
args = ap.parse_args()

n = args.n
b, v = ic.plummer(n, seed=1)
if args.precision == "f64":
    b, v = b.astype(np.float64), v.astype(np.float64)
cfgs = [tuple(int(x) for x in (c.split(":") + ["0"])[:3]) for c in args.configs.split(",")]   # variant:jsplit[:flags]
sims = []
for var, js, fl in cfgs:
    s = Simulation(n, precision=args.precision, force_variant=var, jsplit=js, flags=fl,
                   shard=None if args.shard == 1 else (0, n // args.shard))
    s.init(b, v)
    s.set_params(1e-3, 1.0)
    s.simulate(2)
    s.sync()
    s.enable_timing(True)
    sims.append(s)
res = {i: [] for i in range(len(sims))}
for r in range(args.rounds):
    for i, s in enumerate(sims):
        s.simulate(args.steps)
        f, g, c = s.kernel_times()
        res[i].append((f, g))
rows = n // args.shard
print("N=%d rows=%d steps/round=%d rounds=%d" % (n, rows, args.steps, args.rounds))
print("%-34s %10s %10s %12s %8s %10s" % ("variant", "K1 ms min", "K1 ms med", "pairs/s(min)", "%roof", "K2 us"))
for i, s in enumerate(sims):
    f = sorted(x[0] for x in res[i])
    g = sorted(x[1] for x in res[i])
    pairs = rows * (n - 1)
    rate = pairs / (f[0] * 1e-3)
    roof = 157.3e12 / 20 * (0.5 if args.precision == "f64" else 1.0)
    print("%-34s %10.3f %10.3f %12.4e %8.2f %10.1f" % (s.variant + ("" if not cfgs[i][2] else "_flags%d" % cfgs[i][2]), f[0], f[len(f) // 2], rate, 100 * rate / roof, 1e3 * g[0]))
    s.close()
