#!/usr/bin/env python3
"""A longer randomised differential run than tests/test_fuzz_gpu.py carries (same checks, many more seeds, larger systems,
the planner's own choice as well as pinned shapes, single and multi-shard handles, both precisions): one force evaluation per
case against the fp64 oracle.  Needs a GPU; prints one line per failure and a summary.  usage: fuzz_campaign.py [cases] [seed0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
sys.path.insert(0, ROOT)
from nbody3d_amd import MultiSimulation, Simulation  # noqa: E402
from oracle import oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ORDERED = [0, 1, 2, 4, 14, 116, 164, 22, 24, 28, 34, 38, 304014, 308014, 402644, 601014, 601018]
SYM = [704013, 708013, 708011, 716013, 716011, 708014]


def system(rng, n):
    b = np.zeros((n, 4), np.float32)
    b[:, :3] = rng.normal(size=(n, 3)) * rng.choice([0.1, 1.0, 30.0])
    b[:, 3] = rng.random(n) * rng.choice([1e-3, 1.0, 1e4]) + (0 if rng.random() < 0.3 else 1e-6)
    if n > 3 and rng.random() < 0.3:
        b[rng.integers(n), 3] = 0.0
        b[rng.integers(n)] = b[rng.integers(n)]
    v = np.zeros((n, 4), np.float32)
    v[:, :3] = rng.normal(size=(n, 3)) * 0.1
    return b, v


fails, t0, kinds = 0, time.time(), {}
for c in range(cases):
    rng = np.random.default_rng(seed0 + c)
    kind = rng.choice(["auto", "ordered", "sym", "multi"], p=[0.35, 0.2, 0.3, 0.15])
    f64 = bool(rng.random() < 0.25)
    eps2 = float(rng.choice([1e-4, 1e-6, 2.5e-3]))
    G = float(rng.choice([1.0, 1e-4, 7.5]))
    tag = dict(case=seed0 + c, kind=str(kind), f64=f64, eps2=eps2, G=G)
    try:
        if kind == "multi":
            g = int(rng.choice([2, 3, 4, 5, 8]))
            n = int(rng.integers(300, 30000))
            b, v = system(rng, n)
            dt_np = np.float64 if f64 else np.float32
            with MultiSimulation(n, g, precision="f64" if f64 else "f32", eps2=eps2) as ms:
                tag.update(n=n, g=g, name=ms.variant)
                ms.init(b.astype(dt_np), v.astype(dt_np))
                ms.simulate(1, 1e-4, G)
                bb, vv, aa = ms.read()
        else:
            n = int(rng.choice([rng.integers(1, 600), rng.integers(600, 9000), rng.integers(9000, 60000)]))
            variant = 0 if kind == "auto" else int(rng.choice(ORDERED if kind == "ordered" else SYM))
            if f64 and kind == "sym":
                variant = 708013
            jsplit = int(rng.choice([0, 0, 1, 2, 3, 5, 8])) if variant else 0
            b, v = system(rng, n)
            dt_np = np.float64 if f64 else np.float32
            with Simulation(n, eps2=eps2, force_variant=variant, jsplit=jsplit, precision="f64" if f64 else "f32") as sim:
                tag.update(n=n, variant=variant, jsplit=jsplit, name=sim.variant)
                sim.init(b.astype(dt_np), v.astype(dt_np))
                sim.simulate(1, 1e-4, G)
                bb, vv, aa = sim.read()
        ra = oracle.accel_f64(b.astype(np.float64), G, eps2=eps2)
        scale = max(float(np.abs(ra[:, :3]).max()), 1e-300)
        err = float(np.abs(aa[:, :3] - ra[:, :3]).max() / scale)
        tol = 1e-11 if f64 else 2e-5
        ok = np.isfinite(aa).all() and np.isfinite(bb).all() and err <= tol and np.all(aa[:, 3] == 0) and np.array_equal(bb[:, 3], b[:, 3].astype(dt_np))
        kinds[tag["name"].split("_js")[0].split("_w")[0]] = kinds.get(tag["name"].split("_js")[0].split("_w")[0], 0) + 1
        if not ok:
            fails += 1
            print("FAIL", tag, "err=%.3e tol=%.1e" % (err, tol), flush=True)
    except Exception as e:                       # an engine error is a finding too
        fails += 1
        print("ERROR", tag, repr(e), flush=True)
    if (c + 1) % 50 == 0:
        print("... %d cases, %d failures, %.0f s" % (c + 1, fails, time.time() - t0), flush=True)
print("fuzz campaign: %d cases from seed %d, %d failures, %.0f s; kernel forms seen: %s" % (cases, seed0, fails, time.time() - t0, kinds))
sys.exit(1 if fails else 0)
