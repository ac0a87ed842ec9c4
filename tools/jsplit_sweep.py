#!/usr/bin/env python3
"""Wall time per step (simulate(k) + sync, graph replay) against the NUMBER OF j-PARTITIONS for pinned kernel shapes:
what the round structure of a launch (workgroups / resident slots) costs at mid sizes.

    python tools/jsplit_sweep.py 40002 304014,308014 16:40            # N  variants  js_lo:js_hi[:step]
    python tools/jsplit_sweep.py 16384 304014,601014,601018 4:32:2

One process, every configuration warmed first, then `--rounds` interleaved timing rounds; best of the rounds."""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

n = int(sys.argv[1])
variants = [int(x) for x in sys.argv[2].split(",")]
rng = [int(x) for x in sys.argv[3].split(":")]
js_list = list(range(rng[0], rng[1] + 1, rng[2] if len(rng) > 2 else 1))
rounds = 3
b, v = ic.plummer(n, seed=1)
est = max(n * n / 4.5e12, 4e-6)
steps = max(16, int(0.12 / est) // 16 * 16)
sims = []
for var in variants:
    for js in ([0] + js_list):
        try:
            s = Simulation(n, force_variant=var, jsplit=js)
        except Exception as e:
            print("skip", var, js, e)
            continue
        s.init(b, v)
        s.simulate(steps, 1e-3, 1.0)
        s.sync()
        sims.append((var, js, s, []))
for r in range(rounds):
    for var, js, s, t in sims:
        t0 = time.perf_counter()
        s.simulate(steps)
        s.sync()
        t.append((time.perf_counter() - t0) / steps)
print("N=%d steps/timing=%d rounds=%d" % (n, steps, rounds))
for var, js, s, t in sims:
    info = s.shape_info()
    best = min(t)
    print("%7d js=%3d  %-36s jps=%6d  %9.2f us  %5.1f %%" % (var, js, s.variant, info["j_per_split"], 1e6 * best,
                                                             100 * n * (n - 1) / best / 7.865e12), flush=True)
    s.close()
