#!/usr/bin/env python3
"""The rank form of the symmetric pass on ONE GPU: g virtual shards (nb_multi, peer-copy reduce-scatter and all-gather, all on
device 0) against the single handle.  The g shards' kernels share the chip, so the wall time of a step is the sum of the
ranks' work plus the rank form's extra kernels and copies: t(g) / t(1) - 1 is the overhead a g-GPU node pays per rank
(before its xGMI transfers), and t(g) / g bounds a rank's step from above.

    python tools/rank_form_scan.py 262144 2,4,8
"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import MultiSimulation, Simulation, capi, ic  # noqa: E402

n = int(sys.argv[1])
gs = [int(x) for x in sys.argv[2].split(",")]
b, v = ic.plummer(n, seed=1)
est = max(n * n / 5.5e12, 4e-6)
steps = max(4, int(0.2 / est))


def timed(sim):
    sim.init(b, v)
    sim.simulate(steps, 1e-3, 1.0)
    sim.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        sim.simulate(steps)
        sim.sync()
        best = min(best, (time.perf_counter() - t0) / steps)
    return best


with Simulation(n) as one:
    t1 = timed(one)
    print("N=%d  single handle %-40s %10.2f us  %5.1f %%" % (n, one.variant, 1e6 * t1, 100 * n * (n - 1) / t1 / 7.865e12), flush=True)
with Simulation(n, flags=capi.NB_FLAG_NO_SYM) as old:
    t0 = timed(old)
    print("N=%d  ordered-pair kernels %-34s %10.2f us  %5.1f %%" % (n, old.variant, 1e6 * t0, 100 * n * (n - 1) / t0 / 7.865e12), flush=True)
for g in gs:
    with MultiSimulation(n, g) as ms:
        t = timed(ms)
        print("N=%d  g=%d virtual shards %-36s %10.2f us  %5.1f %%   x%.3f of the single handle; per rank <= %.2f us" % (
            n, g, ms.variant, 1e6 * t, 100 * n * (n - 1) / t / 7.865e12, t / t1, 1e6 * t / g), flush=True)
