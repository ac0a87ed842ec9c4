#!/usr/bin/env python3
"""What ONE rank of a g-rank partition spends in its force pass, timed on one GPU (nb_force_pass: no communicator needed): the rank
form of the symmetric pass (both phases + nb_sym_reduce) and the ordered-pair i-shard, for rank 0, a middle rank and the last one,
each after 0.4 s of warm-up (a cold measurement reads 10-15 % slow: the clocks ramp).
    python tools/rank_pass.py N g [g ...]"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, capi, ic  # noqa: E402

n = int(sys.argv[1])
b, v = ic.plummer(n, seed=1)


def settled(sim, est_ms):
    """force_pass after >= 0.4 s of the same work (the clocks ramp over tens of milliseconds: the first passes of a fresh process
    read 10-15 % slow), best of three windows of >= 50 ms."""
    reps = max(2, int(50.0 / est_ms))
    t = sim.force_pass(max(2, int(400.0 / est_ms)))
    return min(sim.force_pass(reps) for _ in range(3))


with Simulation(n, flags=capi.NB_FLAG_NO_FUSE) as whole:
    whole.init(b, v); whole.set_params(1e-3, 1.0)
    t1 = settled(whole, n * n / 6.5e9)
    print("N=%d  whole system %-44s force pass %10.3f ms" % (n, whole.variant, t1), flush=True)
for g in [int(a) for a in sys.argv[2:]]:
    rows = n // g
    assert rows * g == n and rows % 1024 == 0
    for r in sorted({0, g // 2, g - 1}):
        for flags, label in ((capi.NB_FLAG_SYM_SHARD, "rank form"), (capi.NB_FLAG_NO_SYM, "i-shard  ")):
            with Simulation(n, shard=(r * rows, rows), flags=flags) as sim:
                sim.init(b, v); sim.set_params(1e-3, 1.0)
                t = settled(sim, n * n / 6.5e9 / g * (1.0 if flags == capi.NB_FLAG_SYM_SHARD else 1.4))
                print("N=%d  g=%d rank %d %s %-46s force pass %10.3f ms  = 1/%.2f of the whole system's" % (n, g, r, label, sim.variant, t, t1 / t), flush=True)
