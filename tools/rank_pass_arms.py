#!/usr/bin/env python3
"""One rank's force pass under a few plan arms:  python3 tools/rank_pass_arms.py N g"""
import os
import sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, capi, ic  # noqa: E402
n, g = int(sys.argv[1]), int(sys.argv[2])
b, v = ic.plummer(n, seed=1)
rows = n // g
with Simulation(n, flags=capi.NB_FLAG_NO_FUSE) as whole:
    whole.init(b, v); whole.set_params(1e-3, 1.0)
    t1 = whole.force_pass(5)
    print("whole", whole.variant, "%.3f ms; / %d = %.3f" % (t1, g, t1 / g), flush=True)
for label, kw in (("default", {}), ("whole sweeps", dict(flags=capi.NB_FLAG_WHOLE_SWEEPS)), ("1 wave/SIMD", dict(jsplit=1)), ("1 wave/SIMD whole", dict(jsplit=1, flags=capi.NB_FLAG_WHOLE_SWEEPS)),
                  ("3 waves/SIMD", dict(jsplit=3))):
    f = kw.pop("flags", 0)
    with Simulation(n, shard=(0, rows), flags=capi.NB_FLAG_SYM_SHARD | f, **kw) as sim:
        sim.init(b, v); sim.set_params(1e-3, 1.0)
        ts = [sim.force_pass(5) for _ in range(3)]
        print("%-18s %-48s %.3f %.3f %.3f ms" % (label, sim.variant, *ts), flush=True)
