#!/usr/bin/env python3
"""One rank's force pass for a kernel trace:  python3 tools/rank_pass_one.py N g rank [reps]"""
import os
import sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, capi, ic  # noqa: E402
n, g, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
b, v = ic.plummer(n, seed=1)
rows = n // g
with Simulation(n, shard=(r * rows, rows), flags=capi.NB_FLAG_SYM_SHARD) as sim:
    sim.init(b, v); sim.set_params(1e-3, 1.0)
    print(sim.variant, sim.force_pass(reps))
