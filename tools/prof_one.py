#!/usr/bin/env python3
"""One system, a handful of steps, for rocprofv3 counter passes:  python3 tools/prof_one.py N [flags] [variant] [jsplit] [steps]"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

n = int(sys.argv[1])
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
jsplit = int(sys.argv[4]) if len(sys.argv) > 4 else 0
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 40
b, v = ic.plummer(n, seed=1)
with Simulation(n, flags=flags, force_variant=variant, jsplit=jsplit) as sim:
    sim.init(b, v)
    sim.enable_timing(True)          # plain launches (no graph): one kernel record per step
    sim.simulate(steps, 1e-3, 1.0)
    print(sim.variant, sim.step_breakdown())
