#!/usr/bin/env python3
"""Runs `steps` steps of one size / variant (for rocprofv3 --kernel-trace: kernel time vs gaps)."""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402

n, steps = int(sys.argv[1]), int(sys.argv[2])
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
jsplit = int(sys.argv[4]) if len(sys.argv) > 4 else 0
b, v = ic.plummer(n, seed=1)
with Simulation(n, force_variant=variant, jsplit=jsplit) as sim:
    sim.init(b, v)
    sim.simulate(steps, 1e-3, 1.0)
    sim.sync()
    sim.simulate(steps)
    sim.sync()
    print(sim.variant)
