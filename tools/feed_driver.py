#!/usr/bin/env python3
"""N=40,002, one step per frame, optional viewer snapshot every frame: per-frame wall time (for rocprofv3 traces too)."""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, ic  # noqa: E402
n, frames = 40002, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
feed = len(sys.argv) > 2 and sys.argv[2] in ("feed", "feedacq")
acq = len(sys.argv) > 2 and sys.argv[2] == "feedacq"
b, v = ic.uniform_cube(n, seed=62)
with Simulation(n) as sim:
    sim.init(b, v); sim.set_params(1e-4, 1e-4)
    for _ in range(50): sim.step()
    sim.sync()
    for rep in range(3):
        t0 = time.perf_counter()
        landed = 0
        for _ in range(frames):
            sim.step()
            if feed:
                sim.request_frame()
            if acq and sim.frame(wait=False) is not None:
                landed += 1
        sim.sync()
        print(("feed+acquire(landed %d)" % landed) if acq else "feed" if feed else "plain", "%.2f us/frame" % (1e6 * (time.perf_counter() - t0) / frames), sim.variant, flush=True)
