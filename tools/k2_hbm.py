#!/usr/bin/env python3
"""The integrate kernel (K2) on its own at sizes where the state is no longer cache-resident
(N >= 4M: 4 arrays x 16 B x N > the 256 MiB Infinity Cache), SURVEY.md §7.2 / §8(d) "K2 roofline".
A full O(N^2) step at these sizes takes seconds to minutes, so the kernel is timed through
nb_integrate_pass (the handle's own K2, launched back to back on its stream, HIP events).
jsplit = 1: K2 reads x, v, a_old and the single force array, writes x, v and the a_old/a_new
buffers swap by pointer = 96 B per body, exactly the algorithmic bytes.

    python tools/k2_hbm.py [N ...]        # default 4194304 8388608 16777216
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, capi  # noqa: E402

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1 << 22, 1 << 23, 1 << 24]
reps = 30
rows = []
for n in sizes:
    rng = np.random.default_rng(7)
    b = rng.random((n, 4), dtype=np.float32)
    v = np.zeros((n, 4), np.float32)
    for js in (1, 8):
        with Simulation(n, force_variant=208011, jsplit=js, flags=capi.NB_FLAG_NO_FUSE) as sim:
            sim.init(b, v)
            sim.set_params(1e-3, 1.0)
            ms = min(sim.integrate_pass(reps) for _ in range(3))
            name = sim.variant
        moved = 96.0 if js == 1 else 96.0 + 16.0 * js
        row = {"n": n, "kernel_variant": name, "jsplit": js, "avg_launch_ms": ms, "algorithmic_bytes": 96 * n,
               "algorithmic_GBps": 96.0 * n / (ms * 1e-3) / 1e9, "moved_GBps": moved * n / (ms * 1e-3) / 1e9,
               "frac_of_8TBps_algorithmic": 96.0 * n / (ms * 1e-3) / 8e12}
        rows.append(row)
        print("N=%9d  %-26s  K2 %8.3f ms  algorithmic %7.1f GB/s (%.1f %% of 8 TB/s)  moved %7.1f GB/s" % (
            n, name, ms, row["algorithmic_GBps"], 100 * row["frac_of_8TBps_algorithmic"], row["moved_GBps"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "k2_hbm.json"), "w"), indent=1)
