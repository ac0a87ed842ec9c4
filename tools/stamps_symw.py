#!/usr/bin/env python3
"""Wave timelines of the symmetric force pass at the reference's sizes, from the diagnostic build (`make -C nbody3d-webgpu_amd/csrc stamps`:
-DNB_STAMPS puts s_memtime stamps at the phase boundaries of nb_force_symw; the product build has none).  Per wave: entry, wave table
read, residents + first travelers landed, first (part of a) sweep done, sweeps done, workgroup met, everything stored; the launch
span and the start / end skew come from s_memrealtime (100 MHz).
    NB_ENGINE_LIB=nbody3d-webgpu_amd/csrc/libnbody3d_hip_stamps.so python tools/stamps_symw.py [N ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("NB_ENGINE_LIB", os.path.join(ROOT, "nbody3d-webgpu_amd", "csrc", "libnbody3d_hip_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, capi, ic  # noqa: E402

L = capi.load_library()
L.nb_debug_stamps.argtypes = [C.c_void_p, C.c_uint32]
assert L.nb_debug_stamps(None, 0) == 0          # allocates the buffer and points the kernels at it
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8192, 13000, 16384, 40002]
for n in sizes:
    b, v = ic.plummer(n, seed=1)
    with Simulation(n) as sim:
        if "symw" not in sim.variant:
            print("N=%d: %s is not the wave-granular symmetric pass" % (n, sim.variant))
            continue
        sim.init(b, v)
        sim.simulate(200, 1e-3, 1.0)            # warm: clocks, caches
        sim.sync()
        W = capi.plan_query(n)["plan"]["W"]
        for rep in range(3):
            sim.simulate(1)
            sim.sync()
            st = np.zeros((W, 16), np.uint64)
            assert L.nb_debug_stamps(st.ctypes.data_as(C.c_void_p), W) == 0
        t = st.astype(np.int64)
        ok = t[:, 4] > 0
        t = t[ok]
        t0 = t[:, 0]
        real0, real1 = t[:, 5], t[:, 6]
        span_us = (real1.max() - real0.min()) / 100.0
        clk = np.median((t[:, 4] - t[:, 0]) / np.maximum(1, (real1 - real0))) / 10.0      # GHz: shader cycles per 10 ns tick of s_memrealtime
        def seg(a, b):
            d = (t[:, b] - t[:, a]) / (clk * 1e3)       # us
            return "%5.2f / %5.2f / %5.2f" % (np.percentile(d, 10), np.median(d), np.percentile(d, 90))
        print("N=%6d %s  waves %d  clock %.2f GHz  launch span %.2f us; start skew %.2f us, end skew (p10..max) %.2f us" % (
            n, sim.variant, len(t), clk, span_us, (real0.max() - real0.min()) / 100.0, (real1.max() - np.percentile(real1, 10)) / 100.0))
        print("   us per wave (p10 / median / p90): entry->table %s | table->loads landed %s | first sweep part %s | rest of the sweeps %s | wait for the workgroup %s | combine + stores drained %s | whole wave %s" % (
            seg(0, 1), seg(1, 2), seg(2, 3), seg(3, 8), seg(8, 9), seg(9, 4), seg(0, 4)), flush=True)
        # when the waves start and end, by wave number (= 4 * workgroup + wave: the order the plan lays the ranges out in) and by XCD
        widx = np.nonzero(ok)[0]
        start = (real0 - real0.min()) / 100.0
        end = (real1 - real0.min()) / 100.0
        oct_ = (widx * 8 // (widx.max() + 1))
        xcc = (t[:, 7] >> 32) & 0xf
        print("   start / end (us after the first wave's start) by eighth of the wave numbers: " + "  ".join("%.2f/%.2f" % (start[oct_ == o].mean(), end[oct_ == o].mean()) for o in range(8)))
        print("   ... by XCD: " + "  ".join("x%d %.2f/%.2f" % (x, start[xcc == x].mean(), end[xcc == x].mean()) for x in sorted(set(xcc.tolist()))), flush=True)
        # the clock every XCD ran at (shader cycles per 10 ns tick over the wave's life), the time its waves were alive, and whether workgroup b ran on XCD b % 8
        ghz = (t[:, 4] - t[:, 0]) / np.maximum(1, (real1 - real0)) / 10.0
        life = (real1 - real0) / 100.0
        print("   ... clock GHz / wave life us by XCD: " + "  ".join("x%d %.3f/%.2f" % (x, np.median(ghz[xcc == x]), life[xcc == x].mean()) for x in sorted(set(xcc.tolist())))
              + "   workgroup b on XCD b %% 8: %.1f %%" % (100.0 * np.mean(xcc == (widx // 4) % 8)), flush=True)
