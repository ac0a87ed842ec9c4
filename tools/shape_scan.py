#!/usr/bin/env python3
"""Wall time per step (simulate(k) + sync, graph replay included) of every plausible launch shape
at a list of sizes: what the planner's cost model (csrc/nb_plan.cpp) is fitted to and checked
against.  One process, interleaved repeats (cdna_hip_programming.md rule 24).

    python tools/shape_scan.py 1024 4096 16384 40002 65536 [--quick]
"""
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import Simulation, capi, ic  # noqa: E402

quick = "--quick" in sys.argv
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024, 4096, 16384, 40002, 65536]
ROOF = 7.865e12
N_CU = 256


def candidates(n):
    out = [("auto", dict()), ("auto_nofuse", dict(flags=capi.NB_FLAG_NO_FUSE))]
    for ipl in (2, 4, 8):
        for ls in (1, 2, 4, 8, 16, 32, 64):
            for tl in (1, 4, 8):
                if (tl == 4 and ls < 16) or (tl == 8 and ls < 32):
                    continue
                wgs = -(-n // ((256 // ls) * ipl))
                if wgs < 96 or wgs > 8192:
                    continue
                if n / ls < 16:
                    continue
                out.append(("fused_ipl%d_ls%d_tl%d" % (ipl, ls, tl), dict(force_variant=400000 + ipl * 1000 + ls * 10 + tl)))
    if n <= 2048:
        out.append(("direct_regs2048", dict(force_variant=502642)))
    if n <= 1024:
        out.append(("direct_regs1024", dict(force_variant=502641)))
    if n >= 16384:
        for ipl in (4, 8):
            for ws in (1, 4):
                for js in (1, 2, 4, 8, 16, 32, 64):
                    ipb = (256 // ws) * ipl
                    wgs = -(-n // ipb) * js
                    if wgs < 512 or wgs > 8192 or n / js / ws < 256:
                        continue
                    out.append(("sgpr_ipl%d_ws%d_js%d" % (ipl, ws, js), dict(force_variant=300000 + ipl * 1000 + 10 + ws, jsplit=js)))
    if n >= 2048:          # j-packed fused step: 64 i-bodies per workgroup of ws waves, q splits across workgroups
        for x, ws in ((4, 4), (8, 8), (6, 16)):
            for js in (1, 2, 3, 4, 5, 6, 8, 12, 16):
                wgs = -(-n // 64) * js
                if wgs < 128 or wgs > 8192 or n / js / ws < 64:
                    continue
                out.append(("jpk_ws%d_js%d" % (ws, js), dict(force_variant=601010 + x, jsplit=js)))
    return out


rows = []
for n in sizes:
    b, v = ic.plummer(n, seed=1)
    steps = max(16, min(2048, int(3e10 / (n * n)) // 16 * 16))
    cands = candidates(n)
    sims = []
    for name, kw in cands:
        try:
            s = Simulation(n, **kw)
        except Exception as e:
            print("skip %s: %s" % (name, e))
            continue
        s.init(b, v)
        s.simulate(32, 1e-3, 1.0)
        s.sync()
        sims.append((name, s))
    best = {}
    for rep in range(2 if quick else 3):
        for name, s in sims:
            t0 = time.perf_counter()
            s.simulate(steps)
            s.sync()
            dt = (time.perf_counter() - t0) / steps
            best[name] = min(best.get(name, 1e9), dt)
    print("\n== N=%d (%d steps per timing) ==" % (n, steps))
    for name, s in sorted(sims, key=lambda t: best[t[0]]):
        us = 1e6 * best[name]
        rate = n * (n - 1) / best[name]
        print("%-26s %-34s %10.2f us/step  %.3e pairs/s  %5.1f %%" % (name, s.variant, us, rate, 100 * rate / ROOF), flush=True)
        rows.append({"n": n, "config": name, "variant": s.variant, "us_per_step": us, "frac": rate / ROOF})
    for _, s in sims:
        s.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "shape_scan.json"), "w"), indent=1)
