import sys, time
sys.path.insert(0, "nbody3d-webgpu_amd")
from nbody3d_amd import Simulation, capi, ic
for n in (13000, 14000, 15000, 16384, 18000, 20000, 22000):
    b, v = ic.plummer(n, seed=1)
    cands = [("sgpr_auto", dict(flags=capi.NB_FLAG_NO_FUSE)), ("jpk_ws8_r", dict(force_variant=601018, jsplit=max(1, round(n / 2048)))),
             ("jpk_ws4_r", dict(force_variant=601014, jsplit=max(1, round(n / 1024)))), ("jpk_ws8_r2", dict(force_variant=601018, jsplit=max(1, round(n / 1400))))]
    sims = []
    for name, kw in cands:
        s = Simulation(n, **kw); s.init(b, v); s.simulate(64, 1e-3, 1.0); s.sync(); sims.append((name, s))
    steps = max(32, int(2e10 / (n * n)) // 16 * 16)
    best = {}
    for rep in range(3):
        for name, s in sims:
            t0 = time.perf_counter(); s.simulate(steps); s.sync(); dt = (time.perf_counter() - t0) / steps
            best[name] = min(best.get(name, 1e9), dt)
    print(n, "  ".join("%s[%s] %.2f" % (name, s.variant.split("_", 2)[-1], 1e6 * best[name]) for name, s in sims), flush=True)
    for _, s in sims: s.close()
