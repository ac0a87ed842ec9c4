"""One random API call sequence on a handle and on the CPU oracle side by side (used by tests/test_fuzz_gpu.py and by
tools/fuzz_api.py for long campaigns): simulate(k) across the graph-replay thresholds, dt / G changes, pause (dt = 0), read,
snapshot + restore, viewer frames -- compared after every read.  Exercises what the engine keeps BETWEEN calls: buffer parity of
the fused steps, captured graphs, the (x, y, z, G m) j-stream copy, the pair-transposed copy, the frame slots."""
import numpy as np

from nbody3d_amd import MultiSimulation, Simulation, ic
from oracle import oracle

VARIANTS = [0, 0, 0, 2, 22, 28, 34, 304014, 402644, 502641, 601014, 704013, 704013, 708013, 708011]


def run_sequence(seed, n_max=2500):
    """Raises AssertionError (with the call log) on a mismatch; returns (variant name, log)."""
    rng = np.random.default_rng(seed)
    f64 = bool(rng.random() < 0.2)
    multi = bool(rng.random() < 0.15)
    n = int(rng.integers(300, n_max))
    variant = 0 if multi else int(rng.choice(VARIANTS))
    if f64 and variant not in (0, 2, 708013):
        variant = 708013 if n > 512 else 0
    dt_np = np.float64 if f64 else np.float32
    run = oracle.run_f64 if f64 else oracle.run_f32
    tol = 1e-10 if f64 else 2e-5
    b, v = ic.plummer(n, seed=int(rng.integers(1 << 30)))
    mb, mv, ma = b.astype(dt_np), v.astype(dt_np), np.zeros((n, 4), dt_np)          # the model's state
    dt, G = 1e-3, 1.0
    log, snap, msnap = [], None, None
    ctx = MultiSimulation(n, int(rng.choice([2, 3, 4])), precision="f64" if f64 else "f32") if multi else \
        Simulation(n, precision="f64" if f64 else "f32", force_variant=variant, jsplit=int(rng.choice([0, 0, 2, 3])) if variant else 0)
    with ctx as sim:
        name = sim.variant
        where = lambda: "seed %d n=%d f64=%s %s: %s" % (seed, n, f64, name, " ".join(log))
        sim.init(mb, mv)
        sim.set_params(dt, G)
        steps_total = 0
        for _ in range(int(rng.integers(4, 12))):
            op = rng.choice(["sim", "sim", "sim", "params", "pause", "read", "snap", "restore", "frame"])
            if op == "sim" and steps_total < 70:
                k = int(rng.choice([1, 1, 2, 3, 15, 16, 17, 33]))
                sim.simulate(k)
                mb, mv, ma = run(mb, mv, ma, dt, G, k)
                steps_total += k
                log.append("sim%d" % k)
            elif op == "params":
                dt, G = float(rng.choice([1e-3, 5e-4, 2e-3])), float(rng.choice([1.0, 1.0, 0.5, 2.5]))
                sim.set_params(dt, G)
                log.append("dt=%g,G=%g" % (dt, G))
            elif op == "pause":
                sim.simulate(int(rng.choice([1, 16, 40])), 0.0, G)          # dt = 0: no dispatch (nbody3d.js:474)
                sim.set_params(dt, G)
                log.append("pause")
            elif op == "snap":
                snap = tuple(x.copy() for x in sim.read())
                msnap = (mb.copy(), mv.copy(), ma.copy())
                log.append("snap")
            elif op == "restore" and snap is not None:
                sim.init(*snap)
                mb, mv, ma = (x.copy() for x in msnap)
                log.append("restore")
            elif op == "frame" and not multi:
                sim.request_frame()
                fb, fs, _ = sim.frame()
                log.append("frame")
                want = np.sqrt((mv[:, :3].astype(np.float64) ** 2).sum(1))
                ftol = max(tol * 10, 2e-7)              # the frame is packed to f32 whatever the handle's precision
                assert np.abs(np.asarray(fb)[:, :3] - mb[:, :3]).max() <= ftol * max(1.0, np.abs(mb[:, :3]).max()), where()
                assert np.abs(np.asarray(fs) - want).max() <= 1e-4 * max(1e-3, want.max()), where()
            else:
                bb, vv, aa = sim.read()
                log.append("read")
                e_pos = float(np.abs(bb[:, :3] - mb[:, :3]).max() / max(1.0, float(np.abs(mb[:, :3]).max())))
                e_vel = float(np.abs(vv[:, :3] - mv[:, :3]).max() / max(1e-3, float(np.abs(mv[:, :3]).max())))
                e_acc = float(np.abs(aa[:, :3] - ma[:, :3]).max() / max(1e-30, float(np.abs(ma[:, :3]).max()))) if steps_total else 0.0
                assert e_pos <= tol and e_vel <= 10 * tol and e_acc <= 10 * tol, where() + " -> pos %.2e vel %.2e acc %.2e" % (e_pos, e_vel, e_acc)
        bb = sim.read()[0]
        e_pos = float(np.abs(bb[:, :3] - mb[:, :3]).max() / max(1.0, float(np.abs(mb[:, :3]).max())))
        assert e_pos <= tol, where() + " -> final pos %.2e" % e_pos
    return name, log
