"""GPU parity tests: the HIP engine, called through the C ABI, against the
oracle and the committed golden vectors.  Tolerances are stated here.

  TOL_POS   = 1e-4  BASELINE.json: "positions within 1e-4 rel of reference after
                    100 steps" (metric: conftest.rel_pos_err)
  TOL_TIGHT = 2e-5  what we actually hold the f32 engine to against the fp64
                    oracle (the fp32 oracle itself sits at ~9e-7; the engine
                    differs from it by fma contraction, v_rsq_f32 (1 ulp) and
                    the j-split summation order)
  TOL_ACC   = 2e-5  single force evaluation, relative to max |a|
"""
import numpy as np
import pytest

from conftest import load_golden32, load_golden64, rel_pos_err
from oracle import oracle
from nbody3d_amd import MultiSimulation, Simulation, ic

pytestmark = pytest.mark.gpu

TOL_POS, TOL_TIGHT, TOL_ACC = 1e-4, 2e-5, 2e-5

# (force_variant, jsplit): every kernel family, with and without a j-split.  Short codes are the
# ABI-1 names; 6-digit codes are K II LL X (include/nbody3d_hip.h): K = 2 packed LDS-tile force
# kernel, 3 packed SGPR-broadcast force kernel (X = waves splitting j), 4 fused one-launch step.
VARIANTS = [(0, 0), (1, 1), (1, 3), (2, 1), (2, 2), (4, 1), (4, 4), (14, 1), (14, 2), (116, 1), (164, 1), (164, 2),
            (22, 1), (22, 3), (24, 1), (24, 2), (28, 1), (28, 2), (34, 1), (34, 3), (38, 1), (38, 2),
            (202021, 1), (202164, 2), (202644, 1), (204081, 3), (204324, 1), (208161, 2), (208644, 1),
            (304014, 1), (304014, 3), (308014, 1), (308014, 2), (308015, 1), (308015, 3),
            (402011, 0), (402161, 0), (402644, 0), (404041, 0), (404324, 0), (408011, 0), (408161, 0), (408644, 0)]


def run_engine(b, v, dt, G, steps, a=None, **kw):
    with Simulation(b.shape[0], **kw) as sim:
        sim.init(b, v, a)
        sim.simulate(steps, dt, G)
        out = sim.read()
        name = sim.variant
    return out + (name,)


@pytest.mark.parametrize("variant,jsplit", VARIANTS)
def test_single_step_matches_oracle_acceleration(variant, jsplit):
    """accel after one call == oracle accelerations of the initial positions."""
    b, v = ic.plummer(2048, seed=11)
    bb, vv, aa, name = run_engine(b, v, 1e-3, 1.0, 1, force_variant=variant, jsplit=jsplit)
    ref = oracle.accel_f64(b, 1.0)
    scale = np.abs(ref[:, :3]).max()
    err = np.abs(aa[:, :3] - ref[:, :3]).max() / scale
    assert err < TOL_ACC, (name, err)
    assert np.all(aa[:, 3] == 0) and np.all(vv[:, 3] == 0)
    assert np.array_equal(bb[:, 3], b[:, 3])           # masses untouched
    # and the integrator: same update applied to the engine's own accelerations
    b2, v2, a2 = oracle.run_f32(b, v, None, 1e-3, 1.0, 1)
    assert rel_pos_err(bb, b2, 1.0) < 1e-6
    assert np.abs(vv - v2).max() < 1e-5 * np.abs(v2).max() + 1e-7


@pytest.mark.parametrize("name,steps", [("plummer1024", 100), ("cube1000", 20), ("disk771", 50), ("galaxy_ref", 30)])
@pytest.mark.parametrize("variant,jsplit", [(0, 0), (2, 1), (1, 2), (164, 1), (22, 1), (24, 2), (28, 1), (38, 2),
                                            (308014, 2), (304015, 3), (402644, 0), (404161, 0), (408041, 0), (601014, 1), (601018, 3), (601016, 2)])
def test_golden_trajectories(manifest, name, steps, variant, jsplit):
    """BASELINE.json config 1 (Plummer N=1024, dt=1e-3, 100 steps) and the
    ragged-N / harsh-mass-ratio fixtures, against fp64 and fp32 oracle vectors."""
    m = manifest[name]
    b0, v0 = load_golden32(name + "_bodies0"), load_golden32(name + "_vel0")
    bb, vv, aa, vname = run_engine(b0, v0, m["dt"], m["G"], steps, force_variant=variant, jsplit=jsplit)
    ref64 = load_golden64("%s_s%d_bodies" % (name, steps))
    ref32 = load_golden32("%s_s%d_bodies" % (name, steps))
    e64 = rel_pos_err(bb, ref64, m["r_scale"])
    e32 = rel_pos_err(bb, ref32, m["r_scale"])
    assert e64 < TOL_TIGHT < TOL_POS, (vname, e64)
    assert e32 < TOL_TIGHT, (vname, e32)
    a32 = load_golden32("%s_s%d_accel" % (name, steps))
    assert np.abs(aa[:, :3] - a32[:, :3]).max() < 1e-4 * np.abs(a32[:, :3]).max()
    v32 = load_golden32("%s_s%d_vel" % (name, steps))
    assert np.abs(vv[:, :3] - v32[:, :3]).max() < 1e-4 * np.abs(v32[:, :3]).max()


def test_mid_size_trajectory_against_fp64_oracle():
    """N=8,192 Plummer, 20 steps, packed kernel with a j-split, against the fp64 oracle
    run live (1.3e9 pair evaluations on the host cores)."""
    n, steps = 8192, 20
    b, v = ic.plummer(n, seed=23)
    bb, vv, aa, name = run_engine(b, v, 1e-3, 1.0, steps, force_variant=24, jsplit=8)
    rb, rv, ra = oracle.run_f64(b, v, None, 1e-3, 1.0, steps)
    assert rel_pos_err(bb, rb, 1.0) < TOL_TIGHT, name
    assert np.abs(aa[:, :3] - ra[:, :3]).max() < TOL_ACC * np.abs(ra[:, :3]).max(), name
    assert np.abs(vv[:, :3] - rv[:, :3]).max() < TOL_TIGHT * max(np.abs(rv[:, :3]).max(), 1.0), name


def test_intermediate_checkpoints_and_restore(manifest):
    """read() then restore() mid-run (util.js:163-178 / :230-244 round trip)
    continues exactly as an uninterrupted run."""
    m = manifest["plummer1024"]
    b0, v0 = load_golden32("plummer1024_bodies0"), load_golden32("plummer1024_vel0")
    with Simulation(1024) as sim:
        sim.init(b0, v0)
        sim.simulate(10, m["dt"], m["G"])
        b10, v10, a10 = sim.read()
        assert rel_pos_err(b10, load_golden32("plummer1024_s10_bodies"), m["r_scale"]) < 1e-6
        sim.simulate(15)
        straight = sim.read()
    with Simulation(1024) as sim2:
        sim2.restore(b10, v10, a10)
        sim2.simulate(15, m["dt"], m["G"])
        resumed = sim2.read()
    for x, y in zip(straight, resumed):
        assert x.tobytes() == y.tobytes()


def test_graph_replay_equals_single_steps():
    """nb_step(k >= 16) replays captured HIP graphs (16 and 128 steps; fused one-launch steps
    ping-pong the position buffers, so a graph is only valid for the buffer parity it was
    captured at); it must be bit-identical to k single-step calls, and follow dt/G
    changes between calls (the graph bakes them in and is re-captured)."""
    b, v = ic.plummer(1024, seed=18)
    with Simulation(1024) as a, Simulation(1024) as c:
        a.init(b, v)
        c.init(b, v)
        a.simulate(40, 1e-3, 1.0)          # 2 graph launches + 8 plain steps
        for _ in range(40):
            c.step(1e-3, 1.0)
        a.simulate(35, 5e-4, 0.5)          # new params: re-capture
        for _ in range(35):
            c.step(5e-4, 0.5)
        a.simulate(300)                    # 2 x 128-step graph + 2 x 16-step graph + 12 plain, odd parity start
        for _ in range(300):
            c.step()
        ra, rc = a.read(), c.read()
    for x, y in zip(ra, rc):
        assert x.tobytes() == y.tobytes()


def test_dt_zero_and_negative_are_noops():
    """`if (dt > 0)` gate, nbody3d.js:474: state stays bit-identical."""
    b, v = ic.plummer(512, seed=12)
    a = np.random.default_rng(0).random((512, 4)).astype(np.float32)
    with Simulation(512) as sim:
        sim.init(b, v, a)
        sim.simulate(3, 0.0, 1.0)
        sim.step(-1e-3)
        bb, vv, aa = sim.read()
    assert bb.tobytes() == b.tobytes() and vv.tobytes() == v.tobytes() and aa.tobytes() == a.tobytes()


def test_first_step_uses_zero_accel_when_none_uploaded():
    """accelBuffer starts zeroed (nbody3d.js:195-199): vel_1 = v0 + dt/2 * a0."""
    b, v = ic.uniform_cube(768, seed=13)
    bb, vv, aa, _ = run_engine(b, v, 1e-2, 1.0, 1)
    exp = v[:, :3] + 0.5 * 1e-2 * aa[:, :3]
    assert np.abs(vv[:, :3] - exp).max() < 1e-6 * max(np.abs(exp).max(), 1e-3)


@pytest.mark.parametrize("n", [1, 2, 63, 255, 257, 1000, 4099])
def test_ragged_sizes(n):
    """N not a multiple of 64/256 (the reference is undefined there, SURVEY.md
    §3.4); here it is bounds-guarded.  N=1: no pairs, body drifts freely."""
    rng = np.random.default_rng(n)
    b = np.zeros((n, 4), np.float32)
    b[:, :3] = rng.random((n, 3)) * 2 - 1
    b[:, 3] = rng.random(n) + 0.5
    v = np.zeros((n, 4), np.float32)
    v[:, :3] = rng.random((n, 3)) - 0.5
    bb, vv, aa, name = run_engine(b, v, 1e-3, 0.01, 3)
    rb, rv, ra = oracle.run_f32(b, v, None, 1e-3, 0.01, 3)
    assert rel_pos_err(bb, rb, 1.0) < 1e-6, name
    assert np.abs(aa - ra).max() <= 2e-5 * max(np.abs(ra).max(), 1e-30), name


@pytest.mark.parametrize("n", [9, 263, 1001, 4099])
@pytest.mark.parametrize("variant", [34, 38, 304014, 308014, 308015])
def test_ragged_sizes_on_the_sgpr_kernel(n, variant):
    """The SGPR-broadcast kernel walks j in batches of 8 with a scalar remainder loop:
    N not a multiple of 8 (and splits that end mid-batch) must still match the oracle."""
    rng = np.random.default_rng(n)
    b = np.zeros((n, 4), np.float32)
    b[:, :3] = rng.random((n, 3)) * 2 - 1
    b[:, 3] = rng.random(n) + 0.5
    v = np.zeros((n, 4), np.float32)
    bb, vv, aa, name = run_engine(b, v, 1e-3, 0.01, 2, force_variant=variant, jsplit=3)
    assert "sgpr" in name
    rb, rv, ra = oracle.run_f32(b, v, None, 1e-3, 0.01, 2)
    assert rel_pos_err(bb, rb, 1.0) < 1e-6, name
    assert np.abs(aa - ra).max() <= 2e-5 * max(np.abs(ra).max(), 1e-30), name


def test_coincident_bodies_and_zero_mass():
    """Self term / coincident pairs contribute exactly 0 (eps2 > 0); zero-mass
    bodies exert nothing but still move."""
    b = np.zeros((256, 4), np.float32)
    b[:128, 3] = 1.0                       # 128 massive bodies all at the origin
    b[128:, 0] = np.linspace(1, 2, 128)    # 128 massless tracers
    v = np.zeros((256, 4), np.float32)
    bb, vv, aa, _ = run_engine(b, v, 1e-3, 1.0, 1)
    assert np.all(aa[:128] == 0)           # coincident: exactly zero
    ref = oracle.accel_f64(b, 1.0)
    assert np.allclose(aa[128:, :3], ref[128:, :3], rtol=2e-5, atol=1e-7)


def test_custom_eps2_and_zero_G():
    """nb_config.eps2 replaces the hard-coded 1e-4 (nbody3d.js:234); G = 0 leaves pure drift."""
    b, v = ic.plummer(600, seed=19)
    for eps2 in (1e-6, 2.5e-3):
        _, _, aa, name = run_engine(b, v, 1e-3, 1.0, 1, eps2=eps2)
        ref = oracle.accel_f64(b, 1.0, eps2=eps2)
        assert np.abs(aa[:, :3] - ref[:, :3]).max() < TOL_ACC * np.abs(ref[:, :3]).max(), (name, eps2)
    bb, vv, aa, _ = run_engine(b, v, 1e-2, 0.0, 3)
    assert np.all(aa == 0) and vv.tobytes() == v.tobytes()
    rb, _, _ = oracle.run_f32(b, v, None, 1e-2, 0.0, 3)
    assert bb.tobytes() == rb.tobytes()


def test_far_pairs_overflow_to_zero_like_the_reference():
    """distSqr^3 overflows binary32 beyond ~2.6e6 length units; inverseSqrt(inf) = 0,
    so such pairs contribute exactly 0 in the reference arithmetic (nbody3d.js:235).
    Same here (v_rsq_f32(inf) = 0), no NaNs."""
    b = np.zeros((512, 4), np.float32)
    b[:256, :3] = np.random.default_rng(0).random((256, 3))
    b[256:, :3] = np.random.default_rng(1).random((256, 3)) + 1e7     # a second cluster 1e7 away
    b[:, 3] = 1.0
    v = np.zeros((512, 4), np.float32)
    bb, vv, aa, _ = run_engine(b, v, 1e-3, 1.0, 1)
    ra = oracle.accel_f32(b, 1.0)
    assert np.isfinite(aa).all() and np.isfinite(bb).all()
    # each cluster only feels itself
    own = oracle.accel_f32(b[:256], 1.0)
    assert np.abs(ra[:256] - own).max() == 0
    assert np.abs(aa[:256, :3] - own[:, :3]).max() < TOL_ACC * np.abs(own[:, :3]).max()


def test_invalid_arguments_raise():
    b, v = ic.plummer(256, seed=20)
    with Simulation(256) as sim:
        with pytest.raises(Exception) as e:
            sim.simulate(1, 1e-3, 1.0)          # step before init: NB_ERR_STATE
        assert "NB_ERR_STATE" in str(e.value)
        with pytest.raises(ValueError):
            sim.init(b[:100], v)                # wrong length caught before any copy
        sim.init(b, v)
        with pytest.raises(Exception) as e:
            sim.set_params(float("nan"), 1.0)
        assert "NB_ERR_INVALID" in str(e.value)
    with pytest.raises(Exception) as e:
        Simulation(256, tile=128)               # only the reference's 256 tile is built
    assert "NB_ERR_INVALID" in str(e.value)


def test_newton_third_law_on_device():
    b, v = ic.plummer(4096, seed=14)
    _, _, aa, _ = run_engine(b, v, 1e-3, 1.0, 1)
    f = (b[:, 3:4].astype(np.float64) * aa[:, :3]).sum(0)
    scale = np.abs(b[:, 3:4] * aa[:, :3]).sum(0)
    assert np.all(np.abs(f) < 2e-6 * scale)


def test_shards_on_one_device_compose_to_the_single_handle_result():
    """SURVEY.md §8(e) 'virtual shard' check: g handles on ONE GPU, each owning
    an i-block, host-side gather of rows between steps == one unsharded handle
    (bit for bit when the launch shape is pinned)."""
    n, g, steps = 2048, 4, 5
    b, v = ic.plummer(n, seed=15)
    kw = dict(force_variant=1, jsplit=2)
    with Simulation(n, **kw) as one:
        one.init(b, v)
        one.simulate(steps, 1e-3, 1.0)
        ref = one.read()
    per = n // g
    sims = [Simulation(n, shard=(r * per, per), **kw) for r in range(g)]
    try:
        for s in sims:
            s.init(b, v)
            s.set_params(1e-3, 1.0)
        bodies = b.copy()
        for _ in range(steps):
            rows = []
            for r, s in enumerate(sims):
                s.step()
                rows.append(s.read(vel=False, accel=False)[0][r * per:(r + 1) * per])
            bodies = np.concatenate(rows)
            for r, s in enumerate(sims):
                bb, vv, aa = s.read()
                s.restore(bodies, vv, aa)
        vel = np.zeros((n, 4), np.float32)
        acc = np.zeros((n, 4), np.float32)
        for r, s in enumerate(sims):
            _, vv, aa = s.read()
            vel[r * per:(r + 1) * per] = vv[r * per:(r + 1) * per]
            acc[r * per:(r + 1) * per] = aa[r * per:(r + 1) * per]
    finally:
        for s in sims:
            s.close()
    assert bodies.tobytes() == ref[0].tobytes()
    assert vel.tobytes() == ref[1].tobytes() and acc.tobytes() == ref[2].tobytes()


@pytest.mark.parametrize("g,variant", [(2, 22), (4, 22), (2, 38), (4, 28)])
def test_overlapped_exchange_with_virtual_shards(g, variant):
    """nb_set_exchange_overlapped on ONE GPU: g shard handles, each with its own
    torch-owned bodies buffer; wait() copies the other shards' rows (as they
    were before this step) on the engine's stream -- what the all-gather does.
    The own-rows-first / rest-after-wait split must reproduce the unsharded
    handle bit for bit."""
    import torch
    n, steps = 4096, 6
    per = n // g
    b, v = ic.plummer(n, seed=22)
    kw = dict(force_variant=variant, jsplit=8)        # j_per_split = 512 divides the shard rows
    with Simulation(n, **kw) as one:
        one.init(b, v)
        one.simulate(steps, 1e-3, 1.0)
        ref = one.read()
    stream = torch.cuda.current_stream().cuda_stream
    bufs = [torch.empty((n, 4), device="cuda", dtype=torch.float32) for _ in range(g)]
    sims = [Simulation(n, shard=(r * per, per), stream=stream, ext_bodies=bufs[r].data_ptr(), **kw) for r in range(g)]
    snap = {}
    calls = {"begin": 0, "wait": 0}
    try:
        for r, s in enumerate(sims):
            s.init(b, v)
            s.set_params(1e-3, 1.0)

            def begin(ptr, esz, nn, sb, sc, st, r=r):
                calls["begin"] += 1
                return 0

            def wait(st, r=r):
                calls["wait"] += 1
                for q in range(g):
                    if q != r:
                        bufs[r][q * per:(q + 1) * per].copy_(snap[q])   # rows of rank q after the previous step
                return 0

            s.set_exchange_overlapped(begin, wait)
        for k in range(steps):
            # what every rank's rows look like after step k-1 (gathered during step k)
            snap = {q: bufs[q][q * per:(q + 1) * per].clone() for q in range(g)}
            for s in sims:
                s.step()
        for s in sims:
            s.sync()                                   # finishes the last pending "gather"
        bodies = np.concatenate([bufs[r][r * per:(r + 1) * per].cpu().numpy() for r in range(g)])
        vel = np.zeros((n, 4), np.float32)
        for r, s in enumerate(sims):
            vel[r * per:(r + 1) * per] = s.read(bodies=False, accel=False)[1][r * per:(r + 1) * per]
    finally:
        for s in sims:
            s.close()
    assert calls["begin"] == g * steps and calls["wait"] == g * steps
    assert bodies.tobytes() == ref[0].tobytes()
    assert vel.tobytes() == ref[1].tobytes()


@pytest.mark.parametrize("n,g", [(2048, 2), (4096, 4), (4096, 8), (1000, 3)])
def test_single_process_multi_shard_handle(n, g):
    """nb_multi_*: g shards in one process with the peer-copy all-gather.  With more
    shards than GPUs they share the device ("virtual shards"), which
    exercises exactly the partition / event / copy logic a real 8-GPU node runs.
    Bit-identical to one unsharded handle when the launch shape is pinned; ragged n
    is padded with zero-mass rows."""
    steps = 7
    b, v = (ic.plummer(n, seed=24) if n % 256 == 0 else ic.uniform_cube(n, seed=24))
    a0 = np.random.default_rng(5).random((n, 4)).astype(np.float32) * 0.01
    a0[:, 3] = 0
    kw = dict(force_variant=1, jsplit=2)
    with MultiSimulation(n, g, **kw) as ms:
        ms.init(b, v, a0)
        ms.simulate(3, 1e-3, 1.0)
        for _ in range(steps - 3):
            ms.step()
        got = ms.read()
        name = ms.variant
    if n % 256 == 0:
        with Simulation(n, **kw) as one:
            one.init(b, v, a0)
            one.simulate(steps, 1e-3, 1.0)
            ref = one.read()
        for x, y in zip(got, ref):
            assert x.tobytes() == y.tobytes(), name
    rb, rv, ra = oracle.run_f32(b, v, a0, 1e-3, 1.0, steps)
    assert rel_pos_err(got[0], rb, 1.0) < 1e-6, name
    assert np.abs(got[2] - ra).max() < TOL_ACC * np.abs(ra).max(), name
    assert np.array_equal(got[0][:, 3], b[:, 3])


def test_multi_handle_default_shape_and_noop():
    b, v = ic.plummer(8192, seed=25)
    with MultiSimulation(8192, 4) as ms:
        ms.init(b, v)
        ms.simulate(2, 0.0, 1.0)                      # dt = 0: no-op (nbody3d.js:474)
        assert ms.read()[0].tobytes() == b.tobytes()
        ms.simulate(5, 1e-3, 1.0)
        got = ms.read()
        ke, pe, mom = ms.diagnostics()
    rke, rpe, _ = oracle.energy(got[0], got[1], 1.0)
    assert abs(ke - rke) < 1e-9 * abs(rke) and abs(pe - rpe) < 1e-6 * abs(rpe)
    rb, _, _ = oracle.run_f64(b, v, None, 1e-3, 1.0, 5)
    assert rel_pos_err(got[0], rb, 1.0) < TOL_TIGHT


def test_exchange_hook_is_called_once_per_step():
    b, v = ic.plummer(512, seed=16)
    calls = []
    with Simulation(512, shard=(0, 256)) as sim:
        sim.init(b, v)
        sim.set_exchange(lambda ptr, esz, n, sb, sc, stream: calls.append((esz, n, sb, sc)) or 0)
        sim.simulate(4, 1e-3, 1.0)
        sim.sync()
        assert calls == [(4, 512, 0, 256)] * 4
        sim.set_exchange(lambda *a: 7)
        with pytest.raises(Exception) as e:
            sim.step()
        assert "NB_ERR_COMM" in str(e.value)


def test_f64_engine_matches_f64_oracle(manifest):
    """BASELINE.json config 5 (fp64 variant): same trajectory as the fp64 oracle."""
    m = manifest["plummer1024"]
    b0 = load_golden32("plummer1024_bodies0").astype(np.float64)
    v0 = load_golden32("plummer1024_vel0").astype(np.float64)
    bb, vv, aa, name = run_engine(b0, v0, m["dt"], m["G"], 100, precision="f64")
    assert bb.dtype == np.float64
    assert rel_pos_err(bb, load_golden64("plummer1024_s100_bodies"), m["r_scale"]) < 1e-12, name


def test_diagnostics_match_host_energy():
    b, v = ic.plummer(3000, seed=17)
    with Simulation(3000) as sim:
        sim.init(b, v)
        sim.set_params(1e-3, 1.0)
        ke, pe, mom = sim.diagnostics()
    rke, rpe, rmom = oracle.energy(b, v, 1.0)
    assert abs(ke - rke) < 1e-9 * abs(rke) and abs(pe - rpe) < 1e-6 * abs(rpe)
    assert np.abs(mom - rmom).max() < 1e-9


def test_energy_drift_reported_and_small(manifest):
    """Energy bookkeeping of SURVEY.md §8(c): KE(vel after call n) with
    PE(positions before call n)."""
    m = manifest["plummer1024"]
    b0, v0 = load_golden32("plummer1024_bodies0"), load_golden32("plummer1024_vel0")
    ke0, pe0, _ = oracle.energy(b0, v0, m["G"])
    with Simulation(1024) as sim:
        sim.init(b0, v0)
        sim.simulate(99, m["dt"], m["G"])
        bprev = sim.read()[0]
        sim.step()
        _, vk, _ = sim.read()
    ke, _, _ = oracle.energy(bprev, vk, m["G"])
    _, pe, _ = oracle.energy(bprev, vk, m["G"])
    drift = abs((ke + pe - (ke0 + pe0)) / (ke0 + pe0))
    assert drift < 5 * m["energy_drift"]["f64"] + 1e-6


@pytest.mark.parametrize("n,variant", [(65536, 0), (65536, 28), (65536, -1), (262144, 0), (262144, 208011), (262144, -1), (1048576, 0)])
def test_full_size_properties(n, variant):
    """BASELINE.json configs 2, 3 and 4 (N=1,048,576) at full size, through size-independent
    properties (the oracle would take minutes): a sampled i-slice against the
    fp64 oracle, Newton's third law, and shard-composition (row blocks of a
    1/8 shard handle equal the full handle's rows).  variant 28 / 208011 pin config 2's
    "LDS tile=256" kernel (the default shape at these sizes streams j through SGPRs)."""
    b, v = (ic.uniform_cube(n, seed=2) if n == 65536 else ic.plummer(n, seed=1))
    from nbody3d_amd import capi
    # variant 0: the default shape (the symmetric pass at these sizes); -1: the default among the ordered-pair kernels
    with Simulation(n, force_variant=max(variant, 0), flags=capi.NB_FLAG_NO_SYM if variant < 0 else 0) as sim:
        sim.init(b, v)
        sim.simulate(1, 1e-3, 1.0)
        bb, vv, aa = sim.read()
        name = sim.variant
    if variant > 0:
        assert "pk_lds256" in name, name
    elif variant < 0:
        assert "sgpr" in name, name
    else:
        assert "symw" in name, name
    rows = np.random.default_rng(0).choice(n, 96, replace=False)
    rows.sort()
    b64 = b.astype(np.float64)
    for i in rows[:: 96 // 24]:
        ref = oracle.accel_f64(b64, 1.0, i0=int(i), i1=int(i) + 1)[0, :3]
        assert np.abs(aa[i, :3] - ref).max() < TOL_ACC * max(np.abs(ref).max(), 1e-3), (name, i)
    f = (b[:, 3:4].astype(np.float64) * aa[:, :3]).sum(0)
    assert np.all(np.abs(f) < 1e-5 * np.abs(b[:, 3:4] * aa[:, :3]).sum(0))
    per = n // 8
    with Simulation(n, shard=(3 * per, per)) as sh:
        sh.init(b, v)
        sh.simulate(1, 1e-3, 1.0)
        sb, sv, sa = sh.read()
        sname = sh.variant
    blk = slice(3 * per, 4 * per)
    assert rel_pos_err(sb[blk], bb[blk], 1.0) < 1e-6, sname
    assert np.abs(sa[blk] - aa[blk]).max() < TOL_ACC * np.abs(aa[blk]).max(), sname
