"""Randomised differential test: random sizes, kernel shapes, split counts, softening and
time steps against the oracle.  Exists because shape-dependent indexing bugs (e.g. the LS=64
chunk rounding that read past the LDS tile) only show up for particular (N, shape, split)
combinations that the hand-picked cases may miss."""
import numpy as np
import pytest

from conftest import rel_pos_err
from oracle import oracle
from nbody3d_amd import MultiSimulation, Simulation

pytestmark = pytest.mark.gpu

VARIANTS = [0, 1, 2, 4, 14, 116, 164, 22, 24, 28, 34, 38]


def random_system(rng, n):
    b = np.zeros((n, 4), np.float32)
    b[:, :3] = rng.normal(size=(n, 3)) * rng.choice([0.1, 1.0, 30.0])
    b[:, 3] = rng.random(n) * rng.choice([1e-3, 1.0, 1e4]) + (0 if rng.random() < 0.3 else 1e-6)
    if n > 3 and rng.random() < 0.3:
        b[rng.integers(n), 3] = 0.0                      # a massless tracer
        b[rng.integers(n)] = b[rng.integers(n)]          # a coincident pair
    v = np.zeros((n, 4), np.float32)
    v[:, :3] = rng.normal(size=(n, 3)) * 0.1
    return b, v


@pytest.mark.parametrize("seed", range(48))
def test_random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 600), rng.integers(600, 5000)]))
    variant = int(rng.choice(VARIANTS))
    jsplit = int(rng.choice([0, 1, 2, 3, 5, 8, 13])) if variant else 0
    eps2 = float(rng.choice([1e-4, 1e-6, 2.5e-3]))
    G = float(rng.choice([1.0, 1e-4, 7.5]))
    dt = float(rng.choice([1e-3, 1e-4]))
    steps = 1      # one force evaluation: later steps of these (deliberately stiff) systems amplify
                   # rounding differences chaotically, which is physics, not a kernel property
    b, v = random_system(rng, n)
    tag = dict(seed=seed, n=n, variant=variant, jsplit=jsplit, eps2=eps2, G=G, dt=dt, steps=steps)
    with Simulation(n, eps2=eps2, force_variant=variant, jsplit=jsplit) as sim:
        sim.init(b, v)
        sim.simulate(steps, dt, G)
        bb, vv, aa = sim.read()
        tag["name"] = sim.variant
    rb, rv, _ = oracle.run_f32(b, v, None, dt, G, steps, eps2=eps2)
    ra = oracle.accel_f64(b.astype(np.float64), G, eps2=eps2)      # accelerations of the initial positions
    scale = max(float(np.abs(ra[:, :3]).max()), 1e-30)
    assert np.isfinite(bb).all() and np.isfinite(aa).all(), tag
    tag["acc_err"] = float(np.abs(aa[:, :3] - ra[:, :3]).max() / scale)
    tag["pos_err"] = rel_pos_err(bb, rb, max(float(np.abs(b[:, :3]).max()), 1e-3))
    assert tag["acc_err"] <= 2e-5, tag          # vs the fp64 oracle
    assert tag["pos_err"] < 1e-5, tag           # vs the fp32 oracle (stiff systems: dt^2 a is large)
    assert np.array_equal(bb[:, 3], b[:, 3]) and np.all(aa[:, 3] == 0), tag


@pytest.mark.parametrize("seed", range(8))
def test_random_multi_shard_case(seed):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.integers(300, 4000))
    g = int(rng.choice([2, 3, 4, 5, 8]))
    b, v = random_system(rng, n)
    with MultiSimulation(n, g) as ms:
        ms.init(b, v)
        ms.simulate(1, 1e-3, 1.0)
        bb, vv, aa = ms.read()
    rb, rv, _ = oracle.run_f32(b, v, None, 1e-3, 1.0, 1)
    ra = oracle.accel_f64(b.astype(np.float64), 1.0)
    assert np.abs(aa[:, :3] - ra[:, :3]).max() <= 2e-5 * max(float(np.abs(ra[:, :3]).max()), 1e-30), (seed, n, g)
    assert rel_pos_err(bb, rb, max(float(np.abs(b[:, :3]).max()), 1e-3)) < 1e-5, (seed, n, g)


SYM_VARIANTS = [704013, 708013, 708011, 716013, 716011, 708014]


@pytest.mark.parametrize("seed", range(40))
def test_random_symmetric_pass_case(seed):
    """The symmetric force pass on random systems: every resident count / traveler count / workgroup form, random waves per SIMD
    (or segment counts), sizes from just above one super-block to 30 of them, ragged N, stiff mass ratios, tracers and coincident
    bodies, random softening and G -- one force evaluation against the fp64 oracle, both precisions."""
    rng = np.random.default_rng(9000 + seed)
    variant = int(rng.choice(SYM_VARIANTS))
    S = 64 * (variant // 1000 % 100) * (4 if variant % 10 == 4 else 1)
    n = int(rng.integers(S + 1, 30 * S)) if rng.random() < 0.8 else int(S * rng.integers(2, 12))
    n = min(n, 30000)
    f64 = variant == 708013 and rng.random() < 0.4
    jsplit = int(rng.choice([0, 1, 2, 3])) if variant % 10 != 4 else int(rng.choice([0, 3, 7, 16]))
    eps2 = float(rng.choice([1e-4, 1e-6, 2.5e-3]))
    G = float(rng.choice([1.0, 1e-4, 7.5]))
    b, v = random_system(rng, n)
    tag = dict(seed=seed, n=n, variant=variant, jsplit=jsplit, eps2=eps2, G=G, f64=f64)
    dt_np = np.float64 if f64 else np.float32
    with Simulation(n, eps2=eps2, force_variant=variant, jsplit=jsplit, precision="f64" if f64 else "f32") as sim:
        sim.init(b.astype(dt_np), v.astype(dt_np))
        sim.simulate(1, 1e-4, G)
        bb, vv, aa = sim.read()
        tag["name"] = sim.variant
    if n > S:
        assert "sym" in tag["name"], tag
    ra = oracle.accel_f64(b.astype(np.float64), G, eps2=eps2)
    scale = max(float(np.abs(ra[:, :3]).max()), 1e-30)
    assert np.isfinite(bb).all() and np.isfinite(aa).all(), tag
    tag["acc_err"] = float(np.abs(aa[:, :3] - ra[:, :3]).max() / scale)
    assert tag["acc_err"] <= (1e-12 if f64 else 2e-5), tag
    assert np.array_equal(bb[:, 3], b[:, 3].astype(dt_np)) and np.all(aa[:, 3] == 0), tag


@pytest.mark.parametrize("seed", range(8))
def test_random_rank_form_case(seed):
    """nb_multi in the rank form (>= 2,048 rows per shard): random shard counts and sizes, f32 and f64."""
    rng = np.random.default_rng(7000 + seed)
    g = int(rng.choice([2, 3, 4, 5, 8]))
    n = int(rng.integers(2048 * g - 500, 2048 * g + 6000))
    f64 = rng.random() < 0.35
    b, v = random_system(rng, n)
    dt_np = np.float64 if f64 else np.float32
    with MultiSimulation(n, g, precision="f64" if f64 else "f32") as ms:
        name = ms.variant
        ms.init(b.astype(dt_np), v.astype(dt_np))
        ms.simulate(1, 1e-4, 1.0)
        bb, vv, aa = ms.read()
    if -(-n // g) >= 2048:
        assert "symwrank" in name, (seed, n, g, name)
    ra = oracle.accel_f64(b.astype(np.float64), 1.0)
    assert np.abs(aa[:, :3] - ra[:, :3]).max() <= (1e-12 if f64 else 2e-5) * max(float(np.abs(ra[:, :3]).max()), 1e-30), (seed, n, g, name)


@pytest.mark.parametrize("seed", range(40000, 40048))
def test_random_api_call_sequence(seed):
    """simulate(k) across the graph-replay thresholds, dt / G changes, pause, read, snapshot + restore, frames -- on a handle
    and on the oracle side by side (tests/api_sequence.py; tools/fuzz_api.py runs the long campaign)."""
    from api_sequence import run_sequence
    run_sequence(seed, n_max=1500)
