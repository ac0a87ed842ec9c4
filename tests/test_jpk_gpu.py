"""GPU tests of the j-packed fused step (nb_step_jpk, force_variant 6 01 01 X), through the C ABI.

The kernel streams j-pairs from a pair-transposed copy of the positions, splits j over the waves of
a workgroup and -- with jsplit > 1 -- over workgroups that meet at a per-i-block ticket inside the
launch (release / acquire at agent scope).  Checked here: the arithmetic against the fp64 oracle, the
in-launch reduction with the partial sums poisoned after use (a stale read would be a NaN), the
pair copy following every way the positions can change (upload, restore, G, raw device pointer), and
that every host-visible entry point follows the live ping-pong buffer.
"""
import numpy as np
import pytest

from conftest import rel_pos_err
from oracle import oracle
from nbody3d_amd import Simulation, capi, ic

pytestmark = pytest.mark.gpu

TOL_ACC = 2e-5
WAVES = {4: 4, 8: 8, 6: 16}          # variant digit X -> waves per workgroup


def code(x):
    return 601010 + x


def run(b, v, steps, dt=1e-3, G=1.0, **kw):
    with Simulation(b.shape[0], **kw) as sim:
        sim.init(b, v)
        sim.simulate(steps, dt, G)
        return sim.read() + (sim.variant,)


@pytest.mark.parametrize("x", [4, 8, 6])
@pytest.mark.parametrize("jsplit", [1, 2, 5])
@pytest.mark.parametrize("n", [1, 2, 63, 129, 1000, 4097])
def test_single_step_and_short_run_match_the_fp64_oracle(x, jsplit, n):
    """Any N (odd, below one workgroup, not a multiple of 64), every workgroup size, with and without the
    split across workgroups: accelerations of the first step and positions after 19 steps (one graph
    replay, both buffer parities)."""
    b, v = (ic.uniform_cube(n, seed=41) if n != 4097 else ic.plummer(n, seed=41))
    bb, vv, aa, name = run(b, v, 1, force_variant=code(x), jsplit=jsplit)
    assert "jpairs_ws%d" % WAVES[x] in name, name
    rb, rv, ra = oracle.run_f64(b, v, None, 1e-3, 1.0, 1)
    scale = np.abs(ra[:, :3]).max() if n > 1 else 1.0
    assert np.abs(aa[:, :3] - ra[:, :3]).max() <= TOL_ACC * scale, name
    assert np.array_equal(bb[:, 3], b[:, 3]) and not aa[:, 3].any()
    bb, vv, aa, name = run(b, v, 19, force_variant=code(x), jsplit=jsplit)
    rb, rv, ra = oracle.run_f64(b, v, None, 1e-3, 1.0, 19)
    assert rel_pos_err(bb, rb, 1.0) < 2e-5, name
    assert np.abs(vv[:, :3] - rv[:, :3]).max() < 1e-4 * max(np.abs(rv[:, :3]).max(), 1e-3), name


@pytest.mark.parametrize("x,jsplit,n", [(4, 7, 5000), (8, 4, 8192), (6, 2, 8192), (4, 16, 3001), (8, 3, 40002)])
def test_in_launch_reduction_with_poisoned_partials(x, jsplit, n):
    """NB_FLAG_POISON: the workgroup that reduces an i-block overwrites the partial sums it has read
    with NaN.  Were a later step to read a partial stale -- out of this XCD's L2 or a CU's L1 instead
    of what the producing workgroup released -- it would read that NaN (or the 0xFF fill of a
    never-written row).  60 steps: finite everywhere and bit-identical to the run without poison,
    and to a second run (the reduction order does not depend on arrival order)."""
    b, v = ic.plummer(n, seed=43) if n % 2 == 0 else ic.uniform_cube(n, seed=43)
    steps = 60 if n < 20000 else 12
    p = run(b, v, steps, force_variant=code(x), jsplit=jsplit, flags=capi.NB_FLAG_POISON)
    q = run(b, v, steps, force_variant=code(x), jsplit=jsplit)
    r = run(b, v, steps, force_variant=code(x), jsplit=jsplit)
    assert "_js%d" % jsplit in p[3], p[3]
    for a in p[:3]:
        assert np.isfinite(a).all(), p[3]
    for a, c, d in zip(p[:3], q[:3], r[:3]):
        assert a.tobytes() == c.tobytes() == d.tobytes(), p[3]


def test_same_arithmetic_as_the_two_kernel_sgpr_step_within_rounding():
    """Against the large-N production kernel on the same state (different summation order only)."""
    n = 16384
    b, v = ic.plummer(n, seed=44)
    jb, jv, ja, jn = run(b, v, 5, force_variant=code(6), jsplit=1)
    sb, sv, sa, sn = run(b, v, 5, force_variant=304014, jsplit=16)
    assert "jpairs" in jn and "sgpr" in sn
    assert rel_pos_err(jb, sb, 1.0) < 2e-6
    assert np.abs(ja[:, :3] - sa[:, :3]).max() < 1e-5 * np.abs(sa[:, :3]).max()


def test_pair_copy_follows_uploads_restores_G_and_raw_pointers():
    n = 4100
    b, v = ic.plummer(n, seed=45)
    with Simulation(n, force_variant=code(8), jsplit=2) as s, Simulation(n, force_variant=code(8), jsplit=2) as t:
        s.init(b, v); s.simulate(7, 1e-3, 1.0)
        state = s.read()
        s.simulate(6)                       # odd count: the live buffers are the other pair now
        s.restore(*state)                   # positions rewritten from outside the step
        s.simulate(9)
        t.init(b, v); t.simulate(16, 1e-3, 1.0)
        for x, y in zip(s.read(), t.read()):
            assert x.tobytes() == y.tobytes()
        # the same through graph replays (an upload between two replays of the same captured graph)
        s.simulate(40); t.simulate(40)
        for wander in (33, 32):              # the restore lands on either buffer parity: with or without a re-capture
            state = s.read()
            s.simulate(wander)
            s.restore(*state)
            s.simulate(48); t.simulate(48)
            for x, y in zip(s.read(), t.read()):
                assert x.tobytes() == y.tobytes(), wander
        # G is folded into the pair copy: a change must rebuild it
        s.simulate(3, 1e-3, 0.25); t.simulate(3, 1e-3, 0.25)
        for x, y in zip(s.read(), t.read()):
            assert x.tobytes() == y.tobytes()
        got = s.read()
    rb, rv, ra = oracle.run_f64(b, v, None, 1e-3, 1.0, 16 + 40 + 48 + 48)
    rb, rv, ra = oracle.run_f64(rb, rv, ra, 1e-3, 0.25, 3)
    assert rel_pos_err(got[0], rb, 1.0) < 2e-5
    # a raw device pointer handed out: the engine assumes the caller wrote through it
    with Simulation(n, force_variant=code(4)) as s:
        s.init(b, v); s.simulate(2, 1e-3, 1.0)
        before = s.read()
        s.device_ptr("bodies")
        s.simulate(2)
        after = s.read()
    with Simulation(n, force_variant=code(4)) as t:
        t.init(b, v); t.simulate(4, 1e-3, 1.0)
        for x, y in zip(after, t.read()):
            assert x.tobytes() == y.tobytes()
    assert before[0].tobytes() != after[0].tobytes()


def test_entry_points_follow_the_live_buffers():
    """Single steps, graph replay, diagnostics and the frame feed between steps, dt = 0."""
    n = 6000
    b, v = ic.plummer(n, seed=46)
    with Simulation(n, force_variant=code(8), jsplit=3) as s, Simulation(n, force_variant=304014, jsplit=8) as t:
        for sim in (s, t):
            sim.init(b, v); sim.set_params(1e-3, 1.0)
        done = 0
        for k in (1, 2, 17, 3, 128):
            s.simulate(k); t.simulate(k)
            done += k
            ks, ps = s.diagnostics()[:2]
            kt, pt = t.diagnostics()[:2]
            assert abs(ks - kt) < 1e-6 * abs(kt) and abs(ps - pt) < 1e-6 * abs(pt)
            s.request_frame()
            fb, fspeed, fstep = s.frame(wait=True)
            rb, rv, ra = s.read()
            assert fstep == done and fb.tobytes() == rb.tobytes()
            assert rel_pos_err(rb, t.read()[0], 1.0) < 5e-6
        frozen = s.read()
        s.simulate(5, 0.0)                  # paused: `if (dt > 0)`, nbody3d.js:474
        for x, y in zip(frozen, s.read()):
            assert x.tobytes() == y.tobytes()
        with pytest.raises(Exception) as e:
            s.set_exchange(lambda *a: 0)
        assert "NB_ERR_STATE" in str(e.value)


def test_sharded_and_f64_handles_never_take_the_pair_kernel():
    n = 8192
    b, v = ic.plummer(n, seed=47)
    with Simulation(n, force_variant=code(8), shard=(0, 4096)) as s:
        assert "jpairs" not in s.variant, s.variant
    with Simulation(n, shard=(0, 4096)) as s:                     # nor by the automatic choice (n = 8,192 is inside its range)
        assert "jpairs" not in s.variant, s.variant
    with Simulation(n, flags=capi.NB_FLAG_NO_SYM) as s:           # a whole f32 system without the symmetric pass: the one-launch ordered-pair steps
        assert "jpairs" in s.variant or "fused_lds" in s.variant, s.variant
    with Simulation(n, precision="f64", force_variant=code(8)) as s:
        assert "jpairs" not in s.variant and s.variant.startswith("f64"), s.variant
    with Simulation(n, force_variant=code(8), flags=capi.NB_FLAG_NO_FUSE) as s:
        assert "jpairs" not in s.variant, s.variant


@pytest.mark.parametrize("n,jsplit", [(8192, 4), (10000, 0)])
def test_jpk_fenced_fallback_is_bit_identical(n, jsplit):
    """NB_FLAG_JPK_FENCED: plain partial stores + an agent-scope release on the ticket instead of write-through stores +
    a relaxed ticket -- the conservative fallback for parts / partition modes where the sc1 sequence might not hold."""
    b, v = ic.plummer(n, seed=76)
    fast = run(b, v, 40, force_variant=601018, jsplit=jsplit)
    safe = run(b, v, 40, force_variant=601018, jsplit=jsplit, flags=capi.NB_FLAG_JPK_FENCED)
    assert "jpairs" in fast[3] and fast[3] == safe[3]
    for x, y in zip(fast[:3], safe[:3]):
        assert x.tobytes() == y.tobytes(), fast[3]


def test_default_shape_at_an_auto_selected_jpk_size_with_poisoned_partials():
    """validate the planner's OWN choice of the j-packed step, not only pinned variants -- N = 9,000 with no shape pin
    and the symmetric pass switched off (since round 5 the default from N ~ 7,000 is the symmetric pass; this is the ordered-pair
    default there, what a handle that cannot pair up symmetrically gets) lands on the j-packed step with a split across workgroups;
    400 steps through graph replay with every consumed partial overwritten by NaN: finite, and bit-identical to the unpoisoned and to
    the fenced run."""
    n = 9000
    b, v = ic.plummer(n, seed=77)
    p = run(b, v, 400, flags=capi.NB_FLAG_POISON | capi.NB_FLAG_NO_SYM)
    q = run(b, v, 400, flags=capi.NB_FLAG_NO_SYM)
    r = run(b, v, 400, flags=capi.NB_FLAG_JPK_FENCED | capi.NB_FLAG_NO_SYM)
    assert "jpairs" in q[3] and "_js1" != q[3][-4:], q[3]
    for a in p[:3]:
        assert np.isfinite(a).all(), p[3]
    for x, y, z in zip(p[:3], q[:3], r[:3]):
        assert x.tobytes() == y.tobytes() == z.tobytes(), q[3]
