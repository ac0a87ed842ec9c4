"""GPU tests at G != 1: every packed f32 kernel multiplies (G*m_j)*inv per pair -- the reference's product (nbody3d.js:236) -- through
the (x, y, z, G*m) j-stream copy.  The step forms agree with each other as tightly as at G = 1, and the copy follows uploads,
G changes, graph replay, exchanges (virtual shards, the overlapped exchange) and raw device pointers.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden32, load_golden64, rel_pos_err
from oracle import oracle
from nbody3d_amd import MultiSimulation, Simulation, capi, ic

pytestmark = pytest.mark.gpu

TOL_ACC, TOL_TIGHT = 2e-5, 2e-5
TOL_F64 = 1e-12


def run(b, v, steps, dt=1e-3, G=1.0, **kw):
    with Simulation(b.shape[0], **kw) as sim:
        sim.init(b, v)
        sim.simulate(steps, dt, G)
        return sim.read() + (sim.variant,)


# ---- G != 1: the (x, y, z, G*m) j-stream -----------------------------------------------------------

@pytest.mark.parametrize("G", [1e-4, 0.37, 3.0])
@pytest.mark.parametrize("n,fused,two", [(1000, 502641, 202644), (4096, 402644, 202644), (7001, 404324, 204324), (4096, 408161, 208161)])
def test_fused_and_two_kernel_steps_stay_bit_identical_at_any_G(G, n, fused, two):
    """Round 2 applied G to the finished sums in the packed kernels: identical bits only at G = 1.  Now the fused,
    registers-only and two-kernel forms of one loop shape are bit-identical at every G, through graph replay (19 steps)."""
    b, v = (ic.plummer(n, seed=71) if n % 256 == 0 else ic.uniform_cube(n, seed=71))
    fb, fv, fa, fname = run(b, v, 19, G=G, force_variant=fused)
    tb, tv, ta, tname = run(b, v, 19, G=G, force_variant=two, jsplit=1)
    assert "fused" in fname and "fused" not in tname, (fname, tname)
    assert fb.tobytes() == tb.tobytes() and fv.tobytes() == tv.tobytes() and fa.tobytes() == ta.tobytes(), (fname, tname, G)
    rb, rv, ra = oracle.run_f64(b, v, None, 1e-3, G, 19)
    assert rel_pos_err(fb, rb, 1.0) < TOL_TIGHT, fname


@pytest.mark.parametrize("name,steps", [("galaxy_ref", 30), ("disk771", 50)])
def test_g_not_one_fixtures_agree_across_kernel_families_and_with_the_fp32_oracle(manifest, name, steps):
    """The two committed G = 1e-4 fixtures: LDS-tile, SGPR, fused, registers-only and j-packed steps against the fp32
    oracle's vector and against each other.  Before the j-stream carried G*m the packed kernels sat ~2x further from the
    j-packed step (which always folded G into the masses) than they do at G = 1."""
    m = manifest[name]
    b0, v0 = load_golden32(name + "_bodies0"), load_golden32(name + "_vel0")
    ref32 = load_golden32("%s_s%d_bodies" % (name, steps))
    outs = {}
    for label, fv, js in (("lds", 22, 1), ("sgpr", 304014, 2), ("fused", 402644, 0), ("regs", 502641, 0), ("jpk", 601014, 2), ("scalar", 1, 1)):
        bb, vv, aa, vname = run(b0, v0, steps, dt=m["dt"], G=m["G"], force_variant=fv, jsplit=js)
        outs[label] = bb
        assert rel_pos_err(bb, ref32, m["r_scale"]) < 2e-6, (vname, rel_pos_err(bb, ref32, m["r_scale"]))
    for label, bb in outs.items():
        assert rel_pos_err(bb, outs["jpk"], m["r_scale"]) < 2e-6, label
    assert outs["fused"].tobytes() == outs["regs"].tobytes()


@pytest.mark.parametrize("n,variant", [(1024, 0), (4096, 402644), (20000, 0), (20000, 28)])
def test_j_stream_copy_follows_G_changes_uploads_graphs_and_raw_pointers(n, variant):
    """The copy is rebuilt when G changes (1 -> 0.5 -> 1 -> 0.25), after a restore on either ping-pong parity, and after a
    raw device pointer was handed out; graph replays (>= 16 steps) and single steps must give the same bits."""
    b, v = ic.plummer(n, seed=72)
    with Simulation(n, force_variant=variant) as a, Simulation(n, force_variant=variant) as c:
        a.init(b, v)
        c.init(b, v)
        for G, k in ((1.0, 17), (0.5, 35), (1.0, 3), (0.25, 20)):
            a.simulate(k, 1e-3, G)
            for _ in range(k):
                c.step(1e-3, G)
        for x, y in zip(a.read(), c.read()):
            assert x.tobytes() == y.tobytes(), a.variant
        state = a.read()
        a.simulate(5)                     # odd: a fused handle now lives in the other buffer pair
        a.restore(*state)
        a.simulate(21)
        c.simulate(21)
        for x, y in zip(a.read(), c.read()):
            assert x.tobytes() == y.tobytes(), a.variant
        a.device_ptr("bodies")            # the engine must assume the caller wrote through it
        a.simulate(2)
        c.simulate(2)
        got, want = a.read(), c.read()
        name = a.variant
    for x, y in zip(got, want):
        assert x.tobytes() == y.tobytes(), name
    # (the trajectory itself is covered by the parity suites; here only that both ways of driving the handle agree)


@pytest.mark.parametrize("g,variant", [(2, 22), (4, 308014), (8, 28)])
def test_virtual_shards_at_G_not_one_equal_the_single_handle(g, variant):
    """nb_multi with g shards on one GPU at G = 0.01: after every peer-copy all-gather each shard rebuilds the other shards'
    rows of its (x, y, z, G*m) copy; bit-identical to one unsharded handle of the same launch shape."""
    n, steps = 4096, 7
    b, v = ic.plummer(n, seed=73)
    kw = dict(force_variant=variant, jsplit=4)
    with Simulation(n, flags=capi.NB_FLAG_NO_FUSE, **kw) as one:
        one.init(b, v)
        one.simulate(steps, 1e-3, 0.01)
        ref = one.read()
    with MultiSimulation(n, g, **kw) as ms:
        ms.init(b, v)
        ms.simulate(3, 1e-3, 0.01)
        for _ in range(steps - 3):
            ms.step()
        got = ms.read()
        name = ms.variant
    for x, y in zip(got, ref):
        assert x.tobytes() == y.tobytes(), name


def test_overlapped_exchange_at_G_not_one_waits_and_stays_exact():
    """G != 1 on a handle whose rows are exchanged: the step waits for the gather, rebuilds the j-stream copy and runs the
    whole force pass (documented: the overlapped form only overlaps at G = 1).  Same bits as the unsharded handle."""
    import torch
    n, g, steps = 4096, 2, 5
    per = n // g
    b, v = ic.plummer(n, seed=74)
    kw = dict(force_variant=22, jsplit=8)
    with Simulation(n, **kw) as one:
        one.init(b, v)
        one.simulate(steps, 1e-3, 0.3)
        ref = one.read()
    stream = torch.cuda.current_stream().cuda_stream
    bufs = [torch.empty((n, 4), device="cuda", dtype=torch.float32) for _ in range(g)]
    sims = [Simulation(n, shard=(r * per, per), stream=stream, ext_bodies=bufs[r].data_ptr(), **kw) for r in range(g)]
    snap = {}
    try:
        for r, s in enumerate(sims):
            s.init(b, v)
            s.set_params(1e-3, 0.3)

            def wait(st, r=r):
                for q in range(g):
                    if q != r:
                        bufs[r][q * per:(q + 1) * per].copy_(snap[q])
                return 0

            s.set_exchange_overlapped(lambda *a: 0, wait)
        for _ in range(steps):
            snap = {q: bufs[q][q * per:(q + 1) * per].clone() for q in range(g)}
            for s in sims:
                s.step()
        for s in sims:
            s.sync()
        bodies = np.concatenate([bufs[r][r * per:(r + 1) * per].cpu().numpy() for r in range(g)])
    finally:
        for s in sims:
            s.close()
    assert bodies.tobytes() == ref[0].tobytes()
