"""GPU tests of the step FORMS, through the C ABI: the fused one-launch step (ping-pong positions: nb_step_fused / nb_step_direct,
SURVEY.md §8 f3) against the two-kernel step, bit for bit; the ordered-pair fp64 launch shapes against the fp64 oracle (BASELINE
config 5's arithmetic; the symmetric fp64 pass is in test_sym_gpu.py); the integrate kernel on its own; which form the planner
picks by size; what a whole-system handle reports about its shape.
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


# ---- fused step -------------------------------------------------------------------------------

# (bodies per lane, lanes per body, 256-body tile units per LDS stage)
FUSED_SHAPES = [(2, 1, 1), (2, 16, 1), (2, 64, 1), (2, 64, 4), (2, 32, 4), (4, 4, 1), (4, 32, 1), (4, 16, 4),
                (8, 1, 1), (8, 2, 1), (8, 8, 1), (8, 64, 1), (8, 64, 4), (2, 64, 8), (2, 32, 8), (4, 64, 8), (8, 32, 8)]


@pytest.mark.parametrize("ipl,ls,tl", FUSED_SHAPES)
@pytest.mark.parametrize("n", [1000, 4096, 7001])
def test_fused_step_is_bit_identical_to_the_two_kernel_step(ipl, ls, tl, n):
    """nb_step_fused<NG,LS,TL> == nb_force_pk<NG,LS,TL> (one j-split) + nb_integrate: same loop,
    same in-wave reduction, same leapfrog -- 19 steps so that graph replay (16) and both
    ping-pong parities are exercised."""
    b, v = (ic.plummer(n, seed=31) if n % 256 == 0 else ic.uniform_cube(n, seed=31))
    code = ipl * 1000 + ls * 10 + tl
    fb, fv, fa, fname = run(b, v, 19, force_variant=400000 + code)
    tb, tv, ta, tname = run(b, v, 19, force_variant=200000 + code, jsplit=1)
    assert "fused" in fname and "fused" not in tname and tname.endswith("_js1"), (fname, tname)
    assert fb.tobytes() == tb.tobytes() and fv.tobytes() == tv.tobytes() and fa.tobytes() == ta.tobytes(), (fname, tname)
    rb, rv, ra = oracle.run_f64(b, v, None, 1e-3, 1.0, 19)
    assert rel_pos_err(fb, rb, 1.0) < 2e-5, fname


@pytest.mark.parametrize("n,x", [(1, 1), (63, 1), (771, 1), (1000, 1), (1024, 1), (1000, 2), (1025, 2), (1500, 2), (2048, 2)])
def test_registers_only_fused_step_is_bit_identical_to_the_tiled_fused_step(n, x):
    """nb_step_direct<16|32> (N <= 1,024 | 2,048: each lane's j-bodies loaded straight into registers,
    no LDS) == nb_step_fused<1,64,4>: same lane -> j mapping, same order, same reduction."""
    b, v = ic.uniform_cube(n, seed=35)
    db, dv, da, dname = run(b, v, 19, force_variant=502640 + x)
    fb, fv, fa, fname = run(b, v, 19, force_variant=402644)
    assert "fused_regs%d" % (1024 * x) in dname and "fused_lds1024" in fname, (dname, fname)
    assert db.tobytes() == fb.tobytes() and dv.tobytes() == fv.tobytes() and da.tobytes() == fa.tobytes(), (dname, fname)
    rb, _, _ = oracle.run_f32(b, v, None, 1e-3, 1.0, 19)
    assert rel_pos_err(db, rb, 1.0) < 2e-5, dname


def test_registers_only_step_is_refused_above_its_size():
    b, v = ic.plummer(4096, seed=36)
    _, _, _, name = run(b, v, 1, force_variant=502641)       # 4,096 bodies do not fit 16 rows per lane
    assert "fused_lds1024" in name, name


def test_default_small_system_takes_the_fused_path_and_flag_disables_it():
    b, v = ic.plummer(4096, seed=32)
    fb, fv, fa, fname = run(b, v, 5)
    nb_, nv, na, nname = run(b, v, 5, flags=capi.NB_FLAG_NO_FUSE)
    assert "fused" in fname and "fused" not in nname, (fname, nname)
    rb, _, ra = oracle.run_f64(b, v, None, 1e-3, 1.0, 5)
    for x, a, name in ((fb, fa, fname), (nb_, na, nname)):
        assert rel_pos_err(x, rb, 1.0) < 2e-5, name
        assert np.abs(a[:, :3] - ra[:, :3]).max() < TOL_ACC * np.abs(ra[:, :3]).max(), name


def test_fused_handle_interleaves_steps_reads_restores_and_diagnostics():
    """Positions live in one of two buffers; every entry point must follow the live one."""
    n = 2048
    b, v = ic.plummer(n, seed=33)
    with Simulation(n, force_variant=402644) as f, Simulation(n, force_variant=202644, jsplit=1) as t:
        for s in (f, t):
            s.init(b, v)
            s.set_params(1e-3, 1.0)
        for k in (1, 2, 17, 3):                  # odd and even step counts, one graph replay
            for s in (f, t):
                s.simulate(k)
            rf, rt = f.read(), t.read()
            for x, y in zip(rf, rt):
                assert x.tobytes() == y.tobytes(), k
            kf, pf, mf = f.diagnostics()
            kt, pt, mt = t.diagnostics()
            assert kf == kt and pf == pt
        state = f.read()
        f.simulate(5); t.simulate(5)
        want = t.read()
        f.restore(*state)                        # mid-run restore on an odd ping-pong parity
        f.simulate(5)
        for x, y in zip(f.read(), want):
            assert x.tobytes() == y.tobytes()
        with pytest.raises(Exception) as e:
            f.set_exchange(lambda *a: 0)
        assert "NB_ERR_STATE" in str(e.value)


# ---- fp64 (BASELINE.json config 5) ------------------------------------------------------------

F64_SHAPES = [(1, 1), (1, 3), (2, 1), (2, 2), (4, 1), (4, 5), (14, 1), (14, 2), (116, 1), (116, 2), (164, 1), (164, 3), (0, 0)]


@pytest.mark.parametrize("variant,jsplit", F64_SHAPES)
def test_f64_single_step_matches_fp64_oracle(variant, jsplit):
    n = 2048
    b, v = ic.plummer(n, seed=41)
    b, v = b.astype(np.float64), v.astype(np.float64)
    bb, vv, aa, name = run(b, v, 1, precision="f64", force_variant=variant, jsplit=jsplit)
    assert name.startswith("f64"), name
    ref = oracle.accel_f64(b, 1.0)
    assert np.abs(aa[:, :3] - ref[:, :3]).max() < TOL_F64 * np.abs(ref[:, :3]).max(), name
    rb, rv, ra = oracle.run_f64(b, v, None, 1e-3, 1.0, 1)
    assert rel_pos_err(bb, rb, 1.0) < TOL_F64, name


@pytest.mark.parametrize("variant,jsplit", F64_SHAPES)
@pytest.mark.parametrize("name,steps", [("plummer1024", 100), ("galaxy_ref", 30), ("disk771", 50)])
def test_f64_golden_trajectories(manifest, name, steps, variant, jsplit):
    m = manifest[name]
    b0 = load_golden32(name + "_bodies0").astype(np.float64)
    v0 = load_golden32(name + "_vel0").astype(np.float64)
    bb, vv, aa, vname = run(b0, v0, steps, dt=m["dt"], G=m["G"], precision="f64", force_variant=variant, jsplit=jsplit)
    ref = load_golden64("%s_s%d_bodies" % (name, steps))
    assert rel_pos_err(bb, ref, m["r_scale"]) < TOL_F64, (vname, name)


def test_f64_full_size_properties():
    """Config 5 at N=262,144 with the launch shape the engine picks for it: sampled rows against
    the fp64 oracle at 1e-12, Newton's third law, and agreement with the f32 engine."""
    n = 262144
    b, v = ic.plummer(n, seed=1)
    b64, v64 = b.astype(np.float64), v.astype(np.float64)
    bb, vv, aa, name = run(b64, v64, 1, precision="f64")
    assert name.startswith("f64"), name
    rows = np.sort(np.random.default_rng(1).choice(n, 24, replace=False))
    for i in rows:
        ref = oracle.accel_f64(b64, 1.0, i0=int(i), i1=int(i) + 1)[0, :3]
        assert np.abs(aa[i, :3] - ref).max() < TOL_F64 * max(np.abs(ref).max(), 1e-3), (name, i)
    f = (b64[:, 3:4] * aa[:, :3]).sum(0)
    assert np.all(np.abs(f) < 1e-11 * np.abs(b64[:, 3:4] * aa[:, :3]).sum(0))
    fb, fv, fa, fname = run(b, v, 1)
    assert np.abs(fa[:, :3] - aa[:, :3]).max() < TOL_ACC * np.abs(aa[:, :3]).max(), fname


def test_integrate_pass_measures_the_integrator_alone():
    n = 1 << 20
    b, v = ic.uniform_cube(n, seed=63)
    with Simulation(n, force_variant=208011, jsplit=1, flags=capi.NB_FLAG_NO_FUSE) as sim:
        sim.init(b, v)
        sim.set_params(1e-3, 1.0)
        ms = sim.integrate_pass(20)
        assert 0 < ms < 5.0                      # 100 MB of traffic: tens of microseconds
    with Simulation(4096) as fused:
        fused.init(*ic.plummer(4096, seed=1))
        with pytest.raises(Exception) as e:
            fused.integrate_pass(1)
        assert "NB_ERR_STATE" in str(e.value)


# ---- the automatic launch shape ----------------------------------------------------------------

@pytest.mark.parametrize("n,family,classic", [(512, "fused_regs", None), (1024, "fused_regs", None), (2002, "fused_lds", None), (4096, "fused_lds", None),
                                              (6000, "fused_lds", None), (7000, "symw_ipl8", "fused_jpairs"), (8192, "symw_ipl8", "fused_jpairs"), (9000, "symw_ipl8", "fused_jpairs"), (12000, "symw", "fused_jpairs"), (14000, "symw", None),
                                              (20000, "symw", "sgpr"), (32768, "symw", "sgpr"),
                                              (40002, "symw_ipl16_j1_w2048", "sgpr"), (65536, "symw_ipl16", "sgpr"), (131072, "symw_ipl16_j1_w2048", "sgpr"),
                                              (262144, "symw_ipl16_j1_w2048", "sgpr_ipl8_ws4"), (500010, "symw_ipl16", "sgpr"), (1048576, "symw_ipl16", "sgpr")])
def test_default_launch_shape_family_by_size(n, family, classic):
    """the planner's (csrc/nb_plan.cpp) pick per system size, as measured best (profiles/r02/size_scan_final_4k_65k.txt below N ~ 14,000,
    profiles/r04/sym_units_scan_workgroup_reduce.txt above): from N ~ 10,000 the symmetric pass; with NB_FLAG_NO_SYM the
    ordered-pair families of round 2.  (A refit of one model constant once moved N = 12,000 .. 32,768 onto a shape 1-6 %
    slower without any test noticing.)"""
    with Simulation(n) as s:
        assert family in s.variant, (n, s.variant)
    if classic:
        with Simulation(n, flags=capi.NB_FLAG_NO_SYM) as s:
            assert classic in s.variant and "sym" not in s.variant, (n, s.variant)
    with Simulation(n, shard=(0, (n // 2 + 255) // 256 * 256 if n > 512 else n)) as s:      # a rank's shard never fuses, nor pairs up symmetrically
        assert "fused" not in s.variant and "sym" not in s.variant, (n, s.variant)


def test_shape_info_of_a_whole_system_handle():
    with Simulation(262144) as s:
        info = s.shape_info()
        assert info["jsplit"] >= 1 and info["own_splits"] == 0 and "symw" in s.variant
    with Simulation(262144, flags=capi.NB_FLAG_NO_SYM) as s:
        info = s.shape_info()
        assert info["jsplit"] >= 1 and info["own_splits"] == 0 and "_js%d" % info["jsplit"] in s.variant
        assert info["j_per_split"] * info["jsplit"] >= 262144
    with Simulation(4096) as s:
        assert s.shape_info()["own_splits"] == 0


@pytest.mark.parametrize("n,prec", [(1000, "f32"), (5000, "f32"), (12000, "f32"), (16384, "f32"), (40002, "f32"), (100000, "f32"), (40002, "f64")])
def test_plan_query_is_what_create_builds(n, prec):
    """nb_plan_query (the planner on the host alone, tests/test_planner_cpu.py) and nb_create agree on this device."""
    q = capi.plan_query(n, precision=prec, n_cu=0, clock_hz=0)
    with Simulation(n, precision=prec) as sim:
        assert sim.variant == q["variant"]
        assert sim.shape_info() == {k: q[k] for k in ("jsplit", "j_per_split", "own_split0", "own_splits")}
