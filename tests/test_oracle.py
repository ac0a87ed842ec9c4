"""CPU tests of the oracle (oracle/nb_oracle.c) -- the pins SURVEY.md §8(c)
asks for, since the reference holds no tests or golden vectors of its own."""
import os

import numpy as np
import pytest

from conftest import load_golden32, load_golden64, rel_pos_err
from oracle import oracle
from nbody3d_amd import ic

EPS2 = oracle.EPS2


def two_body(r, m=3.0):
    b = np.zeros((2, 4), np.float32)
    b[1, :3] = r
    b[0, 3] = 1.0
    b[1, 3] = m
    return b


@pytest.mark.parametrize("r", [(1, 0, 0), (0.3, -0.4, 1.2), (1e-3, 0, 0), (100.0, 50.0, -20.0)])
def test_two_body_known_answer(r):
    """a = G m r / (r^2 + eps^2)^(3/2)  (nbody3d.js:232-237)."""
    G = 0.5
    b = two_body(r)
    a32 = oracle.accel_f32(b, G)
    a64 = oracle.accel_f64(b, G)
    rv = b[1, :3].astype(np.float64)   # the f32-rounded separation both oracles see
    d2 = (rv ** 2).sum() + EPS2
    exp0 = G * 3.0 * rv / d2 ** 1.5
    exp1 = -G * 1.0 * rv / d2 ** 1.5
    assert np.allclose(a64[0, :3], exp0, rtol=1e-12, atol=0)
    assert np.allclose(a64[1, :3], exp1, rtol=1e-12, atol=0)
    assert np.allclose(a32[0, :3], exp0, rtol=5e-6, atol=0)
    assert a32[0, 3] == 0 and a64[1, 3] == 0


def test_coincident_bodies_give_exact_zero():
    """r = 0 -> 0 * finite = exactly 0 thanks to eps2."""
    b = two_body((0, 0, 0))
    assert np.all(oracle.accel_f32(b, 1.0) == 0)
    assert np.all(oracle.accel_f64(b, 1.0) == 0)


def test_integrator_constant_field_algebra():
    """SURVEY.md §8(c) pin 2.  One heavy far-away source makes a (nearly)
    constant field; after k calls  x_k = x0 + k dt v0 + k(k+1)/2 dt^2 a  and
    vel_k = v0 + (k - 1/2) dt a  (first call has a_old = 0, nbody3d.js:195-199)."""
    G, dt, k = 1.0, 1e-3, 7
    b = np.zeros((2, 4), np.float64)
    b[1] = (1e6, 0, 0, 1e12)     # a on body 0 = G m / r^2 = 1.0 along +x
    b[0, 3] = 1e-30
    v = np.zeros((2, 4), np.float64)
    v[0, :3] = (0.25, -0.5, 0.125)
    a0 = oracle.accel_f64(b, G)[0, :3]
    bk, vk, ak = oracle.run_f64(b, v, None, dt, G, k)
    x_exp = b[0, :3] + k * dt * v[0, :3] + 0.5 * k * (k + 1) * dt * dt * a0
    v_exp = v[0, :3] + (k - 0.5) * dt * a0
    assert np.allclose(bk[0, :3], x_exp, rtol=0, atol=1e-9)   # field varies by ~5e-9 over the path
    assert np.allclose(vk[0, :3], v_exp, rtol=0, atol=1e-9)
    assert np.allclose(ak[0, :3], a0, rtol=1e-9)
    assert bk[0, 3] == b[0, 3] and vk[0, 3] == 0     # mass lane untouched (vel.w = 0)


def test_newton_third_law():
    b, _ = ic.plummer(512, seed=5)
    a = oracle.accel_f64(b, 1.0)
    f = (b[:, 3:4].astype(np.float64) * a[:, :3]).sum(0)
    scale = np.abs(b[:, 3:4] * a[:, :3]).sum(0)
    assert np.all(np.abs(f) < 1e-12 * scale)


def test_dt_zero_is_a_noop():
    """`if (dt > 0)` gate, nbody3d.js:474."""
    b, v = ic.plummer(256, seed=6)
    a = np.random.default_rng(0).random((256, 4)).astype(np.float32)
    b2, v2, a2 = oracle.run_f32(b, v, a, 0.0, 1.0, 5)
    assert b2.tobytes() == b.tobytes() and v2.tobytes() == v.tobytes() and a2.tobytes() == a.tobytes()


@pytest.mark.parametrize("name,steps", [("plummer1024", [1, 10, 100]), ("cube1000", [1, 20]), ("disk771", [1, 50]),
                                        ("galaxy_ref", [1, 30])])
def test_golden_vectors_bit_exact(manifest, name, steps):
    """The committed vectors are reproduced bit for bit by the oracle."""
    m = manifest[name]
    b, v, a = load_golden32(name + "_bodies0"), load_golden32(name + "_vel0"), None
    done = 0
    for k in steps:
        b, v, a = oracle.run_f32(b, v, a, m["dt"], m["G"], k - done)
        done = k
        assert b.tobytes() == load_golden32("%s_s%d_bodies" % (name, k)).tobytes()
        assert v.tobytes() == load_golden32("%s_s%d_vel" % (name, k)).tobytes()
        assert a.tobytes() == load_golden32("%s_s%d_accel" % (name, k)).tobytes()


def test_f32_oracle_within_stated_tolerance_of_f64(manifest):
    """Sets the 'stated fp32 tolerance': BASELINE.json asks 1e-4 after 100 steps."""
    m = manifest["plummer1024"]
    b32 = load_golden32("plummer1024_s100_bodies")
    b64 = load_golden64("plummer1024_s100_bodies")
    err = rel_pos_err(b32, b64, m["r_scale"])
    assert err == pytest.approx(m["f32_vs_f64_max_rel_pos_err"], rel=1e-6)
    assert err < 1e-5
    assert m["energy_drift"]["f32"] < 1e-4


def test_permutation_invariance_within_tolerance():
    """SURVEY.md §8(c) pin 7: j-order only changes rounding."""
    b, _ = ic.plummer(700, seed=7)
    p = np.random.default_rng(1).permutation(700)
    a = oracle.accel_f32(b, 1.0)
    ap = oracle.accel_f32(b[p], 1.0)
    scale = np.abs(a[:, :3]).max()
    assert np.abs(ap[:, :3] - a[p, :3]).max() < 2e-5 * scale


def test_mt_baseline_matches_restatement():
    """The timed CPU baseline computes the same sums (vector-lane order differs)."""
    b, _ = ic.plummer(1536, seed=8)
    a = oracle.accel_f32(b, 1.0, i0=100, i1=400)
    amt, used = oracle.accel_f32_mt(b, 1.0, i0=100, i1=400)
    assert used >= 1
    a64 = oracle.accel_f64(b, 1.0, i0=100, i1=400)
    scale = np.abs(a64[:, :3]).max()
    assert np.abs(amt[:, :3] - a64[:, :3]).max() < 1e-5 * scale
    assert np.abs(a[:, :3] - a64[:, :3]).max() < 1e-5 * scale


def test_shard_ranges_compose():
    """accel over [i0,i1) slices equals the full call (multi-GPU partitioning)."""
    b, _ = ic.uniform_cube(600, seed=9)
    full = oracle.accel_f32(b, 1.0)
    parts = np.concatenate([oracle.accel_f32(b, 1.0, i0=s, i1=e) for s, e in ((0, 256), (256, 512), (512, 600))])
    assert parts.tobytes() == full.tobytes()


def test_plummer_generator_sanity():
    b, v = ic.plummer(4096, seed=1)
    assert b.dtype == np.float32 and b.shape == (4096, 4) and np.all(v[:, 3] == 0)
    assert abs(b[:, 3].sum() - 1.0) < 1e-5
    ke, pe, mom = oracle.energy(b, v, 1.0)
    assert abs(2 * ke / -pe - 1.0) < 0.1          # virial equilibrium
    assert abs(ke + pe + 0.25) < 0.03             # N-body units: E = -1/4
    assert np.all(np.abs(mom) < 1e-6)


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    """The checker itself under -fsanitize=address,undefined (CPU only): golden fixture bit for bit, ragged ranges, the threaded
    baseline kernel, odd sizes (tests/c/oracle_sanitize.c)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    exe = str(tmp_path / "oracle_sanitize")
    cc = subprocess.run([gcc, "-std=c99", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off",
                         "-fno-fast-math", "-fopenmp", os.path.join(root, "tests", "c", "oracle_sanitize.c"),
                         os.path.join(root, "oracle", "nb_oracle.c"), "-o", exe, "-lm"], capture_output=True, text=True, timeout=300)
    if cc.returncode != 0 and "sanitize" in cc.stderr:
        pytest.skip("this gcc has no sanitizer runtime")
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe, os.path.join(root, "tests", "golden")], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "ok oracle under ASan + UBSan" in run.stdout, run.stdout[-1000:] + run.stderr[-3000:]
