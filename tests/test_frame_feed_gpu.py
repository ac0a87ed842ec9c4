"""GPU tests of the viewer frame feed (nbody3d.js:408-415,482-487 reads bodyBuffer / velBuffer in place every frame):
frames equal what read() returns, the feed runs ahead of the copies without blocking the step stream, and a failed slot allocation
(fault injection in the -DNB_TUNING build) is an error code, not a fault.
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


# ---- frame feed, integrate pass ----------------------------------------------------------------

@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_frame_feed_values_equal_read(precision):
    """nb_frame_request / nb_frame_acquire: the snapshot taken after step k equals read() at
    step k (f32 bodies + length(vel.xyz), nbody3d.js:380), however many steps follow it."""
    n = 3000
    b, v = ic.plummer(n, seed=61)
    dt = np.float64 if precision == "f64" else np.float32
    with Simulation(n, precision=precision) as sim:
        sim.init(b.astype(dt), v.astype(dt))
        sim.set_params(1e-3, 1.0)
        with pytest.raises(Exception) as e:
            sim.frame(wait=False)                          # nothing requested yet
        assert "NB_ERR_STATE" in str(e.value)
        done = 0
        for k in (3, 1, 4):
            sim.simulate(k)
            done += k
            sim.request_frame()
            sim.simulate(2)                                # later steps must not disturb the snapshot
            done += 2
            fb, fs, step = sim.frame(wait=True)
            assert step == done - 2
            fb, fs = fb.copy(), fs.copy()
    with Simulation(n, precision=precision) as ref:
        ref.init(b.astype(dt), v.astype(dt))
        ref.simulate(done - 2, 1e-3, 1.0)
        rb, rv, _ = ref.read()
    assert fb.dtype == np.float32 and fs.dtype == np.float32
    assert np.array_equal(fb, rb.astype(np.float32))
    rv32 = rv[:, :3].astype(np.float32)
    want = np.sqrt(rv32[:, 0] * rv32[:, 0] + rv32[:, 1] * rv32[:, 1] + rv32[:, 2] * rv32[:, 2])
    assert np.allclose(fs, want, rtol=2e-6, atol=0)


# ---- fault injection, fallbacks ----------------------------------------------------------------------

FRAME_FAIL_SCRIPT = r"""
import sys
sys.path.insert(0, %(pkg)r)
import numpy as np
from nbody3d_amd import Simulation, capi, ic
assert capi.library_path().endswith("_tuning.so")
n = 3000
b, v = ic.plummer(n, seed=5)
with Simulation(n) as sim:
    sim.init(b, v)
    sim.simulate(3, 1e-3, 1.0)
    for attempt in range(2):                      # a failed set-up must leave nothing half-built behind
        try:
            sim.request_frame()
            print("NO-ERROR")
        except capi.NBodyError as e:
            print("ERR", e.code, str(e)[:90])
        try:
            sim.frame(wait=False)
            print("NO-ERROR")
        except capi.NBodyError as e:
            print("ACQ", e.code)
    sim.simulate(2)                               # the handle itself is still fine
    got = sim.read()[0]
with Simulation(n) as ref:
    ref.init(b, v)
    ref.simulate(5, 1e-3, 1.0)
    print("SAME", got.tobytes() == ref.read()[0].tobytes())
"""


@pytest.mark.parametrize("slot", [0, 2])
def test_frame_slot_allocation_failure_is_an_error_code_not_a_fault(slot):
    """nb_frame_request published its stream before the four slots existed, so a failed allocation left
    null buffers behind and the NEXT request packed into them (a GPU memory fault).  The set-up is all-or-nothing now; the
    -DNB_TUNING build fails the k-th slot on request (NB_TEST_FAIL_FRAME_SLOT)."""
    lib = os.path.join(ROOT, "nbody3d-webgpu_amd", "csrc", "libnbody3d_hip_tuning.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.dirname(lib), "-s", "tuning"])
    env = dict(os.environ, NB_ENGINE_LIB=lib, NB_TEST_FAIL_FRAME_SLOT=str(slot))
    p = subprocess.run([sys.executable, "-c", FRAME_FAIL_SCRIPT % {"pkg": os.path.join(ROOT, "nbody3d-webgpu_amd")}],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
    lines = p.stdout.split("\n")
    assert sum(l.startswith("ERR 5") for l in lines) == 2 and "NO-ERROR" not in p.stdout, p.stdout      # NB_ERR_NOMEM, twice
    assert sum(l.startswith("ACQ 4") for l in lines) == 2, p.stdout                                      # nothing requested: NB_ERR_STATE
    assert "SAME True" in p.stdout, p.stdout


def test_frame_feed_runs_ahead_without_blocking_and_snapshots_stay_exact():
    """The functional half of the frame feed at the reference's default size (one snapshot per frame, as render() draws):
    requests never need an acquire in between (a ring of four slots; the host is held back, never the step stream), every
    acquired snapshot is a finished frame of an earlier-or-equal step, and the last one equals read().  (The wall-clock
    comparison with and without snapshots lives in tools/feed_driver.py: a timing gate does not belong in a -x suite.)"""
    n = 40002
    b, v = ic.uniform_cube(n, seed=62)
    with Simulation(n) as sim:
        sim.init(b, v)
        sim.set_params(1e-4, 1e-4)
        last = -1
        for k in range(60):
            sim.step()
            sim.request_frame()
            if k % 7 == 3:
                got = sim.frame(wait=False)
                if got is not None:
                    assert last <= got[2] <= k + 1
                    last = got[2]
        fb, fs, step = sim.frame(wait=True)
        fb = fb.copy()
        assert step == 60
        assert fb.tobytes() == sim.read(vel=False, accel=False)[0].tobytes()
