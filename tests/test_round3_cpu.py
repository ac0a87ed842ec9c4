"""No-GPU checks added in round 3: the reference-default initial conditions at full size, and what the release
build of the engine library must NOT contain."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, PKG
from nbody3d_amd import ic

CSRC = os.path.join(PKG, "csrc")


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_reference_default_system_matches_the_reference_generators_digest():
    """js/ic.js::galaxies at the reference's default UI state (index.html:68-74: 2 galaxies x 20,000 bodies -> N = 40,002,
    G = 1e-4) against the SHA-256 recorded when tests/golden/make_galaxy_fixture.js ran the reference's own generator
    text (nbody3d.js:51-133) on the same seeded stream: the full-size state every `-m gpu` galaxy test and bench.py's
    `also` entry start from is the reference generator's, bit for bit."""
    b, v, gp = ic.reference_galaxies(os.path.join(GOLDEN, "galaxy40002_params.json"))      # raises on a digest mismatch
    assert b.shape == (40002, 4) and v.shape == (40002, 4) and b.dtype == np.float32
    assert gp["n"] == 40002 and gp["minBodies"] == gp["maxBodies"] == 20000 and gp["G"] == 1e-4
    assert b[0, 3] == 1e7 and b[20001, 3] == 1e7                                           # nbody3d.js:62, one per galaxy
    assert np.all((b[1:20001, 3] >= 10) & (b[1:20001, 3] < 50)) and not v[:, 3].any()      # :63-64, :68,123
    assert np.allclose(b[:2].ravel(), gp["first_rows"]["bodies"]) and np.allclose(v[-2:].ravel(), gp["last_rows"]["vel"])
    sb, sv, sp = ic.reference_galaxies(os.path.join(GOLDEN, "galaxy_ref_params.json"))     # the committed small fixture, same route
    assert sb.tobytes() == open(os.path.join(GOLDEN, "galaxy_ref_bodies0.f32"), "rb").read()
    assert sv.tobytes() == open(os.path.join(GOLDEN, "galaxy_ref_vel0.f32"), "rb").read()


def test_release_library_reads_no_model_knobs_from_the_environment():
    """choose_shape called getenv("NB_MODEL_*") on every nb_create.  The constants are compiled in now;
    only the -DNB_TUNING calibration build (make tuning; tools/fit_model.py, fault injection) knows those names."""
    rel = open(os.path.join(CSRC, "libnbody3d_hip.so"), "rb").read()
    assert b"NB_MODEL_" not in rel and b"NB_TEST_FAIL" not in rel
    tun = os.path.join(CSRC, "libnbody3d_hip_tuning.so")
    if not os.path.exists(tun):
        subprocess.check_call(["make", "-C", CSRC, "-s", "tuning"])
    blob = open(tun, "rb").read()
    assert b"NB_MODEL_BOUNDARY" in blob and b"NB_TEST_FAIL_FRAME_SLOT" in blob
