"""No-GPU check of the reference-default initial conditions at full size: js/ic.js::galaxies (the bit-exact port of
generateGalaxy, nbody3d.js:51-133) reproduces the SHA-256 of the reference generator's own output for the default system
(N = 40,002: tests/golden/galaxy40002_params.json, made by tests/golden/make_galaxy_fixture.js)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, PKG
from nbody3d_amd import ic


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
