import json
import os
import sys

import numpy as np
import pytest

# torch bundles its own libamdhip64.so (same soname as /opt/rocm's).  Whichever copy
# is loaded first serves the whole process, and initialising torch's device layer
# after the engine has already brought up the other copy fails ("No HIP GPUs are
# available").  Tests that use torch tensors as the engine's bodies buffer need one
# consistent runtime, so load torch's first -- the same order bench.py uses.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PKG = os.path.join(ROOT, "nbody3d-webgpu_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_sessionstart(session):
    """Build whatever native artefact is missing (engine .so, N-API addon, oracle).
    The driver normally runs __graft_entry__.build() first and ships the built files;
    this only covers a fresh checkout.  Building is not a fallback: if hipcc is
    missing the engine tests fail loudly."""
    import subprocess
    need = [(os.path.join(PKG, "csrc", "libnbody3d_hip.so"), os.path.join(PKG, "csrc")),
            (os.path.join(PKG, "js", "addon", "nb_napi.node"), os.path.join(PKG, "js")),
            (os.path.join(ROOT, "oracle", "libnb_oracle.so"), os.path.join(ROOT, "oracle"))]
    for artefact, d in need:
        if not os.path.exists(artefact):
            subprocess.call(["make", "-C", d, "-s"])


def load_golden(name):
    """Raw little-endian arrays written by tests/golden/make_golden.py."""
    for ext, dt in ((".f32", "<f4"), (".f64", "<f8")):
        p = os.path.join(GOLDEN, name + ext)
        if os.path.exists(p):
            return np.fromfile(p, dtype=dt).reshape(-1, 4)
    raise FileNotFoundError(name)


def load_golden64(name):
    return np.fromfile(os.path.join(GOLDEN, name + ".f64"), dtype="<f8").reshape(-1, 4)


def load_golden32(name):
    return np.fromfile(os.path.join(GOLDEN, name + ".f32"), dtype="<f4").reshape(-1, 4)


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def rel_pos_err(x, ref, r_scale):
    """max_i |x_i - ref_i|_inf / max(|ref_i|_2, r_scale)  -- SURVEY.md §8(d)
    'correctness gate' metric; the 1e-4 target of BASELINE.json is on this."""
    x = np.asarray(x, np.float64)[:, :3]
    ref = np.asarray(ref, np.float64)[:, :3]
    d = np.abs(x - ref).max(axis=1)
    return float((d / np.maximum(np.sqrt((ref ** 2).sum(1)), r_scale)).max())


def have_gpu():
    try:
        from nbody3d_amd import capi
        return capi.device_count() > 0
    except Exception:
        return False
