"""No-GPU checks of the C-ABI library: it loads, exports every symbol the
header declares, and fails loudly (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT, have_gpu
from nbody3d_amd import capi

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nbody3d-webgpu_amd", "csrc")


def header_symbols():
    src = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include"))) if h.endswith(".h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nb_[a-z0-9_]+)\s*\(", src)) - {"nb_exchange_fn"})


def test_library_exports_every_declared_symbol():
    L = capi.load_library()
    syms = header_symbols()
    assert sorted(capi.SYMBOLS) == syms
    for s in syms:
        assert hasattr(L, s), s


def test_abi_version_and_config_layout():
    assert capi.abi_version() == capi.ABI_VERSION == 2 and capi.abi_minor() >= capi.ABI_MINOR == 3
    # layout must match the C struct: 4*4 + 8 + 3*4 (+4 pad) + 2*8 + 3*4 + 5*4 = 88
    assert C.sizeof(capi.nb_config) == 88


def test_bad_config_is_rejected_without_touching_a_device():
    L = capi.load_library()
    h = C.c_void_p()
    cfg = capi.nb_config()
    cfg.struct_size = C.sizeof(capi.nb_config)
    cfg.n = 0
    assert L.nb_create(C.byref(cfg), C.byref(h)) == 1 and not h.value
    assert b"n must be" in L.nb_last_error(None)
    cfg.n = 16
    cfg.eps2 = -1.0
    assert L.nb_create(C.byref(cfg), C.byref(h)) == 1
    cfg.eps2 = 0.0
    cfg.shard_begin, cfg.shard_count = 10, 10
    assert L.nb_create(C.byref(cfg), C.byref(h)) == 1
    cfg.struct_size = 8
    assert L.nb_create(C.byref(cfg), C.byref(h)) == 1
    assert L.nb_step(None, 1) == 1 and L.nb_sync(None) == 1
    L.nb_destroy(None)  # no-op


@pytest.mark.skipif(have_gpu(), reason="checks the no-device error path")
def test_no_device_is_an_ordinary_error_not_a_fallback():
    """Reference: alert + return when WebGPU is missing (nbody3d.js:151-155).
    Here: NB_ERR_NO_DEVICE and an exception; nothing computes on the CPU."""
    assert capi.device_count() == 0
    with pytest.raises(capi.NBodyError) as e:
        capi.Simulation(1024)
    assert e.value.code == 2 and "no CPU fallback" in str(e.value)


@pytest.mark.skipif(have_gpu(), reason="checks the no-device error path")
def test_bench_refuses_to_run_without_a_gpu():
    """bench.py must not fall back to any CPU path (the oracle is only its baseline leg)."""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0 and "no GPU visible" in (p.stderr + p.stdout)
    assert not any(l.startswith("{") for l in p.stdout.splitlines())


@pytest.mark.skipif(have_gpu(), reason="checks the no-device error path")
def test_bench_multi_gpu_self_launch_fails_cleanly_without_gpus():
    """`bench.py --gpus 2` with no launcher: the GPU-free parent starts two ranks through
    torch.distributed.run; with no GPU each rank must stop at the device check (no hang, no JSON)."""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "no GPU visible" in (p.stderr + p.stdout), (p.stderr + p.stdout)[-1500:]
    assert not any(l.startswith("{") for l in p.stdout.splitlines())


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
