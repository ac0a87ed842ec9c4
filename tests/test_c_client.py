"""A plain-C program linked against libnbody3d_hip.so through include/nbody3d_hip.h: the
drop-in boundary exercised with no Python or Node in between (prompt section 2)."""
import os
import subprocess

import pytest

from conftest import GOLDEN, PKG, ROOT, have_gpu

SRC = os.path.join(ROOT, "tests", "c", "abi_client.c")
EXE = os.path.join(ROOT, "tests", "c", "abi_client")
CSRC = os.path.join(PKG, "csrc")


def build():
    lib = os.path.join(CSRC, "libnbody3d_hip.so")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(SRC), os.path.getmtime(lib)):
        subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
                               "-L", CSRC, "-lnbody3d_hip", "-lm", "-Wl,-rpath," + CSRC])
    return EXE


def run():
    p = subprocess.run([build(), GOLDEN], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "FAIL" not in p.stdout, p.stdout + p.stderr
    return p.stdout


@pytest.mark.skipif(have_gpu(), reason="checks the no-device contract")
def test_c_client_reports_missing_device():
    out = run()
    assert "no-device contract" in out and "ok plan query without a device" in out


@pytest.mark.gpu
def test_c_client_runs_the_fixture_on_the_gpu():
    out = run()
    assert "fixture after 10 steps" in out and "ok diagnostics" in out
    assert "ok frame feed" in out and "ok step times" in out
    assert "ok rccl-attached handle" in out and "ok multi handle in RCCL mode" in out
