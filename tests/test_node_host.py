"""The JavaScript host side (north_star: 'host code stays in JavaScript (Node)
calling through a thin C-ABI FFI'): N-API addon + nbody3d_hip.js wrapper."""
import json
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

NODE = shutil.which("node")
SCRIPT = os.path.join(ROOT, "tests", "js", "node_tests.js")
ADDON = os.path.join(ROOT, "nbody3d-webgpu_amd", "js", "addon", "nb_napi.node")


def run(mode):
    if not os.path.exists(ADDON):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "nbody3d-webgpu_amd", "js"), "-s"])
    p = subprocess.run([NODE, SCRIPT, mode], capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert line, "node produced no result: rc=%d\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    res = json.loads(line[-1])
    failed = {k: v for k, v in res["results"].items() if not v["pass"]}
    assert res["ok"] and p.returncode == 0, failed
    return res


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_node_wrapper_and_js_oracle_cpu():
    res = run("cpu")
    assert "js_oracle_bit_exact_vs_golden_s10" in res["results"]


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_node_host_path_on_gpu():
    res = run("gpu")
    assert res["results"]["gpu_s100_vs_f64_oracle_le_1e-4"]["pass"]
