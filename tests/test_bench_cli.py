"""The bench.py contract: one JSON line with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def test_bench_emits_one_well_formed_json_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nbodies", "16384", "--steps", "5", "--warmup", "2",
                        "--cpu-seconds", "0.5"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "pair-interactions/s" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 16384 * 16383 * 5 / (d["ms_per_step"] * 5e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert d["check"]["pass"] is True and d["check"]["max_rel_pos_err_vs_f64_oracle"] < 1e-4
