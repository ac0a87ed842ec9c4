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


def test_force_dist_line_reports_both_protocols_and_a_breakdown_that_adds_up():
    """The N>1 code path on one rank (--force-dist: the process group, the native RCCL communicator, the in-place
    ncclReduceScatter and ncclAllGather all really run): the line names the protocol that ran, times BOTH collectives
    (nb_step_times2), its per-rank parts add up to the step, and the north_star-literal protocol (i-shard + all-gather,
    NB_FLAG_NO_SYM) is measured by the same launch under `also`."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--nbodies", "65536", "--steps", "10", "--warmup", "3",
                        "--no-cpu-baseline", "--no-check"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["value"] and d["config"]["parallelism"] == "pairshard1+reducescatter+allgather", d["config"]
    assert "symwrank" in d["config"]["kernel_variant"]
    x = d["exchange"]
    assert x["reduce_scatter_ms"] > 0 and x["allgather_ms"] > 0 and abs(x["avg_ms"] - x["reduce_scatter_ms"] - x["allgather_ms"]) < 1e-9
    assert x["allgather_bytes_sent_per_rank"] == 65536 * 16 and x["reduce_scatter_bytes_sent_per_rank"] == 0      # one rank: nothing leaves it
    r = d["per_rank"]
    parts = [r[k] for k in ("force_kernel_avg_ms", "sym_reduce_kernel_avg_ms", "reduce_scatter_avg_ms", "integrate_kernel_avg_ms", "allgather_avg_ms")]
    assert all(v > 0 for v in parts) and abs(sum(parts) - r["sum_of_parts_ms"]) < 1e-9
    assert r["sum_of_parts_ms"] <= r["span_ms"] * 1.001
    # the parts and the span are stamped in the event-timed leg; ms_per_step is the wall time of the leg before it (no events): the
    # launches of a rank-form step are not graph-replayed, so the wall-timed step also holds the gaps between its five launches
    assert 0.93 <= r["sum_of_parts_over_span"] <= 1.001 and 0.90 <= r["sum_of_parts_over_ms_per_step"] <= 1.03, r
    lit = d["also"][0]
    assert lit["pass"] and lit["value"] > 0 and lit["config"]["parallelism"] == "ishard1+allgather", lit["config"]
    assert "sym" not in lit["config"]["kernel_variant"]
    assert lit["exchange"]["allgather_ms"] > 0 and lit["exchange"]["reduce_scatter_ms"] is None
    # (three launches of a 1.1 ms step outside a graph: the gaps between them are 2-4 % of it, box to box)
    assert 0.90 <= lit["per_rank"]["sum_of_parts_over_ms_per_step"] <= 1.03, lit["per_rank"]
