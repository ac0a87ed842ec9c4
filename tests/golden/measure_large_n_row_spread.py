#!/usr/bin/env python3
"""Records how far the fp32 ORACLE's own ascending-j sum (the reference's order, nbody3d.js:263-268) sits from an fp64
direct sum on the rows the large-N GPU tests sample -> tests/golden/large_n_row_spread.json.

At N = 2,000,000 / 4,194,304 a row's acceleration is a sum of millions of binary32 terms: the order of the additions alone
moves the result by ~1e-5 of the row's largest component.  The engine's symmetric pass adds the same per-pair products in
another order (residents in registers, traveler layers, K2), so it cannot be held to a tolerance tighter than what the
reference's own order achieves.  tests/test_sym_gpu.py holds every sampled row to
    engine_err <= max(2e-5, 2 * oracle_f32_err)          (errors relative to the row's largest |component| of the fp64 sum)
with oracle_f32_err read from the file this script writes -- the same rule as tests/golden/galaxy40002_spread.json at N = 40,002.
For scale it also records what the binary32 per-pair TERMS themselves cost (the same terms added exactly, in fp64: ~5e-9) and
what a pairwise fp32 summation of them gives (~2e-7): at these sizes the error of a row is all summation ORDER -- the reference's
ascending-j loop adds terms of ~a/N to a running sum of ~a (stagnation: 1e-5 .. 5e-4), and the engine's error is set by how many
terms one accumulator takes in sequence (nb_force_symw flushes its resident sums to a second level every 64 sweeps: DESIGN.md §3).
CPU only, ~20 s.

    python tests/golden/measure_large_n_row_spread.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))
from nbody3d_amd import ic  # noqa: E402
from oracle import oracle  # noqa: E402

# (n, seed, rows): exactly what test_two_million_bodies_take_the_symmetric_pass / test_four_million_bodies_keep_the_symmetric_pass sample
CASES = [(2000000, 7, (0, 1, 999999, 1234567, 1999999)), (4194304, 8, (0, 1, 2097151, 3456789, 4194303))]


def fp64_row(x, m, i):
    d = x - x[i]
    r2 = (d * d).sum(1) + 1e-4
    return (m[:, None] * d / (r2 * np.sqrt(r2))[:, None]).sum(0)


def f32_terms(b, i):
    """The binary32 per-pair terms of row i, operation by operation as oracle/nb_oracle.c::pair_f32 (nbody3d.js:233-236)."""
    d = b[:, :3] - b[i, :3]
    r2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2] + np.float32(1e-4)
    inv = np.float32(1.0) / np.sqrt(r2 * r2 * r2)
    return (b[:, 3] * inv)[:, None] * d


out = {"metric": "max_c |a_f32[i, c] - a_f64[i, c]| / max_c |a_f64[i, c]| for the sampled row i; G = 1, eps2 = 1e-4, ic.plummer(n, seed)",
       "generator": "tests/golden/measure_large_n_row_spread.py (oracle/nb_oracle.c nbo_accel_f32 on single rows vs a numpy fp64 direct sum)",
       "cases": []}
for n, seed, rows in CASES:
    b, _ = ic.plummer(n, seed=seed)
    x = b[:, :3].astype(np.float64)
    m = b[:, 3].astype(np.float64)
    case = {"n": n, "seed": seed, "rows": {}}
    for i in rows:
        want = fp64_row(x, m, i)
        got = oracle.accel_f32(b, 1.0, i0=i, i1=i + 1)[0, :3].astype(np.float64)
        t32 = f32_terms(b, i)
        seq = t32.sum(0, dtype=np.float32).astype(np.float64)          # numpy adds the rows of an (n, 3) array one after the other: ascending j
        assert np.array_equal(seq, got), "the numpy restatement of the terms no longer reproduces the oracle's row"
        pw = np.array([np.ascontiguousarray(t32[:, c]).sum(dtype=np.float32) for c in range(3)], np.float64)      # contiguous axis: pairwise
        exact = t32.astype(np.float64).sum(0)
        scale = np.abs(want).max()
        case["rows"][str(i)] = {"oracle_f32_err": float(np.abs(got - want).max() / scale),
                                "pairwise_f32_err": float(np.abs(pw - want).max() / scale),
                                "f32_terms_added_exactly_err": float(np.abs(exact - want).max() / scale),
                                "a_f64": [float(c) for c in want]}
        print(n, i, {k: v for k, v in case["rows"][str(i)].items() if k != "a_f64"}, flush=True)
    out["cases"].append(case)
json.dump(out, open(os.path.join(HERE, "large_n_row_spread.json"), "w"), indent=1)
