#!/usr/bin/env python3
"""Long campaign of tests/api_sequence.py::run_sequence (random API call sequences on a handle and on the CPU oracle side by
side).  Needs a GPU; the time is mostly the single-threaded oracle's.  usage: fuzz_api.py [sequences] [seed0]"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "nbody3d-webgpu_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from api_sequence import run_sequence  # noqa: E402

seqs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
fails, t0 = 0, time.time()
for q in range(seqs):
    try:
        run_sequence(seed0 + q)
    except Exception as e:
        fails += 1
        print("FAIL", repr(e), flush=True)
    if (q + 1) % 50 == 0:
        print("... %d sequences, %d failures, %.0f s" % (q + 1, fails, time.time() - t0), flush=True)
print("api fuzz: %d sequences from seed %d, %d failures, %.0f s" % (seqs, seed0, fails, time.time() - t0))
sys.exit(1 if fails else 0)
