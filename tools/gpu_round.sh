#!/bin/bash
# Standard GPU pass: full GPU test suite, smoke, bench (default = headline config).
set -u
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2; echo "== $name"; timeout -k 10 "$to" "$@" > "gpurun_out/$name.txt" 2>&1; local rc=$?; echo "== $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name"; tail -5 "gpurun_out/$name.txt"; exit 1; fi; return 0; }
step pytest_gpu 1100 python -m pytest tests -m gpu -x -q
tail -5 gpurun_out/pytest_gpu.txt
step smoke 300 python __graft_entry__.py smoke
tail -2 gpurun_out/smoke.txt
step bench 600 python bench.py
tail -1 gpurun_out/bench.txt | cut -c1-600
