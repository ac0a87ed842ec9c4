#!/bin/bash
# One gpurun call: microbenchmarks, GPU parity tests, first bench line.
# A step that times out (rc 124/137) ends the call; an ordinary failure does not.
set -u
mkdir -p gpurun_out
step() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "== $name" | tee -a gpurun_out/steps.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.txt" 2>&1
  local rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name, stopping"; tail -5 "gpurun_out/$name.txt"; exit 1; fi
  return 0
}
: > gpurun_out/steps.log
step ubench 200 tools/ubench 20000
step pytest_gpu 900 python -m pytest tests -m gpu -x -q
step bench 600 python bench.py --steps 10 --warmup 2
tail -30 gpurun_out/pytest_gpu.txt
tail -3 gpurun_out/bench.txt
