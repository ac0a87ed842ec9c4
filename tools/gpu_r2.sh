#!/bin/bash
# Round-2 GPU pass.  usage: gpu_r2.sh [tests|scan|bench|all]
set -u
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2; echo "== $name"; timeout -k 10 "$to" "$@" > "gpurun_out/$name.txt" 2>&1; local rc=$?; echo "== $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name"; tail -5 "gpurun_out/$name.txt"; exit 1; fi; return 0; }
what=${1:-all}
if [ "$what" = tests ] || [ "$what" = all ]; then
  step pytest_gpu 1100 python -m pytest tests -m gpu -q --timeout 600
  tail -25 gpurun_out/pytest_gpu.txt
fi
if [ "$what" = scan ] || [ "$what" = all ]; then
  step ubench4 120 tools/ubench4
  cut -c1-210 gpurun_out/ubench4.txt
  step shape_scan 900 python tools/shape_scan.py 1024 2048 4096 8192 16384 40002 65536
  grep -A6 "== N=" gpurun_out/shape_scan.txt | cut -c1-150
fi
if [ "$what" = bench ] || [ "$what" = all ]; then
  step bench 600 python bench.py
  tail -1 gpurun_out/bench.txt | cut -c1-3000
fi
