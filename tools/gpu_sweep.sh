#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 300 python tools/sweep.py --configs "2:1,2:2,2:4,4:2,22:1,22:2,22:4,22:8,24:1,24:2,24:4,24:8,28:2,28:4,28:8" > gpurun_out/sweep1.txt 2>&1; rc=$?
cat gpurun_out/sweep1.txt
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -k "single_step or golden" > gpurun_out/pytest2.txt 2>&1; tail -3 gpurun_out/pytest2.txt
