#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 300 python tools/sweep.py --configs "22:16,24:8,24:16,24:32,28:8,28:16,28:32" > gpurun_out/sweep2.txt 2>&1; rc=$?; cat gpurun_out/sweep2.txt; [ $rc -eq 124 ] && exit 1
timeout -k 10 300 python tools/sweep.py --shard 8 --steps 30 --configs "22:8,22:16,22:32,24:16,24:32,24:64,28:32,28:64,2:8" > gpurun_out/sweep2_shard8.txt 2>&1; rc=$?; cat gpurun_out/sweep2_shard8.txt; [ $rc -eq 124 ] && exit 1
timeout -k 10 300 python tools/sweep.py --n 65536 --steps 50 --configs "22:4,22:8,22:16,24:8,24:16,24:32,28:16,28:32,1:4" > gpurun_out/sweep2_n65536.txt 2>&1; rc=$?; cat gpurun_out/sweep2_n65536.txt; [ $rc -eq 124 ] && exit 1
bash tools/gpu_prof.sh --variant 28 --jsplit 8
