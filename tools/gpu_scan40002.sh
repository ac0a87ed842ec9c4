#!/bin/bash
set -u
mkdir -p gpurun_out
CFG=""
for js in 10 11 12 13 14 15 16 18 20 21 22 24 25 26 28 30 32; do CFG="$CFG,304014:$js,308014:$js"; done
timeout -k 10 500 python tools/sweep.py --n 40002 --steps 16 --rounds 3 --configs "0:0${CFG}" > gpurun_out/sweep_n40002.txt 2>&1
echo rc=$?
sort -k2 -n gpurun_out/sweep_n40002.txt | head -50
