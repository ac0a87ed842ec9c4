#!/bin/bash
set -u
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2; echo "== $name"; timeout -k 10 "$to" "$@" > "gpurun_out/$name.txt" 2>&1; local rc=$?; echo "== $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name"; tail -5 "gpurun_out/$name.txt"; exit 1; fi; return 0; }
step pytest_new 600 python -m pytest tests/test_round2_gpu.py -m gpu -q --timeout 600 -k "rehearsal or graph or frame"
tail -3 gpurun_out/pytest_new.txt
step node_default 300 node tests/js/node_default_workload.js 2000
step node_default_feed 300 node tests/js/node_default_workload.js 2000 feed
cat gpurun_out/node_default.txt gpurun_out/node_default_feed.txt
step node_bench 300 node tests/js/node_bench.js
cat gpurun_out/node_bench.txt | tail -15
step energy 600 python tools/energy_horizon.py --steps 1000 --every 100 --out gpurun_out/energy_horizon_r2.json
tail -5 gpurun_out/energy.txt
