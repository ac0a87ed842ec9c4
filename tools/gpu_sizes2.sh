#!/bin/bash
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for n in 1024 2002 4096 8192 16384 20000 32768 40002 65536 100000 262144 500010; do
  steps=$(( 4000000000 / n / n * 40 + 20 )); [ $steps -gt 3000 ] && steps=3000
  timeout -k 10 120 node tests/js/node_bench.js $n $steps | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=%6d  %-30s step %9.2f us  simulate %9.2f us  %.3e pairs/s  %.1f%%' % (d['n'], d['variant'], 1e3*d['ms_per_step'], 1e3*d['ms_per_step_simulate'], d['n']*(d['n']-1)/(1e-3*d['ms_per_step_simulate']), 100*d['n']*(d['n']-1)/(1e-3*d['ms_per_step_simulate'])/7.865e12))"
done
python tools/sweep.py --shard 8 --steps 30 --rounds 3 --configs "0:0"
timeout -k 10 120 node tests/js/node_default_workload.js 1000
