#!/bin/bash
for n in 1024 2048 4096 8192 16384 32768 65536; do
  steps=$(( 400000000 / n / 100 )); [ $steps -gt 4000 ] && steps=4000; [ $steps -lt 50 ] && steps=50
  timeout -k 10 120 node tests/js/node_bench.js $n $steps | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=%6d  %-28s step %8.2f us  simulate %8.2f us  %.3e pairs/s  %.1f%%' % (d['n'], d['variant'], 1e3*d['ms_per_step'], 1e3*d['ms_per_step_simulate'], d['pairs_per_s'], 100*d['frac_fp32_roofline']))"
done
