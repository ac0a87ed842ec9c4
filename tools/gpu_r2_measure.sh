#!/bin/bash
# Round-2 measurement batch (one gpurun call): K2 on HBM, j-split sweep at the headline size,
# configs 2 and 5, power-of-two size scan, rocprofv3 passes of the f32 and f64 headline kernels.
set -u
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2; echo "== $name"; timeout -k 10 "$to" "$@" > "gpurun_out/$name.txt" 2>&1; local rc=$?; echo "== $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name"; tail -5 "gpurun_out/$name.txt"; exit 1; fi; return 0; }
step k2_hbm 300 python tools/k2_hbm.py
cat gpurun_out/k2_hbm.txt
step sweep_js 300 python tools/sweep.py --n 262144 --steps 6 --rounds 3 --configs "308014:4,308014:8,308014:12,308014:16,308014:32,308011:32,304014:16,208011:32"
cat gpurun_out/sweep_js.txt
step bench_cfg2 300 python bench.py --workload cube --nbodies 65536 --no-cpu-baseline
step bench_cfg2_lds 300 python bench.py --workload cube --nbodies 65536 --variant 28 --no-cpu-baseline
step bench_cfg5_f64 400 python bench.py --precision f64 --steps 10 --no-cpu-baseline
for f in bench_cfg2 bench_cfg2_lds bench_cfg5_f64; do grep '^{' gpurun_out/$f.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['config']['kernel_variant'], '%.4e pairs/s' % d['value'], 'frac %.4f' % d['roofline']['frac'], 'K1 %.3f ms' % d['roofline']['avg_launch_ms'])"; done
step size_scan 600 python tools/size_scan.py
cat gpurun_out/size_scan.txt
bash tools/gpu_prof.sh f32
bash tools/gpu_prof.sh f64 --precision f64
