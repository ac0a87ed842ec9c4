#!/bin/bash
# Round-2 measurement batch 2: MFMA co-residency ubench, multi-GPU rank shapes on one GPU, profile of the default.
set -u
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2; echo "== $name"; timeout -k 10 "$to" "$@" > "gpurun_out/$name.txt" 2>&1; local rc=$?; echo "== $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name"; tail -5 "gpurun_out/$name.txt"; exit 1; fi; return 0; }
step ubench3 250 tools/ubench3 50000
grep -A4 "(1b)" gpurun_out/ubench3.txt
step sweep_shard8 300 python tools/sweep.py --n 262144 --shard 8 --steps 8 --rounds 3 --configs "0:0,308014:8,308014:16,308014:32,308014:64,308011:128,304014:32"
cat gpurun_out/sweep_shard8.txt
step sweep_shard4 300 python tools/sweep.py --n 262144 --shard 4 --steps 8 --rounds 3 --configs "0:0,308014:16,308014:32"
cat gpurun_out/sweep_shard4.txt
step sweep_shard2 300 python tools/sweep.py --n 262144 --shard 2 --steps 6 --rounds 3 --configs "0:0,308014:8,308014:16"
cat gpurun_out/sweep_shard2.txt
step sweep_weak8 400 python tools/sweep.py --n 1048576 --shard 8 --steps 3 --rounds 2 --configs "0:0,308014:16"
cat gpurun_out/sweep_weak8.txt
step bench_cfg2 300 python bench.py --workload cube --nbodies 65536 --no-cpu-baseline --warmup 300 --steps 100
step bench_cfg2_lds 300 python bench.py --workload cube --nbodies 65536 --variant 28 --no-cpu-baseline --warmup 300 --steps 100
step bench_forcedist 300 python bench.py --force-dist --no-cpu-baseline
step bench_forcedist_overlap 300 python bench.py --force-dist --overlap --no-cpu-baseline
step bench_forcedist_torch 300 python bench.py --force-dist --exchange torch --no-cpu-baseline
for f in bench_cfg2 bench_cfg2_lds bench_forcedist bench_forcedist_overlap bench_forcedist_torch; do grep '^{' gpurun_out/$f.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['config']['kernel_variant'], '%.4e pairs/s' % d['value'], 'frac %.4f' % d['roofline']['frac'], 'K1 %.3f ms' % d['roofline']['avg_launch_ms'], d.get('exchange'))"; done
bash tools/gpu_prof.sh f32
