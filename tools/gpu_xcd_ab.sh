#!/bin/bash
# time + FETCH_SIZE for LDS/SGPR x XCD-remap on/off
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for v in 28 38; do for r in 0 1; do
  if [ $r -eq 0 ]; then export NB_NO_XCD_REMAP=1; else unset NB_NO_XCD_REMAP; fi
  OUT=gpurun_out/xcd_ab/v${v}_r${r}; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o p --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-check --variant $v --jsplit 32 > $OUT/log.txt 2>&1
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --variant $v --jsplit 32 > $OUT/bench.txt 2>&1
  python3 - <<PY
import csv, json
rows=[r for r in csv.DictReader(open("$OUT/p_counter_collection.csv")) if "nb_force" in r["Kernel_Name"]]
f=sum(float(r["Counter_Value"]) for r in rows)/len(rows)
b=json.loads(open("$OUT/bench.txt").read().strip().splitlines()[-1])
print("variant $v remap $r: FETCH_SIZE %.1f KB (x2 = %.1f MB)  K1 %.3f ms  %.2f%% roofline" % (f, 2*f/1024, b["roofline"]["avg_launch_ms"], 100*b["roofline"]["frac"]))
PY
done; done
