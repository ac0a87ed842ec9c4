#!/bin/bash
# memory-side counters of the force kernel: previous library vs the tree's
set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
N=$1; shift
OUT=gpurun_out/ab_pmc2_$N; mkdir -p $OUT
for which in old new; do
  if [ $which = old ]; then export NB_ENGINE_LIB=$PWD/tools/ab/prev/nbody3d-webgpu_amd/csrc/libnbody3d_hip.so; else unset NB_ENGINE_LIB; fi
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_SALU"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/${which}_$i -o p --output-format csv -- python3 tools/prof_one.py $N "$@" > $OUT/${which}_$i.log 2>&1 || { echo "$which $i failed"; tail -3 $OUT/${which}_$i.log; }
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for which in ("old", "new"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("%s/%s_*/**/*counter_collection.csv" % (out, which), recursive=True):
        for r in csv.DictReader(open(f)):
            if "nb_force_symw" not in r["Kernel_Name"]:
                continue
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(which, {k: round(v[0] / max(v[1], 1)) for k, v in sorted(agg.items())})
PY
