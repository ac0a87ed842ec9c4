#!/bin/bash
# rocprofv3 counter passes of tools/prof_one.py with the previous library (tools/ab/prev) and the tree's: where does a kernel's time go?
#   gpurun -- 'bash tools/ab_pmc.sh 40002 256'
set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
N=$1; shift
OUT=gpurun_out/ab_pmc_$N; mkdir -p $OUT
for which in old new; do
  if [ $which = old ]; then export NB_ENGINE_LIB=$PWD/tools/ab/prev/nbody3d-webgpu_amd/csrc/libnbody3d_hip.so; else unset NB_ENGINE_LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/${which}_sq1 -o p --output-format csv -- python3 tools/prof_one.py $N "$@" > $OUT/${which}_sq1.log 2>&1 || { echo "$which sq1 failed"; tail -3 $OUT/${which}_sq1.log; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $OUT/${which}_sq2 -o p --output-format csv -- python3 tools/prof_one.py $N "$@" > $OUT/${which}_sq2.log 2>&1 || { echo "$which sq2 failed"; tail -3 $OUT/${which}_sq2.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for which in ("old", "new"):
    for p in ("sq1", "sq2"):
        files = glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, which, p), recursive=True)
        agg = collections.defaultdict(lambda: [0.0, 0])
        for f in files:
            for r in csv.DictReader(open(f)):
                if "nb_force_symw" not in r["Kernel_Name"]:
                    continue
                a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        print(which, p, {k: round(v[0] / max(v[1], 1)) for k, v in sorted(agg.items())})
PY
