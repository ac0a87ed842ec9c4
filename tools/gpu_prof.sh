#!/bin/bash
# rocprofv3 passes for a bench command (kernel trace/stats, then PMC passes on their own).
# usage: gpu_prof.sh <tag> [bench.py args...]   -> gpurun_out/prof_<tag>/
set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
ARGS="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-check --no-also $*"
TRACE_ARGS="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check --no-also $*"   # the stats pass: long enough for the clocks to settle (tools/pmc_summary.py picks the 20 timed launches out of the 26)
run() { local name=$1; shift; timeout -k 10 300 rocprofv3 "$@" -d $OUT/$name -o $name --output-format csv -- python3 $ARGS > $OUT/$name.log 2>&1; local rc=$?; echo "$name rc=$rc"; grep '^{' $OUT/$name.log | tail -1 | cut -c1-300; [ $rc -ge 124 ] && { echo "timeout / signal in $name: stopping"; exit 1; }; return 0; }
ARGS_KEEP="$ARGS"; ARGS="$TRACE_ARGS"; run trace --kernel-trace --stats; ARGS="$ARGS_KEEP"
run pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run pmc_sq2 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
# rocprofv3 nests its output one level deeper (<dir>/<host>/...): flatten for tools/pmc_summary.py
for p in trace pmc_sq1 pmc_sq2 pmc_fetch pmc_write; do
  for f in $(find $OUT/$p -name "*.csv"); do cp "$f" "$OUT/$p/$(basename $f)"; done
done
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_line.json
find $OUT -maxdepth 2 -name "*.csv" | head -20
