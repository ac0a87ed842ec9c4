#!/bin/bash
# rocprofv3 passes for the bench command (kernel trace/stats, then PMC passes on their own).
set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof; mkdir -p $OUT
ARGS="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-check $*"
run() { local name=$1; shift; timeout -k 10 300 rocprofv3 "$@" -d $OUT/$name -o $name --output-format csv -- python3 $ARGS > $OUT/$name.log 2>&1; local rc=$?; echo "$name rc=$rc"; tail -2 $OUT/$name.log | cut -c1-400; [ $rc -eq 124 ] && exit 1; return 0; }
run trace --kernel-trace --stats
run pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run pmc_sq2 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
find $OUT -name "*.csv" | head -30
