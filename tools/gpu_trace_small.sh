#!/bin/bash
# kernel time vs inter-kernel gap at small/mid N (rocprofv3 --kernel-trace)
set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_small; mkdir -p $OUT
run() { local tag=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/$tag -o t --output-format csv -- python3 tools/small_n_driver.py "$@" > $OUT/$tag.log 2>&1; local rc=$?; echo "== $tag rc=$rc $(tail -1 $OUT/$tag.log)"; [ $rc -eq 124 ] && exit 1; f=$(find $OUT/$tag -name "*kernel_trace.csv" | head -1); python3 tools/trace_gaps.py "$f"; return 0; }
run n1024 1024 512
run n4096 4096 256
run n4096_nofuse 4096 256 204644 4
run n16384 16384 64
run n16384_sgpr 16384 64 304014 8
run n16384_lds 16384 64 408644
