#!/bin/bash
# One parametrised GPU pass (replaces round 2's fourteen one-off gpu_*.sh scripts).  Run through gpurun:
#   gpurun --timeout 1100 -- 'bash tools/gpu.sh tests bench'
# Legs, in the order given: tests [pytest args] | new (only the test files named in NB_NEW_TESTS) | smoke | bench | sizes N... | shapes N... |
# prof (rocprofv3 kernel trace + PMC passes of the default bench line).  A leg that times out stops the pass:
# no further GPU step is started after a kill.
set -u
mkdir -p gpurun_out
# A leg that times out (124 / 137) or dies on a signal (rc >= 128: 134 / 139 is how a GPU memory fault or an abort surfaces) ends the
# pass at once -- no further GPU leg is started after a kill or a fault -- and any failing leg makes the script exit non-zero.
FAILED=0
step() { local name=$1 to=$2; shift 2; echo "== $name: $*"; timeout -k 10 "$to" "$@" > "gpurun_out/$name.txt" 2>&1; local rc=$?
         echo "== $name rc=$rc"
         if [ $rc -ge 124 ]; then echo "TIMEOUT / SIGNAL in $name (rc=$rc): stopping the pass"; tail -5 "gpurun_out/$name.txt"; exit 1; fi
         if [ $rc -ne 0 ]; then FAILED=1; fi; return 0; }
while [ $# -gt 0 ]; do
  leg=$1; shift
  case $leg in
    tests) step pytest_gpu 1100 python -m pytest tests -m gpu -x -q --durations=15; tail -25 gpurun_out/pytest_gpu.txt ;;
    new)   step pytest_new 900 python -m pytest ${NB_NEW_TESTS:-tests/test_sym_gpu.py} -m gpu -x -q --durations=10; tail -25 gpurun_out/pytest_new.txt ;;
    smoke) step smoke 300 python __graft_entry__.py smoke; tail -2 gpurun_out/smoke.txt ;;
    bench) step bench 600 python bench.py; tail -1 gpurun_out/bench.txt | cut -c1-3000 ;;
    sizes) args=(); while [ $# -gt 0 ] && [[ $1 =~ ^[0-9]+$ ]]; do args+=("$1"); shift; done
           step size_scan 900 python tools/size_scan.py "${args[@]}"; cat gpurun_out/size_scan.txt ;;
    shapes) args=(); while [ $# -gt 0 ] && [[ $1 =~ ^[0-9]+$ ]]; do args+=("$1"); shift; done
           step shape_scan 1000 python tools/shape_scan.py "${args[@]}"; grep -c . gpurun_out/shape_scan.txt ;;
    prof)  tag=$1; shift; pargs=(); while [ $# -gt 0 ] && [[ $1 == --* ]]; do pargs+=("$1" "$2"); shift 2; done; bash tools/gpu_prof.sh "$tag" "${pargs[@]}" ;;
    py)    script=$1; shift; step "$(basename "$script" .py)" 900 python "$script"; tail -40 "gpurun_out/$(basename "$script" .py).txt" ;;
    *) echo "unknown leg $leg"; exit 2 ;;
  esac
done
exit $FAILED
