#!/bin/bash
# Same-box A/B of the engine library: tools/ab/prev (built from an earlier commit, git-ignored) against the tree's, interleaved.
#   gpurun -- 'bash tools/ab_prev.sh "13000 16384 40002" [extra step_parts args]'
set -u
SIZES=${1:-"16384 40002"}; shift || true
mkdir -p gpurun_out/ab_prev
for r in 1 2; do
  NB_ENGINE_LIB=$PWD/tools/ab/prev/nbody3d-webgpu_amd/csrc/libnbody3d_hip.so timeout -k 10 300 python tools/step_parts.py $SIZES > gpurun_out/ab_prev/old_$r.txt 2>&1 || { tail -3 gpurun_out/ab_prev/old_$r.txt; exit 1; }
  timeout -k 10 300 python tools/step_parts.py $SIZES "$@" > gpurun_out/ab_prev/new_$r.txt 2>&1 || { tail -3 gpurun_out/ab_prev/new_$r.txt; exit 1; }
done
for f in old_1 new_1 old_2 new_2; do echo "== $f"; cut -c1-150 gpurun_out/ab_prev/$f.txt; done
