#!/bin/bash
# prev library (tools/ab/prev) vs the tree's over several "SIZES|step_parts args" legs, interleaved twice:  bash tools/ab_legs.sh "262144|--precision f64" ...
set -u
mkdir -p gpurun_out/ab_legs
i=0
for leg in "$@"; do
  i=$((i+1)); sizes=${leg%%|*}; args=${leg#*|}
  for r in 1 2; do
    NB_ENGINE_LIB=$PWD/tools/ab/prev/nbody3d-webgpu_amd/csrc/libnbody3d_hip.so timeout -k 10 300 python tools/step_parts.py $sizes $args > gpurun_out/ab_legs/leg${i}_old_$r.txt 2>&1 || { tail -3 gpurun_out/ab_legs/leg${i}_old_$r.txt; exit 1; }
    timeout -k 10 300 python tools/step_parts.py $sizes $args > gpurun_out/ab_legs/leg${i}_new_$r.txt 2>&1 || { tail -3 gpurun_out/ab_legs/leg${i}_new_$r.txt; exit 1; }
  done
  for w in old new; do for r in 1 2; do echo "== leg $i ($args) $w $r"; cut -c1-160 gpurun_out/ab_legs/leg${i}_${w}_$r.txt; done; done
done
