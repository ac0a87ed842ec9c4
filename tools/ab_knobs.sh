#!/bin/bash
# Same-box comparison of planner constants (tuning build: NB_MODEL_* from the environment), every arm in its own process, three rounds, interleaved:
#   gpurun -- 'bash tools/ab_knobs.sh "16384 262144" "" "base:NB_MODEL_ODD_XCD=1 NB_MODEL_XCD_START=0" "xcd:NB_MODEL_ODD_XCD=0.978"'
set -u
SIZES=$1; ARGS=$2; shift 2
mkdir -p gpurun_out/ab_knobs
export NB_ENGINE_LIB=$PWD/nbody3d-webgpu_amd/csrc/libnbody3d_hip_tuning.so
for r in 1 2 3; do
  for arm in "$@"; do
    name=${arm%%:*}
    env ${arm#*:} timeout -k 10 300 python tools/step_parts.py $SIZES $ARGS > gpurun_out/ab_knobs/${name}_$r.txt 2>&1 || { tail -3 gpurun_out/ab_knobs/${name}_$r.txt; exit 1; }
  done
done
for arm in "$@"; do name=${arm%%:*}; for r in 1 2 3; do echo "== $name $r"; cut -c1-170 gpurun_out/ab_knobs/${name}_$r.txt; done; done
