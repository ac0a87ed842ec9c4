#!/bin/bash
# Same-box comparison of several builds of the engine library, every arm run twice, interleaved.  An arm is `tree` (the built tree),
# `prev` (tools/ab/prev, a `git archive` of another commit built in place), NAME (tools/ab/libnb_NAME.so) or NAME=path/to/lib.so:
#   gpurun -- 'bash tools/ab_libs.sh "16384 40002" "--flags 256" prev a64=tools/ab/a64/nbody3d-webgpu_amd/csrc/libnbody3d_hip.so tree'
# (tools/ab/ is git-ignored.)
set -u
SIZES=$1; ARGS=$2; shift 2
mkdir -p gpurun_out/ab_libs
for r in 1 2; do
  for arm in "$@"; do
    name=${arm%%=*}
    case $arm in
      tree) unset NB_ENGINE_LIB ;;
      prev) export NB_ENGINE_LIB=$PWD/tools/ab/prev/nbody3d-webgpu_amd/csrc/libnbody3d_hip.so ;;
      *=*) export NB_ENGINE_LIB=$PWD/${arm#*=} ;;
      *) export NB_ENGINE_LIB=$PWD/tools/ab/libnb_$arm.so ;;
    esac
    timeout -k 10 300 python tools/step_parts.py $SIZES $ARGS > gpurun_out/ab_libs/${name}_$r.txt 2>&1 || { tail -3 gpurun_out/ab_libs/${name}_$r.txt; exit 1; }
  done
done
for arm in "$@"; do name=${arm%%=*}; for r in 1 2; do echo "== $name $r"; cut -c1-170 gpurun_out/ab_libs/${name}_$r.txt; done; done
