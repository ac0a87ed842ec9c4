#!/bin/bash
# Same-box comparison of several builds of the engine library (tools/ab/*.so and tools/ab/prev, git-ignored):
#   gpurun -- 'bash tools/ab_libs.sh "16384 40002" "--flags 256" prev STATIC_UPS ALIGN tree'
set -u
SIZES=$1; ARGS=$2; shift 2
mkdir -p gpurun_out/ab_libs
for r in 1 2; do
  for lib in "$@"; do
    case $lib in
      tree) unset NB_ENGINE_LIB ;;
      prev) export NB_ENGINE_LIB=$PWD/tools/ab/prev/nbody3d-webgpu_amd/csrc/libnbody3d_hip.so ;;
      *) export NB_ENGINE_LIB=$PWD/tools/ab/libnb_$lib.so ;;
    esac
    timeout -k 10 300 python tools/step_parts.py $SIZES $ARGS > gpurun_out/ab_libs/${lib}_$r.txt 2>&1 || { tail -3 gpurun_out/ab_libs/${lib}_$r.txt; exit 1; }
  done
done
for lib in "$@"; do for r in 1 2; do echo "== $lib $r"; cut -c1-150 gpurun_out/ab_libs/${lib}_$r.txt; done; done
