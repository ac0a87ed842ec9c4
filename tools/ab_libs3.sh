#!/bin/bash
# several builds of the engine library on one box, interleaved twice:  bash tools/ab_libs3.sh "SIZES" [step_parts args] -- name=path ...
set -u
sizes=$1; shift
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
mkdir -p gpurun_out/ab3
for r in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    NB_ENGINE_LIB=$PWD/$lib timeout -k 10 300 python tools/step_parts.py $sizes "${args[@]}" > gpurun_out/ab3/${name}_$r.txt 2>&1 || { tail -3 gpurun_out/ab3/${name}_$r.txt; exit 1; }
  done
done
for spec in "$@"; do name=${spec%%=*}; for r in 1 2; do echo "== $name $r"; cut -c1-170 gpurun_out/ab3/${name}_$r.txt; done; done
