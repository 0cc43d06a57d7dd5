#!/bin/bash
# on the GPU box: one library (build_variants/base), several run-time switches; usage: tools/probe_env.sh <tag> B T reps "ENV1=a ENV2=b" "ENV..." ...
tag=$1; B=$2; T=$3; reps=$4; shift 4
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$tag
for e in "$@"; do
  echo "=== $e"
  env $e timeout 300 $R/tools/gemm_probe $R/build_variants/base/libxvector_hip.so $B $T $reps 2>&1
done | tee $R/gpurun_out/$tag/probe_${B}x${T}.txt
