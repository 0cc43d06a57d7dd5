#!/bin/bash
# on the GPU box: usage: tools/probe_libs_env.sh <tag> B T reps "<lib> ENV=.. ENV=.." ...   (lib = a directory under build_variants/)
tag=$1; B=$2; T=$3; reps=$4; shift 4
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$tag
for spec in "$@"; do
  lib=${spec%% *}; e=${spec#* }; [ "$e" == "$spec" ] && e=""
  echo "=== $lib $e"
  env $e timeout 300 $R/tools/gemm_probe $R/build_variants/$lib/libxvector_hip.so $B $T $reps 2>&1
done | tee -a $R/gpurun_out/$tag/probe_${B}x${T}.txt
