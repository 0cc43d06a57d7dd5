#!/bin/bash
# on the GPU box: run tools/gemm_probe against every library under build_variants/ (args after the tag go to the probe)
# usage: tools/probe_all.sh <tag> [B T reps]      env: XV_PROBE_ONLY / XV_PROBE_OPS restrict layers / ops for every variant but "base"
tag=${1:-probe}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$tag
for d in $R/build_variants/*/; do
  n=$(basename $d)
  echo "=== $n"
  if [ "$n" == "base" ]; then
    XV_PROBE_ONLY= XV_PROBE_OPS= timeout 120 $R/tools/gemm_probe $d/libxvector_hip.so ${@:-128 200 20} 2>&1
  else
    timeout 120 $R/tools/gemm_probe $d/libxvector_hip.so ${@:-128 200 20} $R/gpurun_out/$tag/stamps_$n.json 2>&1
  fi
done | tee $R/gpurun_out/$tag/probe.txt
