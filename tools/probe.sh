#!/bin/bash
# On the GPU box: the stand-alone C++ GEMM timer (tools/gemm_probe, built by tools/variant.sh unit) over builds of the library and/or run-time
# switches - starts in milliseconds, so one GPU call compares many variants on ONE box.  Output: gpurun_out/<tag>/probe*.txt
#   tools/probe.sh all  <tag> [B T reps]                              every library under build_variants/ (XV_PROBE_ONLY / XV_PROBE_OPS apply to all but "base")
#   tools/probe.sh env  <tag> B T reps "ENV1=a ENV2=b" "ENV..." ...    build_variants/base under several environments
#   tools/probe.sh libs <tag> B T reps "<lib> ENV=.. ENV=.." ...       named libraries (directories under build_variants/), each with its environment
mode=$1; tag=${2:-probe}; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$tag
case $mode in
all)
  for d in $R/build_variants/*/; do
    n=$(basename $d)
    echo "=== $n"
    if [ "$n" == "base" ]; then
      XV_PROBE_ONLY= XV_PROBE_OPS= timeout 120 $R/tools/gemm_probe $d/libxvector_hip.so ${@:-128 200 20} 2>&1
    else
      timeout 120 $R/tools/gemm_probe $d/libxvector_hip.so ${@:-128 200 20} $R/gpurun_out/$tag/stamps_$n.json 2>&1
    fi
  done | tee $R/gpurun_out/$tag/probe.txt ;;
env)
  B=$1; T=$2; reps=$3; shift 3
  for e in "$@"; do
    echo "=== $e"
    env $e timeout 300 $R/tools/gemm_probe $R/build_variants/base/libxvector_hip.so $B $T $reps 2>&1
  done | tee $R/gpurun_out/$tag/probe_${B}x${T}.txt ;;
libs)
  B=$1; T=$2; reps=$3; shift 3
  for spec in "$@"; do
    lib=${spec%% *}; e=${spec#* }; [ "$e" == "$spec" ] && e=""
    echo "=== $lib $e"
    env $e timeout 300 $R/tools/gemm_probe $R/build_variants/$lib/libxvector_hip.so $B $T $reps 2>&1
  done | tee -a $R/gpurun_out/$tag/probe_${B}x${T}.txt ;;
*) echo "usage: tools/probe.sh all|env|libs <tag> ..."; exit 1 ;;
esac
