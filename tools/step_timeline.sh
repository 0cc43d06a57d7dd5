#!/bin/bash
# on the GPU box: one-step kernel timeline + per-queue gap summary of bench.py under the given environment
# usage: tools/step_timeline.sh <outfile-prefix> [ENV=.. ...] [-- bench args]
out=$1; shift
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" == "--" ] && shift
R=${GRAFT_REPO_ROOT:-$PWD}
case $out in /*) ;; *) out=$R/$out ;; esac
export TMPDIR=/tmp
for e in "${envs[@]}"; do export "$e"; done
d=$(mktemp -d /tmp/xvtl.XXXX)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --steps 24 --warmup 5 --single-mode --no-cpu-baseline "$@" > $out.json 2> $out.log)
tr=$(find $d -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $tr 15 > $out.timeline.txt 2>> $out.log
python3 $R/tools/trace_gaps.py $tr 24 > $out.gaps.txt 2>> $out.log
python3 $R/tools/bench_summary.py $out.json >> $out.timeline.txt 2>&1
rm -rf $d
