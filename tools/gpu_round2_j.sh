#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02j; mkdir -p $O
export TMPDIR=/tmp
for prec in f32 f16x3; do
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/tr_$prec -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --single-mode --precision $prec --no-cpu-baseline > $O/bench_$prec.json 2> $O/tr_$prec.log)
  python3 tools/trace_timeline.py $(find $O/tr_$prec -name "*kernel_trace.csv" | head -1) 20 > $O/timeline_$prec.txt
  python3 tools/trace_gaps.py $(find $O/tr_$prec -name "*kernel_trace.csv" | head -1) 30 > $O/trace_summary_$prec.txt
  rm -rf $O/tr_$prec
done
