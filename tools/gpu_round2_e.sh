#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02e; mkdir -p $O
export TMPDIR=/tmp
for prec in f32 f16x3; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$prec -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --single-mode --precision $prec --no-cpu-baseline > $O/bench_profiled_$prec.json 2> $O/stats_$prec.log)
  cp $(find $O/stats_$prec -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$prec.csv
  python3 tools/trace_gaps.py $(find $O/stats_$prec -name "*kernel_trace.csv" | head -1) 30 > $O/trace_summary_$prec.txt
  rm -rf $O/stats_$prec
done
head -40 $O/kernel_stats_f32.csv | cut -d, -f1-4,7 | cut -c1-150
python3 tools/loader_scale.py --procs 8 --threads 8 --batches 80 > $O/loader_scale.json; cat $O/loader_scale.json
