#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh r02
#   -> gpurun_out/profiles_r02/{bench.json, bench_profiled_<prec>.json, kernel_stats_<prec>.csv, trace_summary_<prec>.txt, step_timeline_<prec>.txt,
#      pmc_traffic.json}   for prec in f32 (the headline mode) and f16x3 (the opt-in mode); copy what is to be judged into profiles/
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/profiles_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for prec in f32 f16x3; do
  # 1. kernel trace + stats of the bench command in this mode (CPU leg off, one mode per run so the stats are not mixed)
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$prec -- python3 $R/bench.py --steps 30 --warmup 5 --single-mode --precision $prec --no-cpu-baseline > $out/bench_profiled_$prec.json 2> $out/stats_$prec.log
  cp $(find $out/stats_$prec -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$prec.csv
  python3 $R/tools/trace_gaps.py $(find $out/stats_$prec -name "*kernel_trace.csv" | head -1) 30 > $out/trace_summary_$prec.txt
  python3 $R/tools/trace_timeline.py $(find $out/stats_$prec -name "*kernel_trace.csv" | head -1) 20 > $out/step_timeline_$prec.txt
  rm -rf $out/stats_$prec
done
# 2. HBM-side traffic of the GEMM kernels: separate PMC passes (no trace domains) over the default (headline) run
cmd="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/fetch.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/write.log
python3 $R/tools/pmc_traffic.py $(find $out/fetch -name "*counter_collection.csv" | head -1) $(find $out/write -name "*counter_collection.csv" | head -1) $out/pmc_traffic.json "$cmd" > /dev/null
rm -rf $out/fetch $out/write $out/*.log
# 3. the bench line itself (with the traffic file in place so `roofline.traffic` is filled), CPU baseline included
mkdir -p $R/profiles && cp $out/pmc_traffic.json $R/profiles/${tag}_pmc_traffic.json
cd $R && timeout 900 python3 bench.py --steps 30 --warmup 5 | tail -1 > $out/bench.json
python3 tools/bench_summary.py $out/bench.json
