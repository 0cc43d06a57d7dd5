#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh r01   -> gpurun_out/profiles_r01/{bench.json,kernel_stats.csv,pmc_traffic.json}
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/profiles_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
# 1. kernel trace + stats of the bench command (same flags as the committed bench line, CPU leg off)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_profiled.json 2> $out/stats.log
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 $R/tools/trace_gaps.py $(find $out/stats -name "*kernel_trace.csv" | head -1) 30 > $out/trace_summary.txt
rm -rf $out/stats
# 2. HBM-side traffic: separate PMC passes
cmd="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/write.log
python3 $R/tools/pmc_traffic.py $(find $out/fetch -name "*counter_collection.csv" | head -1) $(find $out/write -name "*counter_collection.csv" | head -1) $out/pmc_traffic.json "$cmd" > /dev/null
rm -rf $out/fetch $out/write $out/*.log
# 3. the bench line itself (with the traffic file in place so `roofline.traffic` is filled), CPU baseline included
mkdir -p $R/profiles && cp $out/pmc_traffic.json $R/profiles/${tag}_pmc_traffic.json
cd $R && python3 bench.py --steps 30 --warmup 5 | tail -1 > $out/bench.json
python3 tools/bench_summary.py $out/bench.json
head -12 $out/trace_summary.txt
