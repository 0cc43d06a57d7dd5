#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/profile_round.sh r02 2>&1 | tail -12
O=$GRAFT_REPO_ROOT/gpurun_out/profiles_r02
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/ew -- python3 $GRAFT_REPO_ROOT/tools/elementwise_bench.py > $O/ew.log 2>&1)
python3 tools/elementwise_summary.py $(find $O/ew -name "*kernel_trace.csv" | head -1) > $O/elementwise.json; rm -rf $O/ew $O/ew.log
python3 tools/extract_bench.py > $O/extract_bench.txt 2>&1; tail -6 $O/extract_bench.txt
