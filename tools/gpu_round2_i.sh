#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02i; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q -k "not shipped" > $O/tests.log 2>&1; tail -2 $O/tests.log
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/ew -- python3 $GRAFT_REPO_ROOT/tools/elementwise_bench.py > $O/ew.log 2>&1)
python3 tools/elementwise_summary.py $(find $O/ew -name "*kernel_trace.csv" | head -1) > $O/elementwise.json; rm -rf $O/ew
python3 -c "
import json; d=json.load(open('$O/elementwise.json'))
for k in d['kernels']: print('%-52s ch %4d  %7.1f us  %6.1f MB  %5.2f TB/s  %.3f' % (k['kernel'][:52], k['channels'], k['avg_us'], k['algorithmic_mb'], k['tb_per_s'], k['frac_of_8tbs']))"
python bench.py --single-mode --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python tools/bench_summary.py $O/bench.json | head -2
