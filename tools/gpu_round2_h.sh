#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q > $O/ops.log 2>&1; tail -2 $O/ops.log
python tools/gemm_bench.py 2>&1 | grep tdnn | tee $O/gemm_bench_f32.log
python bench.py --single-mode --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python tools/bench_summary.py $O/bench.json
