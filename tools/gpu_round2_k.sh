#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02k; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_trainer.py tests/test_gpu_c_abi.py tests/test_gpu_parallel_rccl.py -x -q -k "not shipped" > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python tools/bench_summary.py $O/bench.json | grep -E "^f32|^f16x3|e2e"
