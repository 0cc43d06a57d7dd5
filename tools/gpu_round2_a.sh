#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
nproc > gpurun_out/r02a/host.txt; free -g >> gpurun_out/r02a/host.txt
(cd /tmp && TMPDIR=/tmp timeout 120 rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r02a/counters.txt 2>&1)
timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q -s > gpurun_out/r02a/fullsize.log 2>&1; echo "fullsize rc=$?"
tail -30 gpurun_out/r02a/fullsize.log
timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"
python tools/bench_summary.py gpurun_out/r02a/bench.json; tail -3 gpurun_out/r02a/bench.err
timeout 300 python tools/gemm_bench.py > gpurun_out/r02a/gemm_bench_f32.log 2>&1; cat gpurun_out/r02a/gemm_bench_f32.log
timeout 1500 tools/pmc_mfma.sh gpurun_out/r02a/pmc_mfma; head -c 3000 gpurun_out/r02a/pmc_mfma/pmc_mfma.json
