#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops_f16x3.py -x -q > $O/ops.log 2>&1; tail -3 $O/ops.log
echo "=== f32 old kernel (XV_NT_PW=0)"; XV_NT_PW=0 python tools/gemm_bench.py 2>&1 | grep tdnn
echo "=== f32 per-wave staging BK=8"; python tools/gemm_bench.py 2>&1 | grep tdnn
echo "=== f16"; python tools/gemm16_bench.py 2>&1 | grep -E "tdnn|sum"
tools/variant_libs.sh xv_gemm.hip gemm_bench.py -- "pw16:-DXV_PW_BK=16" "pw8w5:-DXV_PW_WGS=5" "pw8:" 2>&1 | grep -E "variant|tdnn2|tdnn4|tdnn5" | tee $O/variants_pw.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python tools/bench_summary.py $O/bench.json
