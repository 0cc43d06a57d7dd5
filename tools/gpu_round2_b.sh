#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pooling.py -x -q > $O/pooling.log 2>&1; tail -5 $O/pooling.log
tools/variant_libs.sh xv_gemm.hip gemm_bench.py -- "base:" "w5:-DXV_WGS_PER_CU=5" "prio:-DXV_NT_SETPRIO=1" "noreads:-DXV_NT_ABLATE=1" "nostage:-DXV_NT_ABLATE=2" "mfmaonly:-DXV_NT_ABLATE=3" "bk32:-DXV_TILE_K=32" "base2:" 2>&1 | tee $O/variants_f32.log
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmcw -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py > $GRAFT_REPO_ROOT/$O/pmcw.log 2>&1)
python3 tools/pmc_summary.py $(find $O/pmcw -name "*counter_collection.csv" | head -1) | tee $O/pmc_wait.txt
rm -rf $O/pmcw
