#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
tools/variant_libs.sh xv_gemm.hip gemm_bench.py -- "base:" "prio3:-DXV_NT_STAGGER=1" "sleep:-DXV_NT_STAGGER=2" "pipe:-DXV_NT_PIPE=1" "pipeprio:-DXV_NT_PIPE=1 -DXV_NT_STAGGER=1" "base2:" 2>&1 | grep -E "variant|fwd|dgrad" | tee $O/variants_stagger.log
