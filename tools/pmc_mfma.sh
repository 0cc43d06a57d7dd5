#!/bin/bash
# MFMA-utilisation evidence for the GEMM kernels: one rocprofv3 --pmc pass (+ --kernel-trace for the
# durations; no other trace domain) per {program} x {random, all-zero operands}.
# usage (GPU box, repo root): tools/pmc_mfma.sh <outdir>      -> <outdir>/pmc_mfma.json (+ raw CSVs removed)
out=${1:-gpurun_out/pmc_mfma}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$out
export TMPDIR=/tmp
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
for prog in gemm gemm16; do
  for scale in 1 0; do
    tag=${prog}_bench_x${scale}
    export XV_DATA_SCALE=$scale
    (cd /tmp && timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $R/$out/$tag -- python3 $R/tools/bench_kernel.py $prog > $R/$out/$tag.log 2>&1)
  done
done
unset XV_DATA_SCALE
python3 $R/tools/pmc_mfma_summary.py $R/$out > $R/$out/pmc_mfma.json
for d in $R/$out/*/; do rm -rf "$d"; done
