#!/bin/bash
# run 26: per-workgroup stamps of the evenly scheduled forward GEMMs at the shipped batch shape (64 x 300)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_run26; mkdir -p $O
for l in tdnn2 tdnn3; do echo "===== $l (64 x 300, auto = even schedule)"; tools/gemm_probe build_variants/diag/libxvector_hip.so 64 300 5 - $l | grep -A20 "^stamps" | head -22; done > $O/stamps.txt 2>&1
echo "===== tdnn2 (64 x 300, dp)" >> $O/stamps.txt; XV_NT_SCHED=dp tools/gemm_probe build_variants/diag/libxvector_hip.so 64 300 5 - tdnn2 | grep -A20 "^stamps" | head -22 >> $O/stamps.txt 2>&1
cat $O/stamps.txt
