#!/bin/bash
# run 23: where the K = 512 forward GEMMs spend their time (per-workgroup stamps)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_run23; mkdir -p $O
for l in tdnn4 tdnn5 tdnn1 tdnn2; do echo "===== $l"; tools/gemm_probe build_variants/diag/libxvector_hip.so 128 200 5 - $l | grep -A22 "^stamps" | head -24; done > $O/stamps.txt 2>&1
cat $O/stamps.txt
