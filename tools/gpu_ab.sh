#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -s -k "S5" > $O/fs.log 2>&1; echo "rc=$?"; grep "S5\|passed\|failed\|Error" $O/fs.log | tail; free -g | head -2
