#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "pool" 2>&1 | tail -15
