#!/bin/bash
cd $GRAFT_REPO_ROOT
for q in 4 8 2; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q python3 tools/e2e_ab.py --rounds 2 2>&1 | grep -v amdgpu.ids; done
