#!/bin/bash
# round 6, GPU call 1: the scheduled update - parity first, then a same-box A/B of the two new switches, then a timeline of the default
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_run1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_scheduled_update.py tests/test_gpu_c_abi.py -x -q > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -5 $O/pytest_new.log
tools/ab_env.sh $O/ab_eager.txt 3 "|--chunks 64 --frames 200:400" "base:XV_EAGER_UPDATE=0 XV_SEG_WGRAD=0" "eager:XV_EAGER_UPDATE=1 XV_SEG_WGRAD=0" "eager_seg1:XV_EAGER_UPDATE=1 XV_SEG_WGRAD=1" "eager_seg2:XV_EAGER_UPDATE=1 XV_SEG_WGRAD=2" "seg1:XV_EAGER_UPDATE=0 XV_SEG_WGRAD=1" > /dev/null
cat $O/ab_eager.txt
tools/step_timeline.sh $O/tl_default
tools/step_timeline.sh $O/tl_base XV_EAGER_UPDATE=0 XV_SEG_WGRAD=0
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "all tests rc=$?"; tail -5 $O/pytest_all.log
