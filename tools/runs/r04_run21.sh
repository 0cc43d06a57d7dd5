#!/bin/bash
# run 21: timelines of the base and the new tree (are the hand-over gaps gone?)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_run21; mkdir -p $O
tools/step_timeline.sh $O/tl_new
tools/step_timeline.sh $O/tl_base XV_LIB=$GRAFT_REPO_ROOT/build_variants/r04_base/libxvector_hip.so
tail -3 $O/tl_new.timeline.txt; tail -3 $O/tl_base.timeline.txt
