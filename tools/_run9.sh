cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_tl; mkdir -p $O
tools/step_timeline.sh $O/s1 -- > /dev/null 2>&1
cat $O/s1.timeline.txt; cat $O/s1.gaps.txt
