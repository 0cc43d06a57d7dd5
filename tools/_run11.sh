cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_att; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_pooling.py tests/test_gpu_ops.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
L=/root/repo/build_variants
tools/ab_env.sh $O/ab.txt 2 "--attention" "head:XV_LIB=$L/head/libxvector_hip.so" "new:XV_B=0" > /dev/null
tools/step_timeline.sh $O/s4 -- --attention > /dev/null 2>&1
cat $O/tests.txt $O/ab.txt; grep -n "att_score\|att_pool_dw" $O/s4.timeline.txt
