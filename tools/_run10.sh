cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_lh; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_trainer.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
L=/root/repo/build_variants
tools/ab_env.sh $O/ab.txt 3 "|--chunks 64 --frames 200:400" "head:XV_LIB=$L/head/libxvector_hip.so" "new:XV_B=0" > /dev/null
tools/step_timeline.sh $O/s4 -- --attention > /dev/null 2>&1
tools/step_timeline.sh $O/s5 -- --extended --frames 400 > /dev/null 2>&1
tools/step_timeline.sh $O/s1 -- > /dev/null 2>&1
cat $O/tests.txt $O/ab.txt
