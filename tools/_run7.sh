cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_rows; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_streamk_schedule.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
L=/root/repo/build_variants
for v in norows new; do lib=tf_kaldi_speaker_amd/libxvector_hip.so; [ $v == norows ] && lib=build_variants/norows/libxvector_hip.so
  echo "=== $v"; XV_PROBE_OPS=fd tools/gemm_probe $lib 128 200 10; done > $O/probe_128x200.txt 2>&1
for v in norows new; do lib=tf_kaldi_speaker_amd/libxvector_hip.so; [ $v == norows ] && lib=build_variants/norows/libxvector_hip.so
  echo "=== $v"; XV_PROBE_OPS=fd tools/gemm_probe $lib 64 300 10; done > $O/probe_64x300.txt 2>&1
tools/ab_env.sh $O/ab.txt 3 "|--chunks 64 --frames 200:400" "norows:XV_LIB=$L/norows/libxvector_hip.so" "rows:XV_B=0" "tree:build_variants/r04_tree" > /dev/null
cat $O/tests.txt $O/probe_128x200.txt $O/probe_64x300.txt $O/ab.txt
