cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_tnrows; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
L=/root/repo/build_variants
for v in head new; do lib=tf_kaldi_speaker_amd/libxvector_hip.so; [ $v == head ] && lib=build_variants/head/libxvector_hip.so
  echo "=== $v"; XV_PROBE_OPS=w tools/gemm_probe $lib 128 200 10; done > $O/probe_128x200.txt 2>&1
tools/ab_env.sh $O/ab.txt 3 "|--chunks 64 --frames 200:400" "head:XV_LIB=$L/head/libxvector_hip.so" "new:XV_B=0" > /dev/null
cat $O/tests.txt $O/probe_128x200.txt $O/ab.txt
