cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_tn2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops_f16x3.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
for shape in "128 200" "64 300"; do set -- $shape
  XV_PROBE_OPS=w tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so $1 $2 10 > $O/probe_$1x$2.txt 2>&1
done
tools/ab_env.sh $O/ab.txt 2 "|--chunks 64 --frames 200:400" "new:XV_B=0" "new_s3b:XV_GEMM_SLOTS=3b" "tree:build_variants/r04_tree" > /dev/null
cat $O/tests.txt $O/probe_128x200.txt $O/probe_64x300.txt $O/ab.txt
