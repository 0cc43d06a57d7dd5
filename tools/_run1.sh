cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_tn; mkdir -p $O
for f in 2 4; do
  echo "== tests XV_TN_FORM=$f"; XV_TN_FORM=$f timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad or affine or gemm" 2>&1 | tail -3
done > $O/tests.txt 2>&1
XV_TN_FORM=4 timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -3 >> $O/tests.txt
for shape in "128 200" "64 300"; do set -- $shape
  for f in 1 2 4; do echo "=== XV_TN_FORM=$f"; XV_PROBE_OPS=w XV_TN_FORM=$f tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so $1 $2 10; done > $O/probe_$1x$2.txt 2>&1
done
tools/ab_env.sh $O/ab.txt 2 "|--chunks 64 --frames 200:400" "f1:XV_TN_FORM=1" "f2:XV_TN_FORM=2" "f4:XV_TN_FORM=4" "tree:build_variants/r04_tree" > /dev/null
cat $O/tests.txt; cat $O/probe_128x200.txt; cat $O/ab.txt
