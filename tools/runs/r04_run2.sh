#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_extract_batched.py tests/test_gpu_trainer.py -x -q -m gpu -k "extract or drivers or roundtrip" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python3 tools/extract_driver_bench.py 2>&1 | grep -v amdgpu.ids > $O/extract_driver.txt; cat $O/extract_driver.txt
for w in 1024 896 768 640 512; do echo "=== XV_TN_WGS=$w"; XV_TN_WGS=$w XV_PROBE_OPS=w tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so 128 200 10; done > $O/tn_wgs.txt 2>&1; cat $O/tn_wgs.txt
for a in 48 64 96 1000; do echo "=== XV_TN_AHEAD_MIN=$a"; XV_TN_AHEAD_MIN=$a XV_PROBE_OPS=w tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so 128 200 10; done > $O/tn_ahead.txt 2>&1; cat $O/tn_ahead.txt
