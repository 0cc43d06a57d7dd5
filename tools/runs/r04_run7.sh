#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run7; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "ops or engine or extract or c_abi" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
XV_PROBE_OPS=w tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so 128 200 10 > $O/probe.txt 2>&1; cat $O/probe.txt
for i in 1 2 3; do
  python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S1 wpc3', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('isolated_frac'))"
  XV_LIB=$GRAFT_REPO_ROOT/build_variants/tn4/libxvector_hip.so python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S1 wpc4', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('isolated_frac'))"
done
python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S3', d['ms_per_step'], d['value'])"
tools/step_timeline.sh $O/tl
timeout 600 python3 tools/extract_driver_bench.py 2>&1 | grep -v amdgpu.ids > $O/extract_driver.txt; cat $O/extract_driver.txt
