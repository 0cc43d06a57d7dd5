#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run4; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for w in 1 0; do echo "=== XV_TN_ORDER=$w"; XV_TN_ORDER=$w XV_PROBE_OPS=w tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so 128 200 10; done > $O/tn_order.txt 2>&1; cat $O/tn_order.txt
for i in 1 2 3; do for v in 1 0; do XV_TN_ORDER=$v python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S1 order=$v', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('isolated_frac'))"; done; done
python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S3', d['ms_per_step'], d['value'])"
tools/step_timeline.sh $O/tl
timeout 600 python3 tools/extract_driver_bench.py 2>&1 | grep -v amdgpu.ids > $O/extract_driver.txt; cat $O/extract_driver.txt
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/ew -- python3 $GRAFT_REPO_ROOT/tools/elementwise_bench.py > $GRAFT_REPO_ROOT/$O/ew.log 2>&1)
python3 tools/elementwise_summary.py $(find $O/ew -name "*kernel_trace.csv" | head -1) > $O/elementwise.json; rm -rf $O/ew; cat $O/elementwise.json | head -60
