#!/bin/bash
# run 20: hand-over events on kernel completion signals, single join wait, affine pooled BN backward: tests + same-box A/B vs the last commit
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_run20; mkdir -p $O
timeout 60 tools/sync_cost_probe 40 10 256 256 2>&1 | tail -2 > $O/handover_check.txt
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_ops.py tests/test_gpu_pooling.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
one() { t=$1; shift; "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['value'])"; }
for i in 1 2 3 4; do
  XV_LIB=$GRAFT_REPO_ROOT/build_variants/r04_base/libxvector_hip.so one "S1 base" python3 bench.py --steps 60 --warmup 15 --single-mode --no-cpu-baseline
  one "S1 new" python3 bench.py --steps 60 --warmup 15 --single-mode --no-cpu-baseline
done > $O/ab_s1.txt 2>&1
for i in 1 2 3; do
  XV_LIB=$GRAFT_REPO_ROOT/build_variants/r04_base/libxvector_hip.so one "S3 base" python3 bench.py --steps 60 --warmup 15 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400
  one "S3 new" python3 bench.py --steps 60 --warmup 15 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400
done > $O/ab_s3.txt 2>&1
cat $O/handover_check.txt $O/tests.txt $O/ab_s1.txt $O/ab_s3.txt
