#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run12; mkdir -p $O
for v in late1 late2; do XV_LIB=$GRAFT_REPO_ROOT/build_variants/$v/libxvector_hip.so timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "train_step_matches_oracle and not extended and not shipped" 2>&1 | tail -1; done
one() { tag=$1; shift; "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  (cd build_variants/r03_tree && one "S1 r03" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline)
  one "S1 r04" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline
  for v in late1 late2; do XV_LIB=$GRAFT_REPO_ROOT/build_variants/$v/libxvector_hip.so one "S1 r04-$v" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline; done
done 2>&1 | tee $O/ab_s1.txt
for i in 1 2 3; do
  (cd build_variants/r03_tree && one "S3 r03" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400)
  one "S3 r04" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400
  for v in late1 late2 sk168; do XV_LIB=$GRAFT_REPO_ROOT/build_variants/$v/libxvector_hip.so one "S3 r04-$v" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400; done
done 2>&1 | tee $O/ab_s3.txt
XV_LIB=$GRAFT_REPO_ROOT/build_variants/late1/libxvector_hip.so tools/step_timeline.sh $O/tl_late1
XV_LIB=$GRAFT_REPO_ROOT/build_variants/late2/libxvector_hip.so tools/step_timeline.sh $O/tl_late2
timeout 600 python3 tools/extract_driver_bench.py 2>&1 | grep -v amdgpu.ids > $O/extract_driver.txt; cat $O/extract_driver.txt
