#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run10; mkdir -p $O
one() { tag=$1; shift; "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for i in 1 2; do
  (cd build_variants/r03_tree && one "S1 r03" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline)
  one "S1 r04-768" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline
  for v in tn640 tn896 tn960 tn1008; do XV_LIB=$GRAFT_REPO_ROOT/build_variants/$v/libxvector_hip.so one "S1 r04-$v" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline; done
done 2>&1 | tee $O/ab_s1.txt
for i in 1 2; do
  (cd build_variants/r03_tree && one "S3 r03" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400)
  one "S3 r04-768" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400
  for v in tn640 tn896 tn960 tn1008; do XV_LIB=$GRAFT_REPO_ROOT/build_variants/$v/libxvector_hip.so one "S3 r04-$v" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400; done
done 2>&1 | tee $O/ab_s3.txt
