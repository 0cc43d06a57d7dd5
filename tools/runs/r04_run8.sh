#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run8; mkdir -p $O
b() { # tag, dir, env..., -- args
  tag=$1; dir=$2; shift 2
  (cd $dir && env "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])")
}
for i in 1 2 3; do
  (cd build_variants/r03_tree && python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S1 r03', d['ms_per_step'], d['value'])")
  python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S1 r04', d['ms_per_step'], d['value'])"
done 2>&1 | tee $O/ab_s1.txt
for i in 1 2 3; do
  (cd build_variants/r03_tree && python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S3 r03', d['ms_per_step'], d['value'])")
  python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S3 r04', d['ms_per_step'], d['value'])"
  XV_LIB=$GRAFT_REPO_ROOT/build_variants/sk168/libxvector_hip.so python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S3 r04-sk168', d['ms_per_step'], d['value'])"
done 2>&1 | tee $O/ab_s3.txt
for i in 1 2; do
  (cd build_variants/r03_tree && python3 bench.py --steps 30 --warmup 5 --single-mode --no-cpu-baseline --extended --frames 400 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S5 r03', d['ms_per_step'], d['value'])")
  python3 bench.py --steps 30 --warmup 5 --single-mode --no-cpu-baseline --extended --frames 400 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S5 r04', d['ms_per_step'], d['value'])"
done 2>&1 | tee $O/ab_s5.txt
tools/step_timeline.sh $O/tl
tools/step_timeline.sh $O/tl_s3 -- --chunks 64 --frames 300
