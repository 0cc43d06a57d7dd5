#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run11; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
one() { tag=$1; shift; "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  (cd build_variants/r03_tree && one "S1 r03" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline)
  one "S1 r04" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline
done 2>&1 | tee $O/ab_s1.txt
for i in 1 2 3; do
  (cd build_variants/r03_tree && one "S3 r03" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400)
  one "S3 r04" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400
  XV_LIB=$GRAFT_REPO_ROOT/build_variants/skld8/libxvector_hip.so one "S3 r04-ld8" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400
done 2>&1 | tee $O/ab_s3.txt
for i in 1 2; do
  (cd build_variants/r03_tree && one "S2 r03" python3 bench.py --steps 30 --warmup 5 --single-mode --no-cpu-baseline --frames 400)
  one "S2 r04" python3 bench.py --steps 30 --warmup 5 --single-mode --no-cpu-baseline --frames 400
  (cd build_variants/r03_tree && one "S4 r03" python3 bench.py --steps 30 --warmup 5 --single-mode --no-cpu-baseline --attention)
  one "S4 r04" python3 bench.py --steps 30 --warmup 5 --single-mode --no-cpu-baseline --attention
  (cd build_variants/r03_tree && one "S5 r03" python3 bench.py --steps 30 --warmup 5 --single-mode --no-cpu-baseline --extended --frames 400)
  one "S5 r04" python3 bench.py --steps 30 --warmup 5 --single-mode --no-cpu-baseline --extended --frames 400
done 2>&1 | tee $O/ab_s245.txt
timeout 600 python3 tools/extract_bench.py 2>&1 | grep -v amdgpu.ids > $O/extract_bench.txt; cat $O/extract_bench.txt
timeout 600 python3 tools/extract_driver_bench.py 2>&1 | grep -v amdgpu.ids > $O/extract_driver.txt; cat $O/extract_driver.txt
