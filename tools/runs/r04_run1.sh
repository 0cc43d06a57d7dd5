#!/bin/bash
# round 4, GPU call 1: full suite, A/B of the late side hand-over, timeline, batched extraction benches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run1; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
ab() { # name env...
  n=$1; shift
  for i in 1 2 3; do
    for v in 0 1; do
      env XV_SEG_SIDE_LATE=$v "$@" python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n late=$v', d['ms_per_step'], d['value'])"
    done
  done
}
ab S1 > $O/ab_s1.txt 2>&1; cat $O/ab_s1.txt
for v in 0 1; do
  env XV_SEG_SIDE_LATE=$v python3 bench.py --steps 60 --warmup 10 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S3 late=$v', d['ms_per_step'], d['value'])"
done > $O/ab_s3.txt 2>&1; cat $O/ab_s3.txt
tools/step_timeline.sh $O/tl_late1 XV_SEG_SIDE_LATE=1
tools/step_timeline.sh $O/tl_late0 XV_SEG_SIDE_LATE=0
timeout 600 python3 tools/extract_bench.py 2>&1 | grep -v amdgpu.ids > $O/extract_bench.txt; cat $O/extract_bench.txt
timeout 600 python3 tools/extract_driver_bench.py 2>&1 | grep -v amdgpu.ids > $O/extract_driver.txt; cat $O/extract_driver.txt
