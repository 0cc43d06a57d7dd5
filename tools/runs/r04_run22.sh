#!/bin/bash
# run 22: tdnn1's weight gradient with fewer, longer splits (512 / 256 / 128 workgroups instead of 1 024)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_run22; mkdir -p $O
one() { t=$1; shift; "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['value'])"; }
for v in few512 few256 few128; do echo "== $v"; tools/gemm_probe build_variants/$v/libxvector_hip.so 128 200 5 | grep "tdnn1 wgrad"; done > $O/probe.txt 2>&1
echo "== r04"; tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so 128 200 5 | grep "tdnn1 wgrad" >> $O/probe.txt
for i in 1 2 3; do
  one "S1 r04" python3 bench.py --steps 60 --warmup 15 --single-mode --no-cpu-baseline
  for v in few512 few256 few128; do XV_LIB=$GRAFT_REPO_ROOT/build_variants/$v/libxvector_hip.so one "S1 $v" python3 bench.py --steps 60 --warmup 15 --single-mode --no-cpu-baseline; done
done > $O/ab_s1.txt 2>&1
for i in 1 2; do
  one "S3 r04" python3 bench.py --steps 60 --warmup 15 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400
  for v in few512 few256 few128; do XV_LIB=$GRAFT_REPO_ROOT/build_variants/$v/libxvector_hip.so one "S3 $v" python3 bench.py --steps 60 --warmup 15 --single-mode --no-cpu-baseline --chunks 64 --frames 200:400; done
done > $O/ab_s3.txt 2>&1
cat $O/probe.txt $O/ab_s1.txt $O/ab_s3.txt
