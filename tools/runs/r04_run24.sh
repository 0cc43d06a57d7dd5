#!/bin/bash
# run 24: the one-workgroup-per-tile NT kernel's epilogue at raised wave priority
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_run24; mkdir -p $O
( echo "== r04"; tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so 128 200 5 | grep -E "fwd|dgrad|^sum"
  echo "== epi3"; tools/gemm_probe build_variants/epi3/libxvector_hip.so 128 200 5 | grep -E "fwd|dgrad|^sum"
  echo "== epi3 stamps tdnn4"; tools/gemm_probe build_variants/epi3d/libxvector_hip.so 128 200 5 - tdnn4 | grep -A9 "^stamps" ) > $O/probe.txt 2>&1
one() { t=$1; shift; "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  one "S1 r04" python3 bench.py --steps 60 --warmup 15 --single-mode --no-cpu-baseline
  XV_LIB=$GRAFT_REPO_ROOT/build_variants/epi3/libxvector_hip.so one "S1 epi3" python3 bench.py --steps 60 --warmup 15 --single-mode --no-cpu-baseline
done > $O/ab_s1.txt 2>&1
cat $O/probe.txt $O/ab_s1.txt
