#!/bin/bash
# run 27: "hy" schedule of the NT GEMM (whole tiles + shares of the remainder): isolated GEMMs, parity, step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_run27; mkdir -p $O
for shape in "128 200" "64 300" "128 400"; do set -- $shape
  echo "=== $1 x $2 base (auto)"; tools/gemm_probe build_variants/r04_base/libxvector_hip.so $1 $2 5 | grep -E "fwd|dgrad|^sum"
  echo "=== $1 x $2 new (auto)"; tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so $1 $2 5 | grep -E "fwd|dgrad|^sum"
  echo "=== $1 x $2 new (XV_NT_SCHED=hy)"; XV_NT_SCHED=hy tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so $1 $2 5 | grep -E "fwd|dgrad|^sum"
done > $O/probe.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
one() { t=$1; shift; "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  for shape in "S1|" "S3|--chunks 64 --frames 200:400" "S2|--frames 400"; do
    name=${shape%%|*}; args=${shape#*|}
    XV_LIB=$GRAFT_REPO_ROOT/build_variants/r04_base/libxvector_hip.so one "$name base" python3 bench.py --steps 50 --warmup 12 --single-mode --no-cpu-baseline $args
    one "$name new" python3 bench.py --steps 50 --warmup 12 --single-mode --no-cpu-baseline $args
  done
done > $O/ab.txt 2>&1
cat $O/probe.txt $O/tests.txt $O/ab.txt
