#!/bin/bash
# One gpurun call that regenerates every measured artefact of a round into gpurun_out/profiles_<tag>/ (copy into profiles/ what is to
# be judged):  gpurun --timeout 3000 -- 'bash tools/gpu_round_final.sh r02'
tag=${1:-r02}
cd $GRAFT_REPO_ROOT
tools/profile_round.sh $tag 2>&1 | tail -12
O=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/ew -- python3 $GRAFT_REPO_ROOT/tools/bench_kernel.py elementwise > $O/ew.log 2>&1)
python3 tools/elementwise_summary.py $(find $O/ew -name "*kernel_trace.csv" | head -1) > $O/elementwise.json; rm -rf $O/ew $O/ew.log
timeout 900 python3 tools/bench_kernel.py pool 2>&1 | grep -v amdgpu.ids > $O/pool_bench.txt
timeout 900 python3 tools/bench_kernel.py segment 2>&1 | grep -v amdgpu.ids > $O/segment_bench.txt
timeout 900 python3 tools/bench_host.py e2e_ab --rounds 2 2>&1 | grep -v amdgpu.ids > $O/e2e_ab.txt
timeout 900 python3 tools/bench_host.py extract 2>&1 | grep -v amdgpu.ids > $O/extract_bench.txt; timeout 900 python3 tools/bench_host.py extract_driver 2>&1 | grep -v amdgpu.ids >> $O/extract_bench.txt; tail -4 $O/extract_bench.txt
timeout 900 python3 bench.py --extended --frames 400 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_extended.json
python3 tools/bench_summary.py $O/bench_extended.json | grep -E "^f32|^f16x3"
timeout 900 python3 tools/shape_sweep.py --precision f32 2>/dev/null > $O/shapes_f32.json
timeout 900 python3 tools/shape_sweep.py --precision f16x3 2>/dev/null > $O/shapes_f16x3.json
timeout 900 python3 tools/bench_host.py loader_scale --batches 400 --need 25800 2>/dev/null | tail -1 > $O/loader_scale.json
timeout 900 python3 tools/bench_host.py trainer 400 2>&1 | grep -v amdgpu.ids | tail -3 > $O/trainer_bench.txt
tools/step_timeline.sh $O/s3_64x300 -- --chunks 64 --frames 300; rm -f $O/s3_64x300.log $O/s3_64x300.json
tools/step_timeline.sh $O/s4 -- --attention; rm -f $O/s4.log $O/s4.json
tools/step_timeline.sh $O/s5 -- --extended --frames 400; rm -f $O/s5.log $O/s5.json
timeout 900 tools/pmc_mfma.sh gpurun_out/profiles_$tag/pmc_mfma_run > /dev/null 2>&1; cp $O/pmc_mfma_run/pmc_mfma.json $O/pmc_mfma.json; rm -rf $O/pmc_mfma_run
for shape in "128 200" "64 300"; do set -- $shape
  for sched in dp sk auto; do
    echo "=== XV_NT_SCHED=$sched"; if [ $sched == auto ]; then timeout 300 tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so $1 $2 5; else XV_NT_SCHED=$sched timeout 300 tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so $1 $2 5; fi
  done > $O/gemm_probe_schedules_$1x$2.txt 2>&1
done
XV_PROBE_EXTRA=1 XV_PROBE_ONLY="att_key0 att_key1 ext_k3" timeout 300 tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so 128 200 5 > $O/gemm_probe_attention_extended.txt 2>&1
XV_PROBE_EXTRA=1 XV_PROBE_ONLY="ext_k3 tdnn5" timeout 300 tools/gemm_probe tf_kaldi_speaker_amd/libxvector_hip.so 128 400 5 >> $O/gemm_probe_attention_extended.txt 2>&1
# the clock inside the forward GEMM of a full S1 step (diagnostics build: tools/variant.sh unit xv_gemm.hip "diag:-DXV_DIAG=1")
if [ -f build_variants/diag/libxvector_hip.so ]; then
  XV_LIB=$GRAFT_REPO_ROOT/build_variants/diag/libxvector_hip.so XV_DIAG_M=24576 XV_DIAG_N=512 XV_DIAG_K=2560 timeout 300 python3 tools/step_clock.py 2>&1 | grep -v amdgpu.ids > $O/step_clock.txt
fi
# same box, alternated: the previous round's tree (build_variants/<prev>_tree: git archive of its last commit, built) against this one
prev=$(ls -d build_variants/r*_tree 2>/dev/null | tail -1)
if [ -n "$prev" ]; then
  one() { t=$1; shift; "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['value'])"; }
  for i in 1 2 3; do
    for shape in "S1|" "S3|--chunks 64 --frames 200:400" "S2|--frames 400"; do
      name=${shape%%|*}; args=${shape#*|}
      (cd $prev && one "$name $(basename $prev)" timeout 300 python3 bench.py --steps 40 --warmup 10 --single-mode --no-cpu-baseline $args)
      one "$name $tag" timeout 300 python3 bench.py --steps 40 --warmup 10 --single-mode --no-cpu-baseline $args
    done
  done > $O/ab_same_box.txt 2>&1
fi
timeout 600 python3 tests/tools/determinism.py 2>&1 | grep -v amdgpu.ids > $O/determinism.txt
ls -la $O
