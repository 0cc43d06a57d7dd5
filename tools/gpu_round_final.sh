#!/bin/bash
# One gpurun call that regenerates every measured artefact of a round into gpurun_out/profiles_<tag>/ (copy into profiles/ what is to
# be judged):  gpurun --timeout 3000 -- 'bash tools/gpu_round_final.sh r02'
tag=${1:-r02}
cd $GRAFT_REPO_ROOT
tools/profile_round.sh $tag 2>&1 | tail -12
O=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/ew -- python3 $GRAFT_REPO_ROOT/tools/elementwise_bench.py > $O/ew.log 2>&1)
python3 tools/elementwise_summary.py $(find $O/ew -name "*kernel_trace.csv" | head -1) > $O/elementwise.json; rm -rf $O/ew $O/ew.log
python3 tools/pool_bench.py 2>&1 | grep -v amdgpu.ids > $O/pool_bench.txt
python3 tools/segment_bench.py 2>&1 | grep -v amdgpu.ids > $O/segment_bench.txt
python3 tools/e2e_ab.py --rounds 2 2>&1 | grep -v amdgpu.ids > $O/e2e_ab.txt
python3 tools/extract_bench.py 2>&1 | grep -v amdgpu.ids > $O/extract_bench.txt; python3 tools/extract_driver_bench.py 2>&1 | grep -v amdgpu.ids >> $O/extract_bench.txt; tail -4 $O/extract_bench.txt
python3 bench.py --extended --frames 400 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_extended.json
python3 tools/bench_summary.py $O/bench_extended.json | grep -E "^f32|^f16x3"
python3 tools/shape_sweep.py --precision f32 2>/dev/null > $O/shapes_f32.json
python3 tools/shape_sweep.py --precision f16x3 2>/dev/null > $O/shapes_f16x3.json
python3 tools/loader_scale.py --batches 400 --need 22800 2>/dev/null | tail -1 > $O/loader_scale.json
ls -la $O
