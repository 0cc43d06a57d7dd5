#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
python tools/shape_sweep.py --precision f32 > $O/shapes_f32.json 2> $O/shapes_f32.err; python -c "
import json; d=json.load(open('$O/shapes_f32.json'))
for k,v in d['shapes'].items(): print('f32',k,v['chunks_per_s'],v['algorithmic_tflops'])"
python tools/shape_sweep.py --precision f16x3 > $O/shapes_f16x3.json 2> $O/shapes_f16x3.err; python -c "
import json; d=json.load(open('$O/shapes_f16x3.json'))
for k,v in d['shapes'].items(): print('f16x3',k,v['chunks_per_s'],v['algorithmic_tflops'])"
python bench.py --extended --frames 400 --no-cpu-baseline > $O/bench_extended.json 2> $O/bench_extended.err; python tools/bench_summary.py $O/bench_extended.json
