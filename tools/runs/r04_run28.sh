#!/bin/bash
# run 28: where the whole-tiles + shares schedule pays in the step: off (hy0) / long K only (r04 tree = mode 1) / short K only (hy2) / both (hy3)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_run28; mkdir -p $O
one() { t=$1; shift; "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  for shape in "S1|" "S3|--chunks 64 --frames 200:400" "S2|--frames 400"; do
    name=${shape%%|*}; args=${shape#*|}
    for v in hy0 hy2 hy3; do XV_LIB=$GRAFT_REPO_ROOT/build_variants/$v/libxvector_hip.so one "$name $v" python3 bench.py --steps 50 --warmup 12 --single-mode --no-cpu-baseline $args; done
    one "$name hy1" python3 bench.py --steps 50 --warmup 12 --single-mode --no-cpu-baseline $args
  done
done > $O/ab.txt 2>&1
sort -s -k1,2 $O/ab.txt
