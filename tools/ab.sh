#!/bin/bash
# Same-box, alternated in-step A/B of run-time switches (and, optionally, of another tree) on the GPU box:
#   tools/ab.sh <out.txt> <rounds> "<shape args>|<shape args>..." "name1:ENV=a ENV2=b" "name2:" "tree:build_variants/r04_tree" ...
# Every (shape, variant) pair runs `bench.py --single-mode --no-cpu-baseline` once per round; one line per run: shape variant ms/step chunks/s.
out=$1; rounds=$2; shapes=$3; shift 3
R=${GRAFT_REPO_ROOT:-$PWD}
# stderr is kept: the library names an XV_* switch it does not know once ("ignoring unknown environment switch") and ignores it - a mistyped
# A/B switch would then silently benchmark the default build, so such a run FAILS here instead of printing a number
ERR=$(mktemp)
one() { "$@" 2>$ERR | tail -1 > $ERR.out; if grep -q "ignoring unknown environment switch" $ERR; then echo "REFUSED: $(grep -m1 'ignoring unknown environment switch' $ERR)"; return 1; fi
        cat $ERR.out | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print(d['ms_per_step'], d['value'], 'dominant', r.get('avg_launch_ms'), r.get('frac'))"; }
IFS='|' read -ra SH <<< "$shapes"
for i in $(seq 1 $rounds); do
  for sh in "${SH[@]}"; do
    for v in "$@"; do
      name=${v%%:*}; spec=${v#*:}
      if [ "$name" == "tree" ] && [ -n "$spec" ]; then
        echo -n "[$sh] $(basename $spec) "; (cd $R/$spec && one timeout 300 python3 bench.py --steps 40 --warmup 10 --single-mode --no-cpu-baseline $sh)
      else
        echo -n "[$sh] $name "; (cd $R && one timeout 300 env $spec python3 bench.py --steps 40 --warmup 10 --single-mode --no-cpu-baseline $sh)
      fi
    done
  done
done 2>&1 | tee $out
