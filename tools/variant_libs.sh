#!/bin/bash
# Build variants of one translation unit of libxvector_hip.so on the GPU box and run a micro-benchmark against each (same box, one after
# the other; rule 24 of the guide: never compare across boxes).
# usage: tools/variant_libs.sh <unit.hip> <bench.py [args]> -- "name1:-DFLAGS..." "name2:-DFLAGS..." ...
unit=$1; bench=$2; shift 2
[ "$1" == "--" ] && shift
R=${GRAFT_REPO_ROOT:-$PWD}
src=$R/tf_kaldi_speaker_amd/csrc
W=/tmp/xv_variants; mkdir -p $W/base
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$src -Wall -Wno-unused-function"
for f in xv_gemm xv_gemm16 xv_skinny xv_elementwise xv_loss xv_attention xv_engine; do
  [ -f $W/base/$f.o ] || hipcc $FL -c $src/$f.hip -o $W/base/$f.o &
done
wait
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  mkdir -p $W/$name
  hipcc $FL $flags -c $src/$unit -o $W/$name/unit.o 2>&1 | grep -E "error|warning: .*spill" 
  objs=""
  for f in xv_gemm xv_gemm16 xv_skinny xv_elementwise xv_loss xv_attention xv_engine; do
    if [ "$f.hip" == "$unit" ]; then objs="$objs $W/$name/unit.o"; else objs="$objs $W/base/$f.o"; fi
  done
  hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $W/$name/libxvector_hip.so
  echo "=== variant $name: $flags"
  XV_LIB=$W/$name/libxvector_hip.so timeout 300 python $R/tools/$bench 2>&1 | grep -E "tdnn|sum|Error|error"
done
