#!/bin/bash
# Build -D variants of one translation unit of libxvector_hip.so HERE (hipcc cross-compiles gfx950 without a GPU) into
# build_variants/<name>/libxvector_hip.so (git-ignored, travels with the gpurun snapshot), so the GPU box only runs them:
#   tools/build_variants.sh xv_gemm.hip "base:" "diag:-DXV_DIAG=2" ...
# then on the box: tools/gemm_probe build_variants/<name>/libxvector_hip.so   (or XV_LIB=... python tools/gemm_bench.py)
unit=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
src=$R/tf_kaldi_speaker_amd/csrc
make -C $src -j8 >/dev/null || exit 1
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$src -Wall -Wno-unused-function"
[ -x $R/tools/gemm_probe ] && [ $R/tools/gemm_probe -nt $R/tools/gemm_probe.cpp ] || hipcc -O2 -std=c++17 $R/tools/gemm_probe.cpp -o $R/tools/gemm_probe -ldl || exit 1
pids=""
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  mkdir -p $R/build_variants/$name
  ( hipcc $FL $flags -c $src/$unit -o $R/build_variants/$name/unit.o 2>&1 | grep -E "error|spill" ;
    objs=""
    for f in xv_gemm xv_gemm16 xv_skinny xv_elementwise xv_loss xv_attention xv_engine; do
      if [ "$f.hip" == "$unit" ]; then objs="$objs $R/build_variants/$name/unit.o"; else objs="$objs $src/build/$f.o"; fi
    done
    hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/build_variants/$name/libxvector_hip.so && rm -f $R/build_variants/$name/unit.o && echo "built $name: $flags" ) &
  pids="$pids $!"
  # at most 4 compilers at a time (8 CPUs, 64 GB)
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 0.5; done
done
wait
