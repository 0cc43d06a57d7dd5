#!/bin/bash
# Build a variant of the WHOLE library (every translation unit with the extra -D flags) HERE into build_variants/<name>/libxvector_hip.so
# (git-ignored, travels with the gpurun snapshot; selected on the box with XV_LIB=/root/repo/build_variants/<name>/libxvector_hip.so):
#   tools/build_full_variant.sh prio0 -DXV_EW_PRIO=0
name=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
src=$R/tf_kaldi_speaker_amd/csrc
out=$R/build_variants/$name; mkdir -p $out/obj
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$src -Wall -Wno-unused-function"
for f in xv_gemm xv_gemm16 xv_skinny xv_elementwise xv_loss xv_attention xv_engine; do
  ( hipcc $FL "$@" -c $src/$f.hip -o $out/obj/$f.o 2>&1 | grep -E "error|spill" ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 0.5; done
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $out/obj/*.o -o $out/libxvector_hip.so && rm -rf $out/obj && echo "built $name: $@"
