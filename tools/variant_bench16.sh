#!/bin/bash
# usage: tools/variant_bench16.sh "<extra hipcc -D flags>" ... ; builds each variant in turn and runs tools/gemm16_bench.py
cd $(dirname $0)/..
for flags in "$@"; do
  echo "=== variant: $flags"
  make -C tf_kaldi_speaker_amd/csrc -j8 -B FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$PWD/include -I. -Wall -Wno-unused-function $flags" 2>&1 | grep -E "error" 
  timeout 300 python tools/gemm16_bench.py 200 ${LAYERS:-tdnn2,tdnn5} 2>&1 | grep -E "tdnn|sum"
done
