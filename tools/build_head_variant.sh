#!/bin/bash
# Build the library of a COMMIT (default HEAD) into build_variants/<name>/libxvector_hip.so for an XV_LIB same-box A/B against the working
# tree (same ABI version required):  tools/build_head_variant.sh head [commit]
name=${1:-head}; commit=${2:-HEAD}
R=$(cd $(dirname $0)/.. && pwd)
tmp=$(mktemp -d /tmp/xvhead.XXXX)
git -C $R archive $commit tf_kaldi_speaker_amd/csrc include tests/c_abi | tar -x -C $tmp
make -C $tmp/tf_kaldi_speaker_amd/csrc -j6 $tmp/tf_kaldi_speaker_amd/libxvector_hip.so 2>&1 | grep -E "error|warning"
mkdir -p $R/build_variants/$name && cp $tmp/tf_kaldi_speaker_amd/libxvector_hip.so $R/build_variants/$name/ && echo "built $name from $(git -C $R rev-parse --short $commit)"
rm -rf $tmp
