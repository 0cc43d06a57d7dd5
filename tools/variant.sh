#!/bin/bash
# Build alternative libxvector_hip.so builds HERE (hipcc cross-compiles gfx950 without a GPU) into build_variants/<name>/ (git-ignored, travels
# with the gpurun snapshot); on the box they are selected with XV_LIB=$GRAFT_REPO_ROOT/build_variants/<name>/libxvector_hip.so
# (tools/ab.sh, tools/probe.sh, any bench).  Same ABI version as the working tree required.
#   tools/variant.sh unit xv_gemm.hip "base:" "diag:-DXV_DIAG=1" ...   one translation unit rebuilt with -D flags, the others from csrc/build
#   tools/variant.sh full prio0 -DXV_EW_PRIO=0                          every translation unit with the flags
#   tools/variant.sh commit head [HEAD~3]                               the library of a commit (default HEAD)
# (tools/gemm_probe is rebuilt when its source is newer.)
mode=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
src=$R/tf_kaldi_speaker_amd/csrc
UNITS="xv_gemm xv_gemm16 xv_skinny xv_elementwise xv_loss xv_attention xv_engine"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$src -Wall -Wno-unused-function"
throttle() { while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 0.5; done; }      # at most 4 compilers at a time (8 CPUs, 64 GB)
case $mode in
unit)
  unit=$1; shift
  make -C $src -j8 >/dev/null || exit 1
  [ -x $R/tools/gemm_probe ] && [ $R/tools/gemm_probe -nt $R/tools/gemm_probe.cpp ] || hipcc -O2 -std=c++17 $R/tools/gemm_probe.cpp -o $R/tools/gemm_probe -ldl || exit 1
  for v in "$@"; do
    name=${v%%:*}; flags=${v#*:}
    mkdir -p $R/build_variants/$name
    ( hipcc $FL $flags -c $src/$unit -o $R/build_variants/$name/unit.o 2>&1 | grep -E "error|spill"
      objs=""
      for f in $UNITS; do
        if [ "$f.hip" == "$unit" ]; then objs="$objs $R/build_variants/$name/unit.o"; else objs="$objs $src/build/$f.o"; fi
      done
      hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/build_variants/$name/libxvector_hip.so && rm -f $R/build_variants/$name/unit.o && echo "built $name: $flags" ) &
    throttle
  done
  wait ;;
full)
  name=$1; shift
  out=$R/build_variants/$name; mkdir -p $out/obj
  for f in $UNITS; do
    ( hipcc $FL "$@" -c $src/$f.hip -o $out/obj/$f.o 2>&1 | grep -E "error|spill" ) &
    throttle
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC $out/obj/*.o -o $out/libxvector_hip.so && rm -rf $out/obj && echo "built $name: $@" ;;
commit)
  name=${1:-head}; commit=${2:-HEAD}
  tmp=$(mktemp -d /tmp/xvhead.XXXX)
  git -C $R archive $commit tf_kaldi_speaker_amd/csrc include tests/c_abi | tar -x -C $tmp
  make -C $tmp/tf_kaldi_speaker_amd/csrc -j6 $tmp/tf_kaldi_speaker_amd/libxvector_hip.so 2>&1 | grep -E "error|warning"
  mkdir -p $R/build_variants/$name && cp $tmp/tf_kaldi_speaker_amd/libxvector_hip.so $R/build_variants/$name/ && echo "built $name from $(git -C $R rev-parse --short $commit)"
  rm -rf $tmp ;;
*) echo "usage: tools/variant.sh unit <unit.hip> \"name:-DFLAGS\" ... | full <name> -DFLAGS ... | commit <name> [commit]"; exit 1 ;;
esac
