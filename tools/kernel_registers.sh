#!/bin/bash
# VGPRs / spills / scratch / LDS of every kernel of a translation unit as hipcc reports them in the code object metadata (no GPU needed):
#   tools/kernel_registers.sh xv_gemm.hip [-DFLAGS...] > profiles/rNN_kernel_registers.txt
unit=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
src=$R/tf_kaldi_speaker_amd/csrc
tmp=$(mktemp /tmp/xvreg.XXXX.s)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$src "$@" -S --cuda-device-only $src/$unit -o $tmp 2>/dev/null || exit 1
echo "# $unit $* : hipcc -S --offload-arch=gfx950, amdhsa.kernels metadata"
printf "%-64s %6s %7s %8s %7s\n" kernel vgprs spilled scratchB ldsB
awk '/^amdhsa.kernels:/{on=1} on && /\.group_segment_fixed_size:/{lds=$2} on && /\.name:/{name=$2} on && /\.private_segment_fixed_size:/{scr=$2} on && /\.vgpr_count:/{v=$2} on && /\.vgpr_spill_count:/{printf "%-64s %6s %7s %8s %7s\n", name, v, $2, scr, lds}' $tmp | while read n v s c l; do printf "%-64s %6s %7s %8s %7s\n" "$(echo $n | c++filt | cut -c1-64)" $v $s $c $l; done
rm -f $tmp
