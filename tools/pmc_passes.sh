#!/bin/bash
# usage: tools/pmc_passes.sh <outdir> "<counters pass 1>" "<counters pass 2>" ... -- <program args>   (one rocprofv3 --pmc run per pass)
out=$1; shift
passes=()
while [ "$1" != "--" ]; do passes+=("$1"); shift; done
shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for p in "${passes[@]}"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $p --output-format csv -d $R/$out/p$i -- "$@" > $R/$out.p$i.log 2>&1)
  f=$(find $R/$out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f | grep -A8 "${PMC_KERNEL:-gemm16_nt}" | head -${PMC_LINES:-7}
  rm -rf $R/$out/p$i
done
