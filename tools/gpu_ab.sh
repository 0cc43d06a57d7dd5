#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "pool" 2>&1 | tail -2
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --single-mode --precision f32 > $O/tr.log 2>&1)
f=$(find $O/tr -name "*kernel_stats.csv" | head -1)
python3 tools/kernel_stats.py $f pooled_stats bn_bwd_apply stat_pool
rm -rf $O/tr
