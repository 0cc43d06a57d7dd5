#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run14; mkdir -p $O
python3 - <<'PY'
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import extract_driver_bench as B
tmp = "/tmp/xv_prof"; os.makedirs(tmp, exist_ok=True)
model = os.path.join(tmp, "exp")
if not os.path.isdir(model): B.make_model(model)
B.make_ark(os.path.join(tmp, "in.ark"), 5000, 5000)
print("ark ready")
PY
PKG=$GRAFT_REPO_ROOT/tf_kaldi_speaker_amd
cd $PKG
for i in 1 2 3; do
  cat /tmp/xv_prof/in.ark > /dev/null
  TF_KALDI_ROOT=$PKG PYTHONPATH=$PKG python3 nnet/lib/extract.py --node tdnn6_dense /tmp/xv_prof/exp ark:/tmp/xv_prof/in.ark ark:/tmp/xv_prof/out.ark 2>&1 | grep Extracted
done
TF_KALDI_ROOT=$PKG PYTHONPATH=$PKG python3 -m cProfile -o /tmp/xv_prof/prof.out nnet/lib/extract.py --node tdnn6_dense /tmp/xv_prof/exp ark:/tmp/xv_prof/in.ark ark:/tmp/xv_prof/out.ark 2>&1 | grep Extracted
python3 -c "
import pstats
p = pstats.Stats('/tmp/xv_prof/prof.out'); p.sort_stats('tottime').print_stats(28)
" | tail -45 > $GRAFT_REPO_ROOT/$O/profile.txt
cat $GRAFT_REPO_ROOT/$O/profile.txt
nproc; python3 -c "
import time
t=time.perf_counter(); s=0
for i in range(3000000): s+=i
print('py loop 3M adds: %.3f s' % (time.perf_counter()-t))"
