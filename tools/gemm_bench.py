"""Micro-benchmark of the three MFMA GEMM kernels on the S1 layer shapes (not a test)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_kaldi_speaker_amd import ops

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us

B = 128
rs = np.random.RandomState(0)
ZERO = float(os.environ.get('XV_DATA_SCALE', '1'))      # 0: all-zero operands (DVFS check, MI355X_MICROARCH.md give-back item 1)
def rnd(*s): return torch.from_numpy((rs.randn(*s) * ZERO).astype(np.float32)).cuda()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
layers = [("tdnn2", T - 4, 512, 5, 512), ("tdnn3", T - 8, 512, 7, 512), ("tdnn4", T - 14, 512, 1, 512), ("tdnn5", T - 14, 512, 1, 1500)]
for name, t_in, c, k, o in layers:
    t_out = t_in - k + 1
    segs = B if k > 1 else B * t_in
    tin = t_in if k > 1 else 1
    tout = tin - k + 1
    x = rnd(segs, tin, c); kern = rnd(k, c, o) * 0.05; bias = rnd(o)
    wt = ops.prep_weight_fwd(kern, c)
    wf = ops.prep_weight_dgrad(kern) if k > 1 else kern.view(c, o)
    dzp = rnd(segs * (tout + 2 * (k - 1)), o)
    fl = 2.0 * segs * tout * k * c * o
    us = timeit(lambda: ops.affine_forward(x, k, wt, bias, o, with_stats=True))
    print("%s fwd   M=%6d K=%5d N=%5d  %8.1f us  %6.1f TF" % (name, segs * tout, k * c, o, us, fl / us / 1e6))
    fl2 = 2.0 * segs * (tout + k - 1) * k * o * c
    us = timeit(lambda: ops.affine_dgrad(dzp, segs, tout, o, k, wf, c))
    print("%s dgrad M=%6d K=%5d N=%5d  %8.1f us  %6.1f TF" % (name, segs * (tout + k - 1), k * o, c, us, fl2 / us / 1e6))
    us = timeit(lambda: ops.affine_wgrad(x, k, c, dzp, tout + 2 * (k - 1), k - 1, o, kern, 1e-2))
    print("%s wgrad M=%6d N=%5d R=%6d  %8.1f us  %6.1f TF (incl. reduce)" % (name, k * c, o, segs * tout, us, fl / us / 1e6))
