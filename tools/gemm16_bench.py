"""Micro-benchmark of the split-precision (f16x3) GEMM kernels on the S1 layer shapes (not a test)."""
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


B = int(os.environ.get("XV_B", "128"))
rs = np.random.RandomState(0)
ZERO = float(os.environ.get('XV_DATA_SCALE', '1'))
def rnd(*s): return torch.from_numpy((rs.randn(*s) * ZERO).astype(np.float32)).cuda()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
layers = [("tdnn1", T, 32, 5, 512), ("tdnn2", T - 4, 512, 5, 512), ("tdnn3", T - 8, 512, 7, 512), ("tdnn4", T - 14, 512, 1, 512),
          ("tdnn5", T - 14, 512, 1, 1500)]
tot = 0.0
for name, t_in, c, k, o in layers:
    if only and name not in only:
        continue
    segs = B if k > 1 else B * t_in
    tin = t_in if k > 1 else 1
    tout = tin - k + 1
    x = rnd(segs * tin, c); kern = rnd(k, c, o) * 0.05; bias = rnd(o)
    xp = ops.split_planes(x)
    wtp = ops.split_planes(ops.prep_weight_fwd(kern, c))
    o_ld = (o + 7) // 8 * 8
    wf = ops.prep_weight_dgrad(kern) if k > 1 else kern.view(c, o)
    if o_ld != o:
        wf = torch.nn.functional.pad(wf.view(c, k, o), (0, o_ld - o)).reshape(c, k * o_ld).contiguous()
    wfp = ops.split_planes(wf)
    dzp = ops.split_planes(rnd(segs * (tout + 2 * (k - 1)), o))
    fl = 2.0 * segs * tout * k * c * o
    us = timeit(lambda: ops.affine_forward_f16x3(xp, segs, tin, k, wtp, bias, o, with_stats=True)); tot += us
    print("%s fwd   M=%6d K=%5d N=%5d  %8.1f us  %6.1f TF" % (name, segs * tout, k * c, o, us, fl / us / 1e6))
    if name != "tdnn1":
        fl2 = 2.0 * segs * (tout + k - 1) * k * o * c
        us = timeit(lambda: ops.affine_dgrad_f16x3(dzp, segs, tout, k, wfp, c)); tot += us
        print("%s dgrad M=%6d K=%5d N=%5d  %8.1f us  %6.1f TF" % (name, segs * (tout + k - 1), k * o_ld, c, us, fl2 / us / 1e6))
    us = timeit(lambda: ops.affine_wgrad_f16x3(xp, segs, tin, k, c, dzp, tout + 2 * (k - 1), k - 1, o, kern, 1e-2)); tot += us
    print("%s wgrad M=%6d N=%5d R=%6d  %8.1f us  %6.1f TF (incl. reduce)" % (name, k * c, o, segs * tout, us, fl / us / 1e6))
print("sum %.1f us" % tot)
