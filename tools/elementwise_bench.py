"""The HBM-bound kernels of the step at the S1 tensor sizes, each alone (not a test): run under
`rocprofv3 --kernel-trace --stats` and feed the stats CSV to tools/elementwise_summary.py, which prices every kernel's
algorithmic bytes against the 8 TB/s HBM peak (north_star: >= 70 % on the HBM-bound kernels).

  python tools/elementwise_bench.py [frames=200]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_kaldi_speaker_amd import ops

B = 128
T = (int(sys.argv[1]) if len(sys.argv) > 1 else 200) - 14
rs = np.random.RandomState(0)


def rnd(*s):
    return torch.from_numpy(rs.randn(*s).astype(np.float32)).cuda()


def run(fn, n=20):
    for _ in range(n):
        fn()
    torch.cuda.synchronize()


for n in (512, 1500):
    rows = B * T
    z, da = rnd(rows, n), rnd(rows, n)
    gamma, beta = rnd(n).abs() + 0.5, rnd(n)
    mm, mv = torch.zeros(n).cuda(), torch.ones(n).cuda()
    part = ops.col_stats(z)
    mean, invstd, scale, shift, zmin, zmax, amax = ops.bn_finalize(part, rows, gamma, beta, 1e-3, 0.99, 0, mm, mv, with_range=True)
    # yardstick: what a plain streaming pass (torch's vectorised element-wise kernel, out = z + 1) reaches at this footprint on this device -
    # 97.5 MB lives in the 256 MB memory-side cache between repetitions, 285.7 MB does not
    tmp = torch.empty_like(z)
    run(lambda: torch.add(z, 1.0, out=tmp))
    run(lambda: ops.col_stats(z))
    run(lambda: ops.bn_finalize(part, rows, gamma, beta, 1e-3, 0.99, 0, mm, mv))
    run(lambda: ops.bn_apply(z, scale, shift, True))
    run(lambda: ops.bn_apply_split(z, scale, shift, True, amax))
    if n == 512:
        run(lambda: ops.bn_relu_backward(da, z, B, T, gamma, mean, invstd, scale, shift, True, 4))
        run(lambda: ops.bn_relu_backward(da, z, B * T, 1, gamma, mean, invstd, scale, shift, True, 0))      # a dense layer: no padding rows (strip form)
        run(lambda: ops.bn_relu_backward_split(da, z, B, T, gamma, mean, invstd, scale, shift, zmin, zmax, True, 4))
    else:
        pool = ops.stat_pool_forward_bn(z, B, T, scale, shift, True)
        dpool = rnd(B, 2 * n)
        run(lambda: ops.stat_pool_forward_bn(z, B, T, scale, shift, True))
        run(lambda: ops.bn_relu_backward_pooled(pool, dpool, B, T, z, gamma, mean, invstd, scale, shift, True))
        run(lambda: ops.bn_relu_backward_pooled_split(pool, dpool, B, T, z, gamma, mean, invstd, scale, shift, zmin, zmax, True))
        # prelu (a per-channel slope + its gradient): the reduction pass has no closed form - bn_bwd_reduce_pooled_kernel<true, true, false>
        slope, dalpha = rnd(n).abs() * 0.2 + 0.01, torch.zeros(n).cuda()
        with ops.activation(slope, dalpha):
            run(lambda: ops.bn_relu_backward_pooled(pool, dpool, B, T, z, gamma, mean, invstd, scale, shift, True))
p, g = rnd(9_830_000), rnd(9_830_000)
run(lambda: ops.sgd_update(p, g, 0.01))
print("done")
