"""The pooled kernels (statistics pooling fused with the last frame layer's BatchNorm; its backward) alone at the S1 size, timed
with events, cold (a 300 MB buffer is rewritten between calls) and warm.  python tools/pool_bench.py [frames=186] [channels=1500]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_kaldi_speaker_amd import ops

B = 128
T = int(sys.argv[1]) if len(sys.argv) > 1 else 186
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
rs = np.random.RandomState(0)
rnd = lambda *s: torch.from_numpy(rs.randn(*s).astype(np.float32)).cuda()
z = rnd(B * T, n)
gamma, beta = rnd(n).abs() + 0.5, rnd(n)
mm, mv = torch.zeros(n).cuda(), torch.ones(n).cuda()
part = ops.col_stats(z)
mean, invstd, scale, shift, zmin, zmax, amax = ops.bn_finalize(part, B * T, gamma, beta, 1e-3, 0.99, 0, mm, mv, with_range=True)
pool, wpos, _ = ops.stat_pool_forward_bn_aux(z, B, T, scale, shift, True)
dpool = rnd(B, 2 * n)
flush = torch.empty(300 << 18, dtype=torch.float32, device="cuda")      # 300 MB: evicts the Infinity Cache


def timed(fn, cold, iters=30):
    tot = 0.0
    for i in range(iters + 3):
        if cold:
            flush.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        if i >= 3:
            tot += a.elapsed_time(b)
    return tot / iters * 1e3


mb = B * T * n * 4 / 1e6
for name, fn, bytes_mb in (("amax (plain streaming read, for reference)", lambda: ops.amax_of(z), mb),
                           ("torch.sum (vendor streaming read)", lambda: torch.sum(z), mb),
                           ("stat_pool_forward_bn", lambda: ops.stat_pool_forward_bn(z, B, T, scale, shift, True), mb),
                           ("bn_relu_backward_pooled (reduce+finalize+apply)", lambda: ops.bn_relu_backward_pooled(pool, dpool, B, T, z, gamma, mean, invstd, scale, shift, True), 3 * mb),
                           ("bn_relu_backward_pooled_aux (closed-form statistics + apply)", lambda: ops.bn_relu_backward_pooled_aux(pool, dpool, wpos, B, T, z, gamma, mean, invstd, scale, shift, True), 2 * mb),
                           ("bn_relu_backward_pooled_split", lambda: ops.bn_relu_backward_pooled_split(pool, dpool, B, T, z, gamma, mean, invstd, scale, shift, zmin, zmax, True), 3 * mb)):
    for cold in (False, True):
        us = timed(fn, cold)
        print("%-62s %s  %7.1f us  %6.1f MB  %.2f TB/s" % (name, "cold" if cold else "warm", us, bytes_mb, bytes_mb / us))
