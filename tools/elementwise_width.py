import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from tf_kaldi_speaker_amd import ops
B, T = 128, 186
rows = B * T
rs = np.random.RandomState(0)
def rnd(*s): return torch.from_numpy(rs.randn(*s).astype(np.float32)).cuda()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for n in (512, 1024, 1500, 1504, 1536, 2048, 3000):
    z, da = rnd(rows, n), rnd(rows, n)
    gamma, beta = rnd(n).abs() + 0.5, rnd(n)
    mm, mv = torch.zeros(n).cuda(), torch.ones(n).cuda()
    part = ops.col_stats(z)
    mean, invstd, scale, shift, zmin, zmax, amax = ops.bn_finalize(part, rows, gamma, beta, 1e-3, 0.99, 0, mm, mv, with_range=True)
    tmp = torch.empty_like(z)
    t = rows * n * 4 / 1e6
    r = {}
    r["torch_add"] = (timeit(lambda: torch.add(z, 1.0, out=tmp)), 2 * t)
    r["bn_apply"] = (timeit(lambda: ops.bn_apply(z, scale, shift, True)), 2 * t)
    r["col_stats"] = (timeit(lambda: ops.col_stats(z)), t)
    r["bwd_dense"] = (timeit(lambda: ops.bn_relu_backward(da, z, rows, 1, gamma, mean, invstd, scale, shift, True, 0)), 5 * t)   # reduce (2t) + apply (3t)
    pool = ops.stat_pool_forward_bn(z, B, T, scale, shift, True)
    dpool = rnd(B, 2 * n)
    r["pool_fwd"] = (timeit(lambda: ops.stat_pool_forward_bn(z, B, T, scale, shift, True)), t)
    r["bwd_pooled"] = (timeit(lambda: ops.bn_relu_backward_pooled(pool, dpool, B, T, z, gamma, mean, invstd, scale, shift, True)), 2 * t)
    print(n, "  ".join("%s %.1f us %.2f TB/s" % (k, us, mb / us) for k, (us, mb) in r.items()), flush=True)
