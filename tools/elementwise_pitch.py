import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from tf_kaldi_speaker_amd import ops, _lib
from tf_kaldi_speaker_amd.ops import _s, _p
B, T = 128, 186
rows = B * T
n = 1500
rs = np.random.RandomState(0)
def rnd(*s): return torch.from_numpy(rs.randn(*s).astype(np.float32)).cuda()
def timeit(fn, k=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3
gamma, beta = rnd(n).abs() + 0.5, rnd(n)
scale, shift = rnd(n), rnd(n)
t = rows * n * 4 / 1e6
for ld in (1500, 1504, 1536, 1600, 2048):
    zf = rnd(rows, ld); af = torch.empty_like(zf)
    part = torch.empty(4 * ((rows + 127) // 128) * n, device="cuda")
    us_a = timeit(lambda: _lib.call("xv_bn_apply", _s(), _p(zf), rows, n, ld, _p(scale), _p(shift), 1, _p(af), ld))
    us_b = timeit(lambda: _lib.call("xv_bn_apply", _s(), _p(zf), rows, n, ld, _p(scale), _p(shift), 1, _p(af), n))
    us_c = timeit(lambda: _lib.call("xv_col_stats", _s(), _p(zf), rows, n, ld, _p(part)))
    print("ld %d: bn_apply (out pitch ld) %.1f us %.2f TB/s | (out dense) %.1f us %.2f TB/s | col_stats %.1f us %.2f TB/s" % (ld, us_a, 2 * t / us_a, us_b, 2 * t / us_b, us_c, t / us_c), flush=True)
