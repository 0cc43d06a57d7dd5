"""Prototype check + timing of the split-precision NT GEMM (experimental xvx_* entry points)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_kaldi_speaker_amd import _lib, ops
lib = C.CDLL(_lib.LIB_PATH)
VP = C.c_void_p
lib.xvx_split_planes.argtypes = [VP, VP, C.c_int, C.c_int, C.c_int, VP, C.c_int, C.c_long, C.c_int, C.c_float]
lib.xvx_gemm16_nt.argtypes = [VP, VP, C.c_long, C.c_long, C.c_int, C.c_int, VP, C.c_long, C.c_long, VP, C.c_long, C.c_int, C.c_int, C.c_int, VP, C.c_int, C.c_float]
lib.xv_last_error.restype = C.c_char_p
def P(t): return VP(t.data_ptr()) if t is not None else VP(0)
def S(): return VP(torch.cuda.current_stream().cuda_stream)
def chk(rc): assert rc == 0, lib.xv_last_error()

def planes(x2d, mode, scale=1.0):
    rows, c = x2d.shape
    ld = (c + 7) // 8 * 8
    out = torch.zeros((mode, rows, ld), dtype=torch.int16, device="cuda")
    chk(lib.xvx_split_planes(S(), P(x2d), rows, c, c, P(out), ld, rows * ld, mode, scale))
    return out, ld

def pow2_scale(x, target=2.0 ** 13):
    m = float(x.abs().max())
    return float(2.0 ** np.floor(np.log2(target / m))) if m > 0 else 1.0

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def run(segs, t_in, c, k, o, mode, check=True):
    rs = np.random.RandomState(0)
    x = torch.from_numpy(np.maximum(rs.randn(segs, t_in, c), 0).astype(np.float32) * 1.3).cuda()   # ReLU-like
    kern = torch.from_numpy((rs.randn(k, c, o) / np.sqrt(k * c)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rs.randn(o).astype(np.float32)).cuda()
    wt = ops.prep_weight_fwd(kern, c)                      # [o][k*c] fp32
    t_out = t_in - k + 1
    M, K = segs * t_out, k * c
    sa = pow2_scale(x) if mode == 2 else 1.0
    sb = pow2_scale(wt) if mode == 2 else 1.0
    xa, lda = planes(x.view(-1, c), mode, sa)
    wb, ldb = planes(wt, mode, sb)
    out = torch.empty((M, o), dtype=torch.float32, device="cuda")
    def f():
        chk(lib.xvx_gemm16_nt(S(), P(xa), lda, xa.shape[1] * lda, t_out, t_in, P(wb), ldb, wb.shape[1] * ldb, P(out), o, M, o, K,
                              P(bias), mode, 1.0 / (sa * sb)))
    f(); torch.cuda.synchronize()
    res = {}
    if check:
        xn = x.cpu().numpy().astype(np.float64); kn = kern.cpu().numpy().astype(np.float64)
        cols = np.stack([xn[:, j:j + t_out, :] for j in range(k)], axis=2).reshape(M, K)
        ref = cols @ kn.reshape(K, o) + bias.cpu().numpy()
        got = out.cpu().numpy().astype(np.float64)
        z32 = ops.affine_forward(x, k, wt, bias, o).cpu().numpy().astype(np.float64)
        res["err16"] = np.abs(got - ref).max() / np.abs(ref).max()
        res["err32"] = np.abs(z32 - ref).max() / np.abs(ref).max()
    us = timeit(f)
    us32 = timeit(lambda: ops.affine_forward(x, k, wt, bias, o))
    fl = 2.0 * M * K * o
    print("mode %d  M=%d K=%d N=%d: split %.1f us (%.1f TF-equiv)   fp32 MFMA %.1f us (%.1f TF)   %s" % (
        mode, M, K, o, us, fl / us / 1e6, us32, fl / us32 / 1e6, " ".join("%s=%.2e" % kv for kv in res.items())))

if __name__ == "__main__":
    for mode in (3, 2):
        run(5, 61, 512, 5, 512, mode)                 # small ragged check
        run(128, 196, 512, 5, 512, mode, check=False) # tdnn2
        run(128 * 186, 1, 512, 1, 1500, mode, check=False)  # tdnn5


def run_tn(segs, t_in, c, k, o, mode, splits, check=True):
    """wgrad: dK[(j,c)][o] = sum_{b,t} x[b][t+j][c] * dz[b][t][o]"""
    lib.xvx_gemm16_tn.argtypes = [VP, VP, C.c_long, C.c_long, C.c_int, VP, C.c_long, C.c_long, C.c_int, C.c_int, VP, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_float]
    rs = np.random.RandomState(1)
    t_out = t_in - k + 1
    x = torch.from_numpy(np.maximum(rs.randn(segs, t_in, c), 0).astype(np.float32)).cuda()
    dz = torch.from_numpy((rs.randn(segs, t_out, o) * 1e-3).astype(np.float32)).cuda()
    kern = torch.zeros((k, c, o), dtype=torch.float32, device="cuda")
    R, M = segs * t_out, k * c
    sa = pow2_scale(x) if mode == 2 else 1.0
    sb = pow2_scale(dz) if mode == 2 else 1.0
    xa, lda = planes(x.view(-1, c), mode, sa)
    zb, ldb = planes(dz.view(-1, o), mode, sb)
    P = torch.empty((splits, M, o), dtype=torch.float32, device="cuda")
    def f():
        chk(lib.xvx_gemm16_tn(S(), P(xa) if False else VP(xa.data_ptr()), lda, xa.shape[1] * lda, t_in, VP(zb.data_ptr()), ldb, zb.shape[1] * ldb, t_out, t_out,
                              VP(P.data_ptr()), M, o, R, splits, mode, 1.0 / (sa * sb)))
    f(); torch.cuda.synchronize()
    res = {}
    if check:
        xn = x.cpu().numpy().astype(np.float64); dn = dz.cpu().numpy().astype(np.float64)
        cols = np.stack([xn[:, j:j + t_out, :] for j in range(k)], axis=2).reshape(R, M)
        ref = cols.T @ dn.reshape(R, o)
        got = P.sum(0).cpu().numpy().astype(np.float64)
        g32 = ops.affine_wgrad(x, k, c, dz.view(-1, o), t_out, 0, o, kern, 0.0).cpu().numpy().astype(np.float64).reshape(M, o)
        res["err16"] = np.abs(got - ref).max() / np.abs(ref).max()
        res["err32"] = np.abs(g32 - ref).max() / np.abs(ref).max()
    us = timeit(f)
    us32 = timeit(lambda: ops.affine_wgrad(x, k, c, dz.view(-1, o), t_out, 0, o, kern, 0.0))
    fl = 2.0 * R * M * o
    print("TN mode %d  M=%d N=%d R=%d splits=%d: split %.1f us (%.1f TF-equiv, no reduce)   fp32 %.1f us (%.1f TF incl reduce)   %s" % (
        mode, M, o, R, splits, us, fl / us / 1e6, us32, fl / us32 / 1e6, " ".join("%s=%.2e" % kv for kv in res.items())))


if __name__ == "__main__":
    for mode in (2, 3):
        run_tn(5, 61, 512, 5, 512, mode, 1)
        run_tn(3, 47, 64, 7, 96, mode, 2)
    run_tn(128, 196, 512, 5, 512, 2, 6, check=False)
    run_tn(128, 196, 512, 5, 512, 2, 12, check=False)
    run_tn(128 * 186, 1, 512, 1, 1504, 2, 10, check=False)
