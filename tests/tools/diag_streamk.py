"""Diagnosis of the evenly scheduled NT GEMM (run on the GPU box): xv_affine_forward against NumPy on shapes that share tiles, with the
error broken down by output tile and matched against partial K sums (a lost / stale / doubled share shows as a missing or extra K range).
usage: python tests/tools/diag_streamk.py            (XV_NT_SCHED=dp|sk forces the schedule)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_kaldi_speaker_amd import ops

CASES = [(3, 40, 30, 5, 512), (5, 61, 512, 5, 512), (4, 50, 512, 7, 512), (1, 333, 512, 1, 1500), (7, 16, 64, 7, 96), (130, 1, 3000, 1, 512),
         (16, 204, 512, 5, 512)]
reps = int(os.environ.get("XV_DIAG_REPS", "3"))
for segs, t_in, c, k, o in CASES:
    rs = np.random.RandomState(segs * 1000 + t_in)
    x = rs.randn(segs, t_in, c).astype(np.float32)
    kern = (rs.randn(k, c, o) / np.sqrt(k * c)).astype(np.float32)
    bias = rs.randn(o).astype(np.float32)
    c_pad = (c + 3) // 4 * 4
    xp = ops.pad_channels(torch.from_numpy(x.reshape(-1, c)).cuda(), c_pad).view(segs, t_in, c_pad)
    wt = ops.prep_weight_fwd(torch.from_numpy(kern).cuda(), c_pad)
    t_out = t_in - k + 1
    # reference per K-step (16 columns of the spliced row, in the generic kernel's order k = j * c_pad + channel)
    xs = np.zeros((segs, t_in, c_pad)); xs[:, :, :c] = x
    rows = np.stack([xs[b, t:t + k].reshape(-1) for b in range(segs) for t in range(t_out)])         # [M][k*c_pad]
    wk = np.zeros((k, c_pad, o)); wk[:, :c] = kern
    wk = wk.reshape(k * c_pad, o)
    K = k * c_pad
    nk = -(-K // 16)
    ref = rows @ wk + bias
    for rep in range(reps):
        z = ops.affine_forward(xp, k, wt, torch.from_numpy(bias).cuda(), o)
        torch.cuda.synchronize()
        got = z.cpu().numpy().astype(np.float64)
        err = np.abs(got - ref)
        bad = err > 1e-3 * np.abs(ref).max()
        print("case %s rep %d: M=%d N=%d K=%d nk=%d  max err %.3e  bad %d / %d" % ((segs, t_in, c, k, o), rep, ref.shape[0], o, K, nk, err.max(), bad.sum(), bad.size))
        if bad.any():
            M = ref.shape[0]
            for tm in range(-(-M // 128)):
                for tn in range(-(-o // 128)):
                    blk = bad[tm * 128:(tm + 1) * 128, tn * 128:(tn + 1) * 128]
                    if not blk.any():
                        continue
                    r0, c0 = np.argwhere(blk)[0]
                    i, j = tm * 128 + r0, tn * 128 + c0
                    # which contiguous K-step range [a, b) reproduces got[i, j] - bias?
                    pref = np.concatenate([[0.0], np.cumsum([rows[i, s * 16:(s + 1) * 16] @ wk[s * 16:(s + 1) * 16, j] for s in range(nk)])])
                    target = got[i, j] - bias[j]
                    best = min(((abs(pref[b] - pref[a] - target), a, b) for a in range(nk + 1) for b in range(a, nk + 1)), key=lambda t: t[0])
                    print("   tile (%d,%d): %d bad; e.g. [%d,%d] got %.5f ref %.5f; closest K-step range [%d,%d) of %d (residual %.2e); no-bias match %.2e"
                          % (tm, tn, blk.sum(), i, j, got[i, j], ref[i, j], best[1], best[2], nk, best[0], abs(got[i, j] - (ref[i, j] - bias[j]))))
            break
