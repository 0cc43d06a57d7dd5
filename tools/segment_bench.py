#!/usr/bin/env python3
"""Segment-level GEMMs alone: xv_segment_gemm (one launch, xv_skinny.hip) against xv_affine_forward (GEMM + slab-sum launches) on the
chain's shapes at S1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tf_kaldi_speaker_amd import ops

SHAPES = [("tdnn6 fwd", 128, 512, 3000), ("tdnn7 fwd", 128, 512, 512), ("logits", 128, 7351, 512), ("d out", 128, 512, 7352),
          ("d tdnn7", 128, 512, 512), ("d pool", 128, 3000, 512)]
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
iters = int(os.environ.get("ITERS", "200"))


def timed(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, m, n, k in SHAPES:
    x = torch.from_numpy(rs.randn(m, k).astype(np.float32)).to(dev)
    wt = torch.from_numpy((rs.randn(n, k) / np.sqrt(k)).astype(np.float32)).to(dev)
    bias = torch.zeros(n, device=dev)
    t_sk = timed(lambda: ops.segment_gemm(x, wt, bias))
    x3 = x.view(m, 1, k)
    t_nt = timed(lambda: ops.affine_forward(x3, 1, wt, bias, n))
    print("%-10s M=%d N=%5d K=%5d  segment_gemm %6.1f us   affine_forward (2 launches) %6.1f us" % (name, m, n, k, t_sk, t_nt))
