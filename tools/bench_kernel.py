#!/usr/bin/env python3
"""Kernel micro-benchmarks of libxvector_hip.so at the S1 tensor sizes - diagnostics, not tests, not the product path.  One file, one
sub-command per question (they used to be eight scripts):

  python tools/bench_kernel.py gemm [frames=200]                fp32-input MFMA GEMMs per layer: forward / data gradient / weight gradient (incl. slab sum)
  python tools/bench_kernel.py gemm16 [frames=200] [tdnn2,tdnn5] the same for the split-precision (f16x3) kernels
  python tools/bench_kernel.py elementwise [frames=200]         every HBM-bound kernel of the step alone - run under `rocprofv3 --kernel-trace` and feed the
                                                                trace to tools/elementwise_summary.py (prices them against 8 TB/s; north_star asks >= 70 %)
  python tools/bench_kernel.py width                            the row-structured HBM-bound kernels against the channel count (512 ... 3 000)
  python tools/bench_kernel.py pitch                            ... against the row pitch of a 1 500-channel tensor
  python tools/bench_kernel.py pool [frames=186] [channels=1500] statistics pooling (+ BatchNorm) and its backward, warm (Infinity Cache) and cold
  python tools/bench_kernel.py segment                          the segment-level GEMMs: one-launch form (xv_skinny.hip) against GEMM + slab-sum launches
  python tools/bench_kernel.py staged                           plain step vs the staged (multi-GPU) backward, without / with a one-rank RCCL all-reduce

Environment: XV_DATA_SCALE=0 (all-zero operands: DVFS check), XV_B (chunks, gemm16), ITERS (segment), XV_LIB (another build of the library).
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

rs = np.random.RandomState(0)
SCALE = float(os.environ.get("XV_DATA_SCALE", "1"))


def rnd(*s):
    return torch.from_numpy((rs.randn(*s) * SCALE).astype(np.float32)).cuda()


def timeit(fn, n=20, warm=3):
    """mean microseconds of n back-to-back calls (events on the current stream)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def bn_vectors(ops, z, rows, n):
    gamma, beta = rnd(n).abs() + 0.5, rnd(n)
    mm, mv = torch.zeros(n).cuda(), torch.ones(n).cuda()
    part = ops.col_stats(z)
    return (gamma, beta, mm, mv, part) + tuple(ops.bn_finalize(part, rows, gamma, beta, 1e-3, 0.99, 0, mm, mv, with_range=True))


def cmd_gemm(argv):
    from tf_kaldi_speaker_amd import ops
    B, T = 128, int(argv[0]) if argv else 200
    for name, t_in, c, k, o in (("tdnn2", T - 4, 512, 5, 512), ("tdnn3", T - 8, 512, 7, 512), ("tdnn4", T - 14, 512, 1, 512), ("tdnn5", T - 14, 512, 1, 1500)):
        segs = B if k > 1 else B * t_in
        tin = t_in if k > 1 else 1
        tout = tin - k + 1
        x, kern, bias = rnd(segs, tin, c), rnd(k, c, o) * 0.05, rnd(o)
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


def cmd_gemm16(argv):
    from tf_kaldi_speaker_amd import ops
    B = int(os.environ.get("XV_B", "128"))
    T = int(argv[0]) if argv else 200
    only = argv[1].split(",") if len(argv) > 1 else None
    tot = 0.0
    for name, t_in, c, k, o in (("tdnn1", T, 32, 5, 512), ("tdnn2", T - 4, 512, 5, 512), ("tdnn3", T - 8, 512, 7, 512), ("tdnn4", T - 14, 512, 1, 512),
                                ("tdnn5", T - 14, 512, 1, 1500)):
        if only and name not in only:
            continue
        segs = B if k > 1 else B * t_in
        tin = t_in if k > 1 else 1
        tout = tin - k + 1
        x, kern, bias = rnd(segs * tin, c), rnd(k, c, o) * 0.05, rnd(o)
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


def cmd_elementwise(argv):
    """No timing of its own: 20 launches of every kernel, for a rocprofv3 kernel trace."""
    from tf_kaldi_speaker_amd import ops
    B, T = 128, (int(argv[0]) if argv else 200) - 14

    def run(fn, n=20):
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    for n in (512, 1500):
        rows = B * T
        z, da = rnd(rows, n), rnd(rows, n)
        gamma, beta, mm, mv, part, mean, invstd, scale, shift, zmin, zmax, amax = bn_vectors(ops, z, rows, n)
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
            pool, wpos, _ = ops.stat_pool_forward_bn_aux(z, B, T, scale, shift, True)
            dpool = rnd(B, 2 * n)
            run(lambda: ops.stat_pool_forward_bn(z, B, T, scale, shift, True))
            run(lambda: ops.bn_relu_backward_pooled(pool, dpool, B, T, z, gamma, mean, invstd, scale, shift, True))
            run(lambda: ops.bn_relu_backward_pooled_aux(pool, dpool, wpos, B, T, z, gamma, mean, invstd, scale, shift, True))      # the step's form: closed-form statistics + apply
            run(lambda: ops.bn_relu_backward_pooled_split(pool, dpool, B, T, z, gamma, mean, invstd, scale, shift, zmin, zmax, True))
            # prelu (a per-channel slope + its gradient): the reduction pass has no closed form - bn_bwd_reduce_pooled_kernel<true, true, false>
            slope, dalpha = rnd(n).abs() * 0.2 + 0.01, torch.zeros(n).cuda()
            with ops.activation(slope, dalpha):
                run(lambda: ops.bn_relu_backward_pooled(pool, dpool, B, T, z, gamma, mean, invstd, scale, shift, True))
    p, g = rnd(9_830_000), rnd(9_830_000)
    run(lambda: ops.sgd_update(p, g, 0.01))
    print("done")


def cmd_width(argv):
    from tf_kaldi_speaker_amd import ops
    B, T = 128, 186
    rows = B * T
    for n in (512, 1024, 1500, 1504, 1536, 2048, 3000):
        z, da = rnd(rows, n), rnd(rows, n)
        gamma, beta, mm, mv, part, mean, invstd, scale, shift, zmin, zmax, amax = bn_vectors(ops, z, rows, n)
        tmp = torch.empty_like(z)
        t = rows * n * 4 / 1e6
        r = {}
        r["torch_add"] = (timeit(lambda: torch.add(z, 1.0, out=tmp), 30, 5), 2 * t)
        r["bn_apply"] = (timeit(lambda: ops.bn_apply(z, scale, shift, True), 30, 5), 2 * t)
        r["col_stats"] = (timeit(lambda: ops.col_stats(z), 30, 5), t)
        r["bwd_dense"] = (timeit(lambda: ops.bn_relu_backward(da, z, rows, 1, gamma, mean, invstd, scale, shift, True, 0), 30, 5), 5 * t)   # reduce (2t) + apply (3t)
        pool = ops.stat_pool_forward_bn(z, B, T, scale, shift, True)
        dpool = rnd(B, 2 * n)
        r["pool_fwd"] = (timeit(lambda: ops.stat_pool_forward_bn(z, B, T, scale, shift, True), 30, 5), t)
        r["bwd_pooled"] = (timeit(lambda: ops.bn_relu_backward_pooled(pool, dpool, B, T, z, gamma, mean, invstd, scale, shift, True), 30, 5), 2 * t)
        print(n, "  ".join("%s %.1f us %.2f TB/s" % (k, us, mb / us) for k, (us, mb) in r.items()), flush=True)


def cmd_pitch(argv):
    from tf_kaldi_speaker_amd import _lib
    from tf_kaldi_speaker_amd.ops import _s, _p
    B, T, n = 128, 186, 1500
    rows = B * T
    scale, shift = rnd(n), rnd(n)
    t = rows * n * 4 / 1e6
    for ld in (1500, 1504, 1536, 1600, 2048):
        zf = rnd(rows, ld)
        af = torch.empty_like(zf)
        part = torch.empty(4 * ((rows + 127) // 128) * n, device="cuda")
        us_a = timeit(lambda: _lib.call("xv_bn_apply", _s(), _p(zf), rows, n, ld, _p(scale), _p(shift), 1, _p(af), ld), 30, 5)
        us_b = timeit(lambda: _lib.call("xv_bn_apply", _s(), _p(zf), rows, n, ld, _p(scale), _p(shift), 1, _p(af), n), 30, 5)
        us_c = timeit(lambda: _lib.call("xv_col_stats", _s(), _p(zf), rows, n, ld, _p(part)), 30, 5)
        print("ld %d: bn_apply (out pitch ld) %.1f us %.2f TB/s | (out dense) %.1f us %.2f TB/s | col_stats %.1f us %.2f TB/s"
              % (ld, us_a, 2 * t / us_a, us_b, 2 * t / us_b, us_c, t / us_c), flush=True)


def cmd_pool(argv):
    from tf_kaldi_speaker_amd import ops
    B = 128
    T = int(argv[0]) if argv else 186
    n = int(argv[1]) if len(argv) > 1 else 1500
    z = rnd(B * T, n)
    gamma, beta, mm, mv, part, mean, invstd, scale, shift, zmin, zmax, amax = bn_vectors(ops, z, B * T, n)
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
                               ("stat_pool_forward_bn_aux (+ wpos: the training step's form)", lambda: ops.stat_pool_forward_bn_aux(z, B, T, scale, shift, True), mb),
                               ("bn_relu_backward_pooled (reduce+finalize+apply)", lambda: ops.bn_relu_backward_pooled(pool, dpool, B, T, z, gamma, mean, invstd, scale, shift, True), 3 * mb),
                               ("bn_relu_backward_pooled_aux (closed-form statistics + apply)", lambda: ops.bn_relu_backward_pooled_aux(pool, dpool, wpos, B, T, z, gamma, mean, invstd, scale, shift, True), 2 * mb),
                               ("bn_relu_backward_pooled_split", lambda: ops.bn_relu_backward_pooled_split(pool, dpool, B, T, z, gamma, mean, invstd, scale, shift, zmin, zmax, True), 3 * mb)):
        for cold in (False, True):
            us = timed(fn, cold)
            print("%-62s %s  %7.1f us  %6.1f MB  %.2f TB/s" % (name, "cold" if cold else "warm", us, bytes_mb, bytes_mb / us))


def cmd_segment(argv):
    from tf_kaldi_speaker_amd import ops
    iters = int(os.environ.get("ITERS", "200"))
    for name, m, n, k in (("tdnn6 fwd", 128, 512, 3000), ("tdnn7 fwd", 128, 512, 512), ("logits", 128, 7351, 512), ("d out", 128, 512, 7352),
                          ("d tdnn7", 128, 512, 512), ("d pool", 128, 3000, 512)):
        x = rnd(m, k)
        wt = rnd(n, k) / float(np.sqrt(k))
        bias = torch.zeros(n).cuda()
        t_sk = timeit(lambda: ops.segment_gemm(x, wt, bias), iters, 10)
        x3 = x.view(m, 1, k)
        t_nt = timeit(lambda: ops.affine_forward(x3, 1, wt, bias, n), iters, 10)
        print("%-10s M=%d N=%5d K=%5d  segment_gemm %6.1f us   affine_forward (2 launches) %6.1f us" % (name, m, n, k, t_sk, t_nt))


def cmd_staged(argv):
    from tf_kaldi_speaker_amd import engine as E
    from tf_kaldi_speaker_amd.parallel import GradAllReduce
    B, T, D, N = 128, 200, 30, 7351
    eng = E.Engine(E.make_config(D, N, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, max_batch=B, max_frames=T), device="cuda:0")
    eng.init_variables(seed=0)
    x = rnd(B, T, D)
    y = torch.from_numpy(rs.randint(0, N, B).astype(np.int32)).cuda()

    def run(ar, n=40):
        for i in range(5):
            eng.train_step(x, y, 0.01, i, allreduce=ar)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            eng.train_step(x, y, 0.01, i, allreduce=ar)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    for _ in range(2):
        print("plain  %.4f ms" % run(None))
        print("staged %.4f ms (4 stages, no collective)" % run(GradAllReduce(None, 1)))
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    for _ in range(2):
        print("staged + one-rank RCCL all-reduce on the comm stream %.4f ms" % run(GradAllReduce(dist, 1, always=True)))
    dist.destroy_process_group()


COMMANDS = {"gemm": cmd_gemm, "gemm16": cmd_gemm16, "elementwise": cmd_elementwise, "width": cmd_width, "pitch": cmd_pitch, "pool": cmd_pool,
            "segment": cmd_segment, "staged": cmd_staged}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in COMMANDS:
        sys.exit(__doc__)
    COMMANDS[sys.argv[1]](sys.argv[2:])
