#!/usr/bin/env python3
"""Headline benchmark: utterance-chunks/sec of full x-vector optimiser steps on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], SURVEY.md section 8d shape S1): standard 5-layer TDNN
x-vector + AM-Softmax (m = 0.2), 128 chunks x 200 frames x 30-dim MFCC per GPU, 7351 speakers,
fp32, plain SGD + L2 - one "step" = forward + loss + backward (+ gradient all-reduce for N > 1)
+ update on synthetic features already resident in HBM.  Weak scaling: 128 chunks per GPU.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` for the
dominant kernel (its launches are bracketed by hipEvents on their stream inside the timed region; an
untimed probe pass brackets every MFMA GEMM kind for the per-kernel table) and, at N = 1,
`cpu_baseline` = the NumPy oracle's train_step timed on this box's host cores (all cores, 16 and 1
BLAS threads; the fastest is `value`).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

# algorithmic work of shape S1 (per 128-chunk step)
B, T, D, NSPK = 128, 200, 30, 7351
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32-input MFMA peak
PEAK_HBM_GBS = 8000.0            # same guide: HBM3E peak (6.29 TB/s measured by a float4 copy)
PEAK_F16_MFMA_TFLOPS = 16 * 157.3   # same guide: fp32-input MFMA = 1/16 of the BF16/F16 rate (~2.5 PF dense)
NKINDS = 6
KIND_NAMES = ["xv_gemm_nt_kernel<true> (forward conv/dense + BN stats)",
              "xv_gemm_nt_kernel<false> (data gradients / logits)",
              "xv_gemm_tn_kernel (weight gradients)",
              "xv_gemm16_nt_kernel<true> (f16x3 forward conv/dense + BN stats)",
              "xv_gemm16_nt_kernel<false> (f16x3 data gradients)",
              "xv_gemm16_tn_kernel (f16x3 weight gradients)"]
KIND_SYMBOLS = ["xv_gemm_nt_kernel<true>", "xv_gemm_nt_kernel<false>", "xv_gemm_tn_kernel",
                "xv_gemm16_nt_kernel<true>", "xv_gemm16_nt_kernel<false>", "xv_gemm16_tn_kernel"]
# algorithmic-FLOP peak of each kind: an f16x3 product issues 3 fp16 MFMAs (hi*hi + hi*lo + lo*hi)
KIND_PEAK = [PEAK_F32_MFMA_TFLOPS] * 3 + [PEAK_F16_MFMA_TFLOPS / 3.0] * 3


def pmc_traffic(kind):
    """HBM-side bytes per launch of the dominant kernel, from the most recent committed rocprofv3 PMC
    passes (profiles/rNN_pmc_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate
    passes).  bench.py cannot run the PMC passes itself, so this is null when no file is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        for name, v in d["kernels"].items():
            if KIND_SYMBOLS[kind] in name:
                return int(v["bytes_per_launch_corrected"])
    except Exception:
        return None
    return None


def step_flops(b, t, d, n, attention=False):
    """2*M*K*N per contraction, backward = 2x forward (attention = shape S4: the key network
    512 -> 1500 -> 1500 of nnet_conf/*_tdnn4_att.json on the T-14 frames)."""
    t1, t2, t3 = t - 4, t - 8, t - 14
    fwd = 2.0 * b * (t1 * 5 * d * 512 + t2 * 2560 * 512 + t3 * 3584 * 512 + t3 * 512 * 512 + t3 * 512 * 1500)
    fwd += 2.0 * b * (3000 * 512 + 512 * 512 + 512 * n)
    if attention:
        fwd += 2.0 * b * t3 * (512 * 1500 + 1500 * 1500)
    return fwd, 3.0 * fwd


def step_bytes(b, t, d, n):
    """Compulsory HBM traffic of one step under perfect fusion: every frame-level activation
    written once + read once forward, read once more backward, its gradient written + read (5 S); input read twice;
    parameters read fwd + bwd, gradient written + read, parameter written (5 P).  fp32 bytes."""
    t1, t2, t3 = t - 4, t - 8, t - 14
    S = 4.0 * b * (t1 * 512 + t2 * 512 + t3 * 512 + t3 * 512 + t3 * 1500)
    P = 4.0 * (5 * d * 512 + 5 * 512 * 512 + 7 * 512 * 512 + 512 * 512 + 512 * 1500 + 3000 * 512 + 512 * 512 + 512 * n
               + 6 * 512 * 5 + 1500 * 5)
    return 5.0 * S + 2.0 * 4.0 * b * t * d + 5.0 * P


def cpu_baseline(seconds_budget=25.0):
    """The oracle (a NumPy port of the reference arithmetic, NOT TensorFlow) on the host cores.  The port is dominated by
    NumPy element-wise passes, and OpenBLAS on every core of a large host can be slower than on one, so three BLAS thread
    counts are timed (all cores, 16, 1 - SURVEY.md section 8d asks for the single-thread and the all-cores rows) and
    `value` is the fastest of them, with `cores` = the threads it used."""
    from oracle import xvector_oracle as O
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threadpool_limits = None
        threads = os.cpu_count() or 1
    cb = 16
    cfg = O.Config(feat_dim=D, num_speakers=NSPK, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True)
    V = O.init_variables(cfg, seed=0, dtype=np.float32)
    rs = np.random.RandomState(0)
    x = rs.randn(cb, T, D).astype(np.float32)
    y = rs.randint(0, NSPK, cb)
    state = {}
    V, state, _ = O.train_step(V, state, cfg, x, y, 0.01, 0)      # warm-up (BLAS thread pool, page faults)
    counts = [threads] if threadpool_limits is None else sorted({threads, min(threads, 16), 1}, reverse=True)
    rows, step = [], 1
    for nt in counts:
        import contextlib
        ctx = threadpool_limits(limits=nt) if threadpool_limits is not None else contextlib.nullcontext()
        with ctx:
            V, state, _ = O.train_step(V, state, cfg, x, y, 0.01, step)      # settle the pool at this size
            step += 1
            t0 = time.time()
            n = 0
            while True:
                V, state, _ = O.train_step(V, state, cfg, x, y, 0.01, step)
                step += 1
                n += 1
                if time.time() - t0 > seconds_budget / (2.0 * len(counts)) or n >= 6:
                    break
            dt = time.time() - t0
        rows.append({"cores": int(nt), "value": round(cb * n / dt, 2), "steps": n, "seconds": round(dt, 1)})
    best = max(rows, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "chunks/s", "cores": best["cores"], "kind": "port",
            "sample": "oracle (NumPy/OpenBLAS fp32 port, this repo) train_step, %d chunks x %d frames x %d-dim, %d speakers, %d steps in %.1f s "
                      "(fastest of the BLAS thread counts in `rows`)" % (cb, T, D, NSPK, best["steps"], best["seconds"]),
            "rows": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames", default=str(T), help="frames per chunk: N (default %d = shape S1) or LO:HI = a length drawn per step "
                    "from a seeded stream, as the reference's loader does (SURVEY 8d shape S3)" % T)
    ap.add_argument("--chunks", type=int, default=B, help="chunks per GPU per step (default %d)" % B)
    ap.add_argument("--attention", action="store_true", help="self-attention pooling of nnet_conf/*_tdnn4_att.json instead of "
                    "statistics pooling (SURVEY 8d shape S4, BASELINE configs[3])")
    ap.add_argument("--precision", choices=["f32", "f16x3"], default=None,
                    help="frame-level GEMM arithmetic (default: the engine's default, env XV_PRECISION)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    # XV_SHARE_GPU=1 (tests on a 1-GPU box): ranks share the visible devices round-robin and talk over gloo - RCCL refuses
    # two ranks on one device.  The measured path is the same code; the numbers of such a run mean nothing.
    share = os.environ.get("XV_SHARE_GPU") == "1"
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    chunks = args.chunks
    if ":" in args.frames:
        t_lo, t_hi = [int(v) for v in args.frames.split(":")]
    else:
        t_lo = t_hi = int(args.frames)
    from tf_kaldi_speaker_amd import _lib, engine as E
    from tf_kaldi_speaker_amd.parallel import GradAllReduce
    lib = _lib.load()

    cfg = E.make_config(D, NSPK, loss_func="additive_margin_softmax", margin_m=0.2, lambda_min=0.0, lambda_base=1000.0,
                        lambda_gamma=1e-4, lambda_power=5.0, last_layer_linear=True, weight_l2_regularizer=1e-2,
                        batchnorm_momentum=0.99, optimizer="sgd", max_batch=chunks, max_frames=t_hi, precision=args.precision,
                        pooling_type="self_attention" if args.attention else "statistics_pooling")
    precision = {v: k for k, v in _lib.PRECISIONS.items()}[int(cfg.precision)]
    eng = E.Engine(cfg, device=str(dev))
    eng.init_variables(seed=0)       # identical replicas on every rank
    rs = np.random.RandomState(1000 + rank)
    nb = 4 if t_lo == t_hi else 16   # rotate a few resident batches (variable length: 16 seeded draws of T)
    ts = [int(rs.randint(t_lo, t_hi + 1)) for _ in range(nb)]
    xs = [torch.from_numpy(rs.randn(chunks, ts[i], D).astype(np.float32)).to(dev) for i in range(nb)]
    ys = [torch.from_numpy(rs.randint(0, NSPK, chunks).astype(np.int32)).to(dev) for _ in range(nb)]
    allreduce = GradAllReduce(dist, world) if world > 1 else None
    lr = 0.01

    def one_step(i):
        eng.train_step(xs[i % nb], ys[i % nb], lr, i, allreduce=allreduce)

    if world > 1 and t_lo != t_hi:
        sys.exit("bench.py: variable-length batches are a single-GPU diagnostic (ranks would draw different T)")

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    launches_per_step = 64
    # Probe pass (untimed, every GEMM kind bracketed): finds the dominant kernel and fills the per-kind table.
    # An event pair costs a few microseconds of queue time per launch (~0.12 ms/step over the ~54 GEMM launches),
    # so the timed region below brackets the dominant kind only.
    probe_steps = 3
    _lib.check(lib.xv_profile_begin(int(probe_steps * launches_per_step)), "xv_profile_begin")
    for i in range(probe_steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize()
    pcnt = (C.c_int64 * NKINDS)()
    pms = (C.c_double * NKINDS)()
    pfl = (C.c_double * NKINDS)()
    _lib.check(lib.xv_profile_end(pcnt, pms, pfl), "xv_profile_end")
    dom = int(np.argmax([pms[k] for k in range(NKINDS)]))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.check(lib.xv_profile_begin_kinds(int(args.steps * launches_per_step), 1 << dom), "xv_profile_begin_kinds")
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    cnt = (C.c_int64 * NKINDS)()
    ms = (C.c_double * NKINDS)()
    fl = (C.c_double * NKINDS)()
    _lib.check(lib.xv_profile_end(cnt, ms, fl), "xv_profile_end")
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # Untimed extra pass with the side stream off: the weight-gradient launches of the timed region run
    # concurrently with data-gradient launches, so their in-region durations include time shared with
    # another kernel.  This pass times every GEMM launch alone (kernel quality in isolation).
    iso_steps = 5
    _lib.check(lib.xv_engine_set_concurrency(eng.h, 0))
    one_step(args.warmup + args.steps)
    torch.cuda.synchronize()
    _lib.check(lib.xv_profile_begin(int(iso_steps * launches_per_step)), "xv_profile_begin")
    for i in range(iso_steps):
        one_step(args.warmup + args.steps + 1 + i)
    torch.cuda.synchronize()
    icnt = (C.c_int64 * NKINDS)()
    ims = (C.c_double * NKINDS)()
    ifl = (C.c_double * NKINDS)()
    _lib.check(lib.xv_profile_end(icnt, ims, ifl), "xv_profile_end")
    _lib.check(lib.xv_engine_set_concurrency(eng.h, 1))
    raw, reg = eng.losses()
    if not np.isfinite(raw):
        sys.exit("bench.py: loss is not finite (%r)" % raw)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * chunks * args.steps / elapsed
        t_mean = float(np.mean([ts[(args.warmup + i) % nb] for i in range(args.steps)]))
        fl_steps = [step_flops(chunks, ts[(args.warmup + i) % nb], D, NSPK, args.attention) for i in range(args.steps)]
        fwd_flops, total_flops = float(np.mean([f[0] for f in fl_steps])), float(np.mean([f[1] for f in fl_steps]))
        by_steps = float(np.mean([step_bytes(chunks, ts[(args.warmup + i) % nb], D, NSPK) for i in range(args.steps)]))
        peak = KIND_PEAK[dom]
        kernels = []
        for k in range(NKINDS):
            if pcnt[k]:     # probe pass: same schedule as the timed region, every kind bracketed
                kernels.append({"kernel": KIND_NAMES[k], "launches_per_step": int(pcnt[k]) // probe_steps, "avg_ms": pms[k] / pcnt[k],
                                "tflops": pfl[k] / (pms[k] * 1e-3) / 1e12, "share_of_step": pms[k] / probe_steps / ms_per_step,
                                "isolated_tflops": ifl[k] / (ims[k] * 1e-3) / 1e12 if icnt[k] else None})
        achieved = fl[dom] / (ms[dom] * 1e-3) / 1e12
        out = {
            "metric": "utterance-chunks/sec (%s-frame x 30-dim)" % (args.frames.replace(":", "-")),
            "value": round(value, 1),
            "unit": "chunks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if precision == "f32" else "f32 (frame-level GEMMs as 3 fp16-plane MFMA products, fp32 accumulate; fp32-level results)",
            "data": "synthetic",
            "config": {"workload": "TDNN x-vector (tdnn.py 5 frame + 2 segment layers, %s) + AM-Softmax m=0.2, "
                                   "full optimiser step (fwd+bwd+L2+SGD%s), %d chunks/GPU x %s frames x %d-dim, %d speakers"
                                   % ("self-attention pooling 512-1500-1500 keys" if args.attention else "stat pooling",
                                      "+RCCL all-reduce" if world > 1 else "", chunks, args.frames, D, NSPK),
                       "chunks_per_gpu": chunks, "frames": t_lo if t_lo == t_hi else [t_lo, t_hi], "mean_frames": t_mean,
                       "feat_dim": D, "num_speakers": NSPK,
                       "precision": precision, "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma", "kernel": KIND_NAMES[dom], "achieved": round(achieved, 2), "peak": round(peak, 1),
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": pmc_traffic(dom),
                         "peak_note": ("dense fp32-input MFMA peak" if dom < 3 else
                                       "algorithmic peak of the f16x3 scheme = dense fp16 MFMA peak (16 x 157.3 TF) / 3 products; "
                                       "achieved counts algorithmic 2*M*N*K once, executed MFMA FLOPs are 3x that"),
                         "achieved_vs_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic_unit": "bytes/launch past L2 (rocprofv3 PMC, profiles/)",
                         "avg_launch_ms": round(ms[dom] / cnt[dom], 4), "launches": int(cnt[dom]),
                         "algorithmic_flops_per_launch": fl[dom] / cnt[dom],
                         "note": "weight-gradient launches (side stream) overlap data-gradient launches in the timed region; "
                                 "`isolated_*` = same kernel timed alone in an extra untimed pass of %d steps" % iso_steps,
                         "isolated_achieved": round(ifl[dom] / (ims[dom] * 1e-3) / 1e12, 2),
                         "isolated_frac": round(ifl[dom] / (ims[dom] * 1e-3) / 1e12 / peak, 4)},
            "step_flops": {"algorithmic_gflop_per_step": round(total_flops / 1e9, 1),
                           "whole_step_tflops": round(total_flops / (ms_per_step * 1e-3) / 1e12, 2),
                           "whole_step_frac_of_f32_mfma_peak": round(total_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                           "whole_step_frac_of_f16x3_peak": round(total_flops / (ms_per_step * 1e-3) / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3.0), 4)},
            "step_bytes": {"algorithmic_mb_per_step": round(by_steps / 1e6, 1),
                           "whole_step_gbs": round(by_steps / (ms_per_step * 1e-3) / 1e9, 1),
                           "whole_step_frac_of_hbm_peak": round(by_steps / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                           "note": "compulsory bytes under perfect fusion (5 S + 2 X + 5 P); the step is MFMA-bound, this fraction "
                                   "cannot exceed ~0.25 even at the f16x3 MFMA roof"},
            "kernels": kernels,
            "kernels_note": "per-kind figures come from an untimed %d-step probe pass with every GEMM launch bracketed by HIP events; "
                            "the timed region brackets only the dominant kind (`roofline`), which keeps the events' queue time out of `value`" % probe_steps,
            "loss": round(raw, 5),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
