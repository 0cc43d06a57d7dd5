#!/usr/bin/env python3
"""Headline benchmark: utterance-chunks/sec of full x-vector optimiser steps on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; called WITHOUT a launcher
   - no WORLD_SIZE in the environment - `--gpus N` starts that command itself as a child process, before anything touches the GPU,
   relays rank 0's JSON line and exits with the child's status)

Workload (BASELINE.json configs[1], SURVEY.md section 8d shape S1): standard 5-layer TDNN
x-vector + AM-Softmax (m = 0.2), 128 chunks x 200 frames x 30-dim MFCC per GPU, 7351 speakers,
plain SGD + L2 - one "step" = forward + loss + backward (+ gradient all-reduce for N > 1)
+ update on synthetic features already resident in HBM.  Weak scaling: 128 chunks per GPU.

Prints ONE JSON line on rank 0 (contract in the task statement).  The headline (`value`, `ms_per_step`,
`roofline`, `dtype` "f32") is the engine's default arithmetic: fp32 operands on the fp32-input MFMA, the
reference's own dtype.  At N = 1 the same run then times the opt-in split-precision mode (`precision:
"f16x3"`: every fp32 operand as two fp16 planes, three fp16 MFMA products per fp32 product, fp32
accumulate) for the same K steps and reports it separately under `"f16x3"` with its own roofline; an
`e2e` leg repeats the headline with the host->device copy of each feature batch inside the timed
region; `cpu_baseline` = the NumPy oracle's fp32 train_step on the full 128-chunk batch on this box's
host cores (1 thread and the faster multi-thread settings).  `roofline` is for the dominant kernel: its
launches are bracketed by hipEvents on their stream inside the timed region; an untimed probe pass
brackets every MFMA GEMM kind for the per-kernel table.  At N > 1 `comm` carries, per rank, the RCCL
world size and the all-reduce time of every gradient slice (events on the communication stream).
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
KIND_NAMES = ["xv_gemm_nt_kernel<true, *> / xv_gemm_nt_sk_kernel<true, *> (forward conv/dense + BN stats)",
              "xv_gemm_nt_kernel<false, *> / xv_gemm_nt_sk_kernel<false, *> (data gradients / logits)",
              "xv_gemm_tn_kernel (weight gradients)",
              "xv_gemm16_nt_kernel<true> (f16x3 forward conv/dense + BN stats)",
              "xv_gemm16_nt_kernel<false> (f16x3 data gradients)",
              "xv_gemm16_tn_kernel (f16x3 weight gradients)"]
KIND_SYMBOLS = ["xv_gemm_nt_kernel<true,", "xv_gemm_nt_kernel<false,", "xv_gemm_tn_kernel",
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


REFERENCE_LAYERS = ((5, 512), (5, 512), (7, 512), (1, 512), (1, 1500))          # tdnn.py:35-127
# BASELINE configs[4] "Deep TDNN (extended context, 10 layers)": no reference counterpart; the 10-layer shape of the
# extended x-vector recipe with contiguous contexts (tf_kaldi_speaker_amd/model/tdnn.py DEFAULT_EXTENDED_LAYERS)
EXTENDED_LAYERS = ((5, 512), (1, 512), (3, 512), (1, 512), (3, 512), (1, 512), (3, 512), (1, 512), (1, 512), (1, 1500))


def step_flops(b, t, d, n, attention=False, layers=REFERENCE_LAYERS):
    """2*M*K*N per contraction, backward = 2x forward (attention = shape S4: the key network
    512 -> 1500 -> 1500 of nnet_conf/*_tdnn4_att.json on the pooled frames)."""
    fwd, cin, tt = 0.0, d, t
    for k, w in layers:
        tt -= k - 1
        fwd += 2.0 * b * tt * k * cin * w
        cin = w
    fwd += 2.0 * b * (2 * cin * 512 + 512 * 512 + 512 * n)
    if attention:
        fwd += 2.0 * b * tt * (layers[-2][1] * 1500 + 1500 * 1500)
    return fwd, 3.0 * fwd


def step_bytes(b, t, d, n, layers=REFERENCE_LAYERS):
    """Compulsory HBM traffic of one step under perfect fusion: every frame-level activation
    written once + read once forward, read once more backward, its gradient written + read (5 S); input read twice;
    parameters read fwd + bwd, gradient written + read, parameter written (5 P).  fp32 bytes."""
    S, P, cin, tt = 0.0, 0.0, d, t
    for k, w in layers:
        tt -= k - 1
        S += 4.0 * b * tt * w
        P += 4.0 * (k * cin * w + 5 * w)
        cin = w
    P += 4.0 * (2 * cin * 512 + 512 * 512 + 512 * n + 2 * 512 * 5)
    return 5.0 * S + 2.0 * 4.0 * b * t * d + 5.0 * P


def cgroup_cpus():
    """CPUs' worth of run time this process's cgroup may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited.  os.cpu_count()
    and the affinity mask do not show it: the round-6 GPU boxes report 256 CPUs and run under a 16-CPU quota, and every host-side figure
    (this baseline's best thread count, the loaders' aggregate rate) is a figure for THAT many cores."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else round(int(quota) / float(period), 2)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
            quota, period = int(f.read()), int(g.read())
        return None if quota <= 0 else round(quota / float(period), 2)
    except (OSError, ValueError):
        return None


def cgroup_throttle():
    """(periods in which this cgroup ran out of CPU quota, microseconds its threads were held) so far, or None."""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            st = dict(line.split()[:2] for line in f if line.strip())
        return int(st["nr_throttled"]), int(st["throttled_usec"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(seconds_budget=70.0):
    """The oracle (a NumPy port of the reference arithmetic, NOT TensorFlow) on the host cores: ONE fp32 train_step of the
    full S1 batch (128 chunks x 200 frames, 7351 speakers) per BLAS thread count - 1 thread (comparable to the reference's
    single_cpu mode, trainer.py:46-50), then 8 / 16 / 32 threads, the cgroup's CPU quota and every core the host reports, as far as
    the budget allows ("every core" is not the best row: the process may only use `cgroup_cpus` of them at a time, more threads than
    that are throttled, and the port is part NumPy element-wise passes).  `value` is the fastest row, `cores` the threads it used."""
    from oracle import xvector_oracle as O
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    ncpu = os.cpu_count() or 1
    cfg = O.Config(feat_dim=D, num_speakers=NSPK, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True)
    V = O.init_variables(cfg, seed=0, dtype=np.float32)
    rs = np.random.RandomState(0)
    x = rs.randn(B, T, D).astype(np.float32)
    y = rs.randint(0, NSPK, B)
    O.train_step(V, {}, cfg, x[:8], y[:8], 0.01, 0)      # warm-up (BLAS thread pool, allocator)
    # rows: 1 thread (comparable to the reference's single_cpu mode), a few mid counts, and EVERY core of the host - three steps each
    # (median) while the budget lasts; the 1-thread and all-core rows always get their three
    quota = cgroup_cpus()
    counts = [1] if threadpool_limits is None else sorted({1, min(8, ncpu), min(16, ncpu), min(32, ncpu), ncpu} |
                                                          ({max(1, min(int(quota), ncpu))} if quota else set()))
    must = {1, ncpu}
    rows, t_start = [], time.time()
    import contextlib
    step_no = [1]

    def timed_steps(nt, n):
        out = []
        ctx = threadpool_limits(limits=nt) if threadpool_limits is not None else contextlib.nullcontext()
        with ctx:
            for _ in range(n):
                t0 = time.time()
                O.train_step(V, {}, cfg, x, y, 0.01, step_no[0])
                step_no[0] += 1
                out.append(time.time() - t0)
        return out
    for nt in counts:
        spent = time.time() - t_start
        if nt not in must and rows and spent + 3 * rows[-1]["seconds"] > seconds_budget:
            continue
        ts = timed_steps(nt, 3 if (nt in must or spent < 0.6 * seconds_budget) else 1)
        rows.append({"cores": int(nt), "value": round(B / float(np.median(ts)), 2), "steps": len(ts), "seconds": round(float(np.median(ts)), 2),
                     "step_seconds": [round(t, 3) for t in ts]})
    best = max(rows, key=lambda r: r["value"])
    # the headline row: more steps at the fastest thread count (at most seven in all), `value` = the median step
    times = list(best["step_seconds"])
    while len(times) < 7 and time.time() - t_start + times[-1] < seconds_budget:
        times += timed_steps(best["cores"], 1)
    med = float(np.median(times))
    return {"value": round(B / med, 2), "unit": "chunks/s", "cores": best["cores"], "kind": "port", "host_cpus": ncpu, "cgroup_cpus": quota,
            "steps": len(times), "step_seconds": [round(t, 3) for t in times],
            "sample": "oracle (NumPy/OpenBLAS fp32 port of the reference arithmetic, this repo - not TensorFlow) train_step on the benchmark "
                      "batch (%d chunks x %d frames x %d-dim, %d speakers): three steps per BLAS thread count (`rows`: 1 thread, mid counts, every "
                      "host core), then up to %d steps at the fastest count; `value` = the median step there" % (B, T, D, NSPK, len(times)),
            "rows": rows}


def _profile_table(lib, _lib, which):
    cnt = (C.c_int64 * NKINDS)()
    ms = (C.c_double * NKINDS)()
    fl = (C.c_double * NKINDS)()
    _lib.check(lib.xv_profile_end(cnt, ms, fl), "xv_profile_end(%s)" % which)
    return [int(c) for c in cnt], [float(m) for m in ms], [float(f) for f in fl]


def run_mode(precision, args, dev, rank, world, dist, chunks, t_lo, t_hi, h2d=False, light=False):
    """Build an engine in `precision`, warm up, then time EXACTLY args.steps optimiser steps between barrier +
    synchronize brackets.  h2d: every step first copies its feature / label batch from pinned host memory on the compute
    stream (the reference boundary hands host arrays).  light: skip the probe / isolated passes (e2e leg)."""
    from tf_kaldi_speaker_amd import _lib, engine as E
    from tf_kaldi_speaker_amd.parallel import GradAllReduce
    lib = _lib.load()
    loss_kw = (dict(loss_func="asoftmax", margin_m=4, lambda_min=10.0, lambda_base=1000.0, lambda_gamma=1e-5, lambda_power=5.0) if args.extended else
               dict(loss_func="additive_margin_softmax", margin_m=0.2, lambda_min=0.0, lambda_base=1000.0, lambda_gamma=1e-4, lambda_power=5.0))
    cfg = E.make_config(D, NSPK, last_layer_linear=True, weight_l2_regularizer=1e-2,
                        batchnorm_momentum=0.99, optimizer="sgd", max_batch=chunks, max_frames=t_hi, precision=precision,
                        pooling_type="self_attention" if args.attention else "statistics_pooling",
                        frame_layers=EXTENDED_LAYERS if args.extended else None, **loss_kw)
    eng = E.Engine(cfg, device=str(dev))
    eng.init_variables(seed=0)       # identical replicas on every rank
    rs = np.random.RandomState(1000 + rank)
    nb = 4 if t_lo == t_hi else 16   # rotate a few batches (variable length: 16 seeded draws of T)
    ts = [int(rs.randint(t_lo, t_hi + 1)) for _ in range(nb)]
    hx = [torch.from_numpy(rs.randn(chunks, ts[i], D).astype(np.float32)) for i in range(nb)]
    hy = [torch.from_numpy(rs.randint(0, NSPK, chunks).astype(np.int32)) for _ in range(nb)]
    if h2d:
        # the product's feed (NativeRandomQueue.device_batches): pinned host batch -> fresh device tensors on a COPY stream that never
        # waits for the GPU (the host runs a step or more ahead, so the copy of batch i lands while step i - 1 computes); the
        # compute stream waits for the copy's event.  [measured] a copy stream that ALSO waited on the compute stream's events (to
        # recycle fixed staging buffers) cost 0.2 - 1.9 ms/step depending on which hardware queue the streams happened to share
        hx, hy = [t.pin_memory() for t in hx], [t.pin_memory() for t in hy]
        copy_stream = torch.cuda.Stream(device=dev)
        import collections
        consumed = collections.deque()      # as the loader: the host stays at most 4 batches ahead of the compute stream

        def feed(i):
            j = i % nb
            if len(consumed) > 4:
                consumed.popleft().synchronize()
            with torch.cuda.stream(copy_stream):
                x = hx[j].to(dev, non_blocking=True)
                y = hy[j].to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)
            x.record_stream(cur)
            y.record_stream(cur)
            return x, y
    else:
        xs, ys = [t.to(dev) for t in hx], [t.to(dev) for t in hy]
    allreduce = GradAllReduce(dist, world, timing=True) if world > 1 else None
    lr = 0.01

    def one_step(i):
        j = i % nb
        if h2d:
            x, y = feed(i)
            eng.train_step(x, y, lr, i, allreduce=allreduce)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
            consumed.append(done)
        else:
            eng.train_step(xs[j], ys[j], lr, i, allreduce=allreduce)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    fence()
    launches_per_step = 96
    res = {"precision": precision, "ts": ts, "nb": nb}
    step0 = args.warmup
    if not light:
        # Probe pass (untimed, every GEMM kind bracketed): finds the dominant kernel and fills the per-kind table.
        # An event pair costs a few microseconds of queue time per launch (~0.12 ms/step over the ~54 GEMM launches),
        # so the timed region below brackets the dominant kind only.
        probe_steps = 3
        _lib.check(lib.xv_profile_begin(int(probe_steps * launches_per_step)), "xv_profile_begin")
        for i in range(probe_steps):
            one_step(step0 + i)
        torch.cuda.synchronize()
        res["probe"] = _profile_table(lib, _lib, "probe") + (probe_steps,)
        dom = int(np.argmax(res["probe"][1]))
        res["dom"] = dom
        fence()
        _lib.check(lib.xv_profile_begin_kinds(int(args.steps * launches_per_step), 1 << dom), "xv_profile_begin_kinds")
    if allreduce is not None:
        allreduce.reset_timing()
    cpu0 = time.process_time()                    # every thread of this process (HIP / RCCL helper threads included)
    thr0 = cgroup_throttle()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(step0 + i)
    t_enq = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    # host side of the timed region: CPUs this rank kept busy (the wait in fence() spins) and the share of the region its thread spent
    # enqueueing - on a box whose cgroup allows fewer CPUs than N ranks burn, the ranks get throttled and the GPU path is not what is measured
    res["host"] = {"cpus_busy": round((time.process_time() - cpu0) / max(elapsed, 1e-9), 2), "enqueue_frac": round(t_enq / max(elapsed, 1e-9), 3),
                   "cgroup_cpus": cgroup_cpus(), "os_cpus": os.cpu_count()}
    thr1 = cgroup_throttle()
    if thr0 is not None and thr1 is not None:     # (cgroup-wide: every rank of the job counts into the same figures)
        res["host"].update(throttled_periods=thr1[0] - thr0[0], throttled_thread_ms=round((thr1[1] - thr0[1]) / 1e3, 1))
    if not light:
        res["timed"] = _profile_table(lib, _lib, "timed")
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    res["elapsed"] = elapsed
    res["comm"] = None
    if allreduce is not None:
        try:
            res["comm"] = allreduce.timing_report()
        except Exception as exc:       # the report must never cost the bench line
            res["comm"] = {"error": repr(exc)}
    if not light:
        # Untimed extra pass with the side stream off: the weight-gradient launches of the timed region run
        # concurrently with data-gradient launches, so their in-region durations include time shared with
        # another kernel.  This pass times every GEMM launch alone (kernel quality in isolation).
        iso_steps = 5
        _lib.check(lib.xv_engine_set_concurrency(eng.h, 0))
        one_step(step0 + args.steps)
        torch.cuda.synchronize()
        _lib.check(lib.xv_profile_begin(int(iso_steps * launches_per_step)), "xv_profile_begin")
        for i in range(iso_steps):
            one_step(step0 + args.steps + 1 + i)
        torch.cuda.synchronize()
        res["iso"] = _profile_table(lib, _lib, "isolated") + (iso_steps,)
        _lib.check(lib.xv_engine_set_concurrency(eng.h, 1))
    raw, reg = eng.losses()
    if not np.isfinite(raw):
        sys.exit("bench.py: loss is not finite (%r)" % raw)
    res["loss"] = raw
    # replicas must stay bit-identical (same initial variables, summed gradients, same update): an order-free integer checksum of the
    # trainable variables' bit patterns, compared across ranks by whoever reads the `comm` report (BN moving statistics are local)
    res["trainable_checksum"] = int(eng.variables[:eng.n_train].view(torch.int32).to(torch.int64).sum().item())
    eng.close()
    del eng
    torch.cuda.empty_cache()
    return res


def summarize(res, args, world, chunks):
    """value / ms_per_step / roofline / per-kernel table of one timed mode."""
    ts, nb = res["ts"], res["nb"]
    steps_t = [ts[(args.warmup + i) % nb] for i in range(args.steps)]
    ms_per_step = res["elapsed"] / args.steps * 1e3
    layers = EXTENDED_LAYERS if args.extended else REFERENCE_LAYERS
    fl_steps = [step_flops(chunks, t, D, NSPK, args.attention, layers) for t in steps_t]
    total_flops = float(np.mean([f[1] for f in fl_steps]))
    by_steps = float(np.mean([step_bytes(chunks, t, D, NSPK, layers) for t in steps_t]))
    out = {"value": round(world * chunks * args.steps / res["elapsed"], 1), "ms_per_step": round(ms_per_step, 4),
           "mean_frames": float(np.mean(steps_t)), "loss": round(res["loss"], 5)}
    if "dom" in res:
        dom = res["dom"]
        pcnt, pms, pfl, probe_steps = res["probe"]
        cnt, ms, fl = res["timed"]
        icnt, ims, ifl, iso_steps = res["iso"]
        peak = KIND_PEAK[dom]
        kernels = []
        for k in range(NKINDS):
            if pcnt[k]:     # probe pass: same schedule as the timed region, every kind bracketed
                kernels.append({"kernel": KIND_NAMES[k], "launches_per_step": pcnt[k] // probe_steps, "avg_ms": round(pms[k] / pcnt[k], 5),
                                "tflops": round(pfl[k] / (pms[k] * 1e-3) / 1e12, 2), "frac_of_peak": round(pfl[k] / (pms[k] * 1e-3) / 1e12 / KIND_PEAK[k], 4),
                                "share_of_step": round(pms[k] / probe_steps / ms_per_step, 4),
                                "isolated_tflops": round(ifl[k] / (ims[k] * 1e-3) / 1e12, 2) if icnt[k] else None})
        achieved = fl[dom] / (ms[dom] * 1e-3) / 1e12
        out["roofline"] = {
            "bound": "mfma", "kernel": KIND_NAMES[dom], "achieved": round(achieved, 2), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": pmc_traffic(dom),
            "peak_note": ("dense fp32-input MFMA peak (MI355X_MICROARCH.md: 157.3 TF)" if dom < 3 else
                          "algorithmic peak of the f16x3 scheme = dense fp16 MFMA peak (16 x 157.3 TF) / 3 products; "
                          "achieved counts algorithmic 2*M*N*K once, executed MFMA FLOPs are 3x that"),
            "achieved_vs_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
            "traffic_unit": "bytes/launch past L2 (rocprofv3 PMC, profiles/)",
            "avg_launch_ms": round(ms[dom] / cnt[dom], 4), "launches": int(cnt[dom]),
            "algorithmic_flops_per_launch": fl[dom] / cnt[dom],
            "note": "weight-gradient launches (side stream) overlap data-gradient launches in the timed region; "
                    "`isolated_*` = same kernel timed alone in an extra untimed pass of %d steps" % iso_steps,
            "isolated_achieved": round(ifl[dom] / (ims[dom] * 1e-3) / 1e12, 2),
            "isolated_frac": round(ifl[dom] / (ims[dom] * 1e-3) / 1e12 / peak, 4)}
        out["kernels"] = kernels
    out["step_flops"] = {"algorithmic_gflop_per_step": round(total_flops / 1e9, 1),
                         "whole_step_tflops": round(total_flops / (ms_per_step * 1e-3) / 1e12, 2),
                         "whole_step_frac_of_f32_mfma_peak": round(total_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                         "whole_step_frac_of_f16x3_peak": round(total_flops / (ms_per_step * 1e-3) / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3.0), 4)}
    out["step_bytes"] = {"algorithmic_mb_per_step": round(by_steps / 1e6, 1),
                         "whole_step_gbs": round(by_steps / (ms_per_step * 1e-3) / 1e9, 1),
                         "whole_step_frac_of_hbm_peak": round(by_steps / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
    return out


def visible_gpu_count():
    """GPUs this process would see, WITHOUT initialising HIP (the parent of the ranks must never hold the GPU): the *_VISIBLE_DEVICES
    lists when set, else the KFD topology (nodes with SIMDs).  None when neither source says anything."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        return n
    except (OSError, ValueError):
        return None


def rank_diagnostics(dev, local_rank):
    """What a first multi-GPU run is read with, per rank: which device this rank really sits on (ordinal, PCI address, uuid), the RCCL
    build torch carries, and the environment that decides how RCCL moves bytes between the ranks.  Never raises."""
    out = {"local_rank": local_rank, "pid": os.getpid()}
    try:
        pr = torch.cuda.get_device_properties(dev)
        out["pci_bus_id"] = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
        out["device_uuid"] = str(getattr(pr, "uuid", ""))
        out["gcn_arch"] = getattr(pr, "gcnArchName", None)
        out["compute_units"] = getattr(pr, "multi_processor_count", None)
        out["hbm_gb"] = round(pr.total_memory / 2 ** 30, 1)
    except Exception as exc:
        out["device_properties_error"] = repr(exc)
    try:
        import torch.cuda.nccl as _nccl
        out["rccl_version"] = ".".join(str(v) for v in _nccl.version())
    except Exception as exc:
        out["rccl_version"] = "unknown (%r)" % (exc,)
    out["hip_version"] = getattr(torch.version, "hip", None)
    out["env"] = {k: v for k, v in sorted(os.environ.items())
                  if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_VISIBLE", "ROCR_VISIBLE", "CUDA_VISIBLE", "XV_")) or k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    return out


def failure_line(n, error, **extra):
    """One JSON line for a run that produced no measurement, so that a broken multi-GPU launch still leaves a diagnosable record."""
    out = {"metric": "utterance-chunks/sec (%d-frame x %d-dim)" % (T, D), "value": None, "unit": "chunks/s", "n_gpus": n, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "data": "synthetic", "error": error}
    out.update(extra)
    return json.dumps(out)


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start one rank per GPU as CHILD processes (torch.distributed.run,
    spawned - never exec'd over this process) and relay their output.  This process never initialises HIP: the device count comes
    from the environment / the KFD topology (visible_gpu_count), the rendezvous port is chosen by the launcher itself
    (--standalone).  If the ranks die or print no result, ONE JSON line with "error", the return code and the tail of their stderr
    is printed instead of the bench line."""
    import collections
    import subprocess
    import threading
    have = visible_gpu_count()
    if have is not None and have < n and os.environ.get("XV_SHARE_GPU") != "1":
        msg = ("bench.py --gpus %d: only %d device(s) visible (XV_SHARE_GPU=1 runs the ranks on shared devices over gloo - a wiring "
               "check, not a measurement)" % (n, have))
        print(msg, file=sys.stderr)
        print(failure_line(n, msg, visible_devices=have))
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(n),
           os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1)
    tail = collections.deque(maxlen=25)
    head = []
    got_line = []

    def relay(src, dst, keep):
        for line in src:
            dst.write(line)
            dst.flush()
            if keep is not None:
                keep.append(line.rstrip("\n"))
                if len(head) < 15:
                    head.append(line.rstrip("\n"))
            elif line.startswith("{") and '"metric"' in line:
                got_line.append(True)

    threads = [threading.Thread(target=relay, args=(proc.stdout, sys.stdout, None)), threading.Thread(target=relay, args=(proc.stderr, sys.stderr, tail))]
    for th in threads:
        th.start()
    rc = proc.wait()
    for th in threads:
        th.join()
    if rc != 0 or not got_line:
        print(failure_line(n, "the %d ranks exited with status %d%s" % (n, rc, "" if got_line else " without printing a result"), returncode=rc,
                           stderr_head=head, stderr_tail=[ln for ln in tail if ln not in head]))
        return rc or 1
    return 0


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
    ap.add_argument("--extended", action="store_true", help="BASELINE configs[4]: 10-layer extended-context TDNN + A-Softmax (m = 4) instead of the "
                    "reference's 5-layer TDNN + AM-Softmax; use with --frames 400 (no reference counterpart, SURVEY.md D4)")
    ap.add_argument("--precision", choices=["f32", "f16x3"], default="f32",
                    help="arithmetic of the headline (default f32 = the engine's default); at one GPU the other mode is timed as well "
                         "and reported separately unless --single-mode")
    ap.add_argument("--single-mode", action="store_true", help="time only --precision (no second mode, no e2e leg)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py --gpus %d was launched with WORLD_SIZE=%d (torch.distributed.run --nproc-per-node must equal --gpus)" % (args.gpus, world))
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
    if world > 1 and t_lo != t_hi:
        sys.exit("bench.py: variable-length batches are a single-GPU diagnostic (ranks would draw different T)")

    head = run_mode(args.precision, args, dev, rank, world, dist, chunks, t_lo, t_hi)
    other, e2e = None, None
    if world == 1 and not args.single_mode:
        other = run_mode("f16x3" if args.precision == "f32" else "f32", args, dev, rank, world, dist, chunks, t_lo, t_hi)
        e2e = run_mode(args.precision, args, dev, rank, world, dist, chunks, t_lo, t_hi, h2d=True, light=True)

    comm_all = None
    if dist is not None:       # per-rank communication report, gathered on every rank (collective), printed by rank 0
        try:
            mine = {"rank": rank, "backend": dist.get_backend(), "world_size": dist.get_world_size(), "device": torch.cuda.get_device_name(dev),
                    "device_index": dev_index, "data_seed": 1000 + rank, "loss": head["loss"], "trainable_checksum": head["trainable_checksum"],
                    "report": head["comm"], "host": head["host"]}
            mine.update(rank_diagnostics(dev, local_rank))
            comm_all = [None] * world
            dist.all_gather_object(comm_all, mine)
        except Exception as exc:       # the report must never cost the bench line
            comm_all = [{"rank": rank, "error": repr(exc)}]

    if rank == 0:
        DT = {"f32": "f32",
              "f16x3": "f16x3 (fp32 tensors; frame-level GEMM operands split into 2 fp16 planes = 22-bit significands, 3 fp16 MFMA products per "
                       "fp32 product, fp32 accumulate - opt-in, narrower operands than the reference's fp32)"}
        hs = summarize(head, args, world, chunks)
        out = {
            "metric": "utterance-chunks/sec (%s-frame x 30-dim)" % (args.frames.replace(":", "-")),
            "value": hs["value"],
            "unit": "chunks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": hs["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DT[args.precision],
            "data": "synthetic",
            "config": {"workload": "%s, "
                                   "full optimiser step (fwd+bwd+L2+SGD%s), %d chunks/GPU x %s frames x %d-dim, %d speakers"
                                   % (("extended TDNN x-vector (10 frame layers, contexts 5/1/3/1/3/1/3/1/1/1, + 2 segment layers, %s) + A-Softmax m=4"
                                       if args.extended else "TDNN x-vector (tdnn.py 5 frame + 2 segment layers, %s) + AM-Softmax m=0.2")
                                      % ("self-attention pooling 512-1500-1500 keys" if args.attention else "stat pooling"),
                                      "+RCCL all-reduce" if world > 1 else "", chunks, args.frames, D, NSPK),
                       "chunks_per_gpu": chunks, "frames": t_lo if t_lo == t_hi else [t_lo, t_hi], "mean_frames": hs["mean_frames"],
                       "feat_dim": D, "num_speakers": NSPK, "precision": args.precision, "parallelism": "dp%d" % world},
            "roofline": hs["roofline"],
            "step_flops": hs["step_flops"],
            "step_bytes": dict(hs["step_bytes"], note="compulsory bytes under perfect fusion (5 S + 2 X + 5 P); the step is MFMA-bound, this "
                                                      "fraction cannot exceed ~0.06 in fp32 / ~0.25 at the f16x3 MFMA roof"),
            "kernels": hs["kernels"],
            "kernels_note": "per-kind figures come from an untimed 3-step probe pass with every GEMM launch bracketed by HIP events; "
                            "the timed region brackets only the dominant kind (`roofline`), which keeps the events' queue time out of `value`",
            "loss": hs["loss"],
            "host": head["host"],
        }
        if other is not None:
            os_ = summarize(other, args, world, chunks)
            out[other["precision"]] = {"dtype": DT[other["precision"]], "value": os_["value"], "unit": "chunks/s", "ms_per_step": os_["ms_per_step"],
                                       "steps": args.steps, "warmup": args.warmup, "roofline": os_["roofline"], "step_flops": os_["step_flops"],
                                       "kernels": os_["kernels"], "loss": os_["loss"],
                                       "note": "same workload, same timed-region protocol, reported separately from the headline"}
        if e2e is not None:
            es = summarize(e2e, args, world, chunks)
            out["e2e"] = {"value": es["value"], "unit": "chunks/s", "ms_per_step": es["ms_per_step"], "precision": args.precision,
                          "h2d_bytes_per_step": int(chunks * es["mean_frames"] * D * 4 + chunks * 4),
                          "note": "headline mode with the pinned-host -> device copy of every feature / label batch inside the timed region, fed as "
                                  "the product feeds it (pinned host batch -> device on a copy stream that never waits for the GPU, the compute stream waits for the copy's event); never `value`"}
        if comm_all is not None:
            out["comm"] = comm_all
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as exc:      # a rank that dies still leaves one diagnosable JSON line (rank 0 only; the launcher reports the rest)
        if int(os.environ.get("RANK", "0")) == 0:
            import traceback
            print(failure_line(int(os.environ.get("WORLD_SIZE", "1")), "%s: %s" % (type(exc).__name__, exc), traceback=traceback.format_exc().splitlines()[-8:]))
        raise
