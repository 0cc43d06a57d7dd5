#!/usr/bin/env python3
"""SURVEY.md section 8(d) sweep: shapes S1-S4 (+ S5 = the 10-layer extended TDNN of BASELINE configs[4]) x seeds 0..4, >= 20 warm-up + >= 100 timed optimiser steps each,
per-step durations from HIP events recorded at the step boundaries on the launching stream -> median / p10 / p90.

  python tools/shape_sweep.py [--seeds 5] [--steps 100] [--warmup 20] [--precision f16x3|f32] > profiles/rNN_shapes.json

S1 128 chunks x 200 frames, S2 128 x 400, S3 64 x T ~ U{200..400} drawn per step from a seeded stream (rates are
reported per step and T-weighted = total frames / total time), S4 = S1 + the self-attention head of
nnet_conf/*_tdnn4_att.json.  Synthetic features resident in HBM, 30-dim, 7351 speakers, AM-Softmax m = 0.2, SGD + L2.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import D, NSPK, step_flops, EXTENDED_LAYERS, REFERENCE_LAYERS

# S5 = BASELINE configs[4]: the 10-layer extended-context TDNN + A-Softmax at 400-frame chunks (no reference counterpart)
SHAPES = [("S1", 128, 200, 200, False, False), ("S2", 128, 400, 400, False, False), ("S3", 64, 200, 400, False, False),
          ("S4", 128, 200, 200, True, False), ("S5", 128, 400, 400, False, True)]


def run(shape, seed, steps, warmup, precision):
    from tf_kaldi_speaker_amd import engine as E
    name, chunks, t_lo, t_hi, att, ext = shape
    loss_kw = (dict(loss_func="asoftmax", margin_m=4, lambda_min=10.0, lambda_base=1000.0, lambda_gamma=1e-5, lambda_power=5.0) if ext else
               dict(loss_func="additive_margin_softmax", margin_m=0.2, lambda_min=0.0, lambda_base=1000.0, lambda_gamma=1e-4, lambda_power=5.0))
    cfg = E.make_config(D, NSPK, last_layer_linear=True, weight_l2_regularizer=1e-2,
                        batchnorm_momentum=0.99, optimizer="sgd", max_batch=chunks, max_frames=t_hi, precision=precision,
                        pooling_type="self_attention" if att else "statistics_pooling", frame_layers=EXTENDED_LAYERS if ext else None, **loss_kw)
    eng = E.Engine(cfg, device="cuda:0")
    eng.init_variables(seed=seed)
    rs = np.random.RandomState(seed)
    nb = 4 if t_lo == t_hi else 16
    ts = [int(rs.randint(t_lo, t_hi + 1)) for _ in range(nb)]
    xs = [torch.from_numpy(rs.randn(chunks, ts[i], D).astype(np.float32)).cuda() for i in range(nb)]
    ys = [torch.from_numpy(np.random.RandomState(seed + 1 + i).randint(0, NSPK, chunks).astype(np.int32)).cuda() for i in range(nb)]
    for i in range(warmup):
        eng.train_step(xs[i % nb], ys[i % nb], 0.01, i)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        eng.train_step(xs[(warmup + i) % nb], ys[(warmup + i) % nb], 0.01, warmup + i)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])
    tt = np.array([ts[(warmup + i) % nb] for i in range(steps)], dtype=np.float64)
    raw, _ = eng.losses()
    assert np.isfinite(raw), "loss is not finite"
    fl = np.array([step_flops(chunks, int(t), D, NSPK, att, EXTENDED_LAYERS if ext else REFERENCE_LAYERS)[1] for t in tt])
    del eng
    return ms, tt, fl, chunks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default=None)
    ap.add_argument("--shapes", default="S1,S2,S3,S4,S5")
    args = ap.parse_args()
    out = {"unit": "chunks/s", "steps": args.steps, "warmup": args.warmup, "seeds": list(range(args.seeds)),
           "precision": args.precision or "default (f32)", "device": torch.cuda.get_device_name(0), "shapes": {}}
    for shape in SHAPES:
        if shape[0] not in args.shapes.split(","):
            continue
        rate, msall, frames, flops = [], [], 0.0, 0.0
        per_seed = []
        for seed in range(args.seeds):
            ms, tt, fl, chunks = run(shape, seed, args.steps, args.warmup, args.precision)
            rate.append(chunks / (ms * 1e-3))
            msall.append(ms)
            frames += float((tt * chunks).sum())
            flops += float(fl.sum())
            per_seed.append(round(float(chunks * len(ms) / (ms.sum() * 1e-3)), 1))
        rate = np.concatenate(rate)
        msall = np.concatenate(msall)
        total_s = float(msall.sum() * 1e-3)
        out["shapes"][shape[0]] = {
            "chunks_per_step": shape[1], "frames": shape[2] if shape[2] == shape[3] else [shape[2], shape[3]], "attention": shape[4],
            "extended_10_layer_tdnn": shape[5],
            "chunks_per_s": {"median": round(float(np.median(rate)), 1), "p10": round(float(np.percentile(rate, 10)), 1),
                             "p90": round(float(np.percentile(rate, 90)), 1), "mean_over_time": round(shape[1] * len(msall) / total_s, 1)},
            "ms_per_step": {"median": round(float(np.median(msall)), 4), "p10": round(float(np.percentile(msall, 10)), 4),
                            "p90": round(float(np.percentile(msall, 90)), 4)},
            "frames_per_s": round(frames / total_s, 1),
            "algorithmic_tflops": round(flops / total_s / 1e12, 2),
            "per_seed_chunks_per_s": per_seed,
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
