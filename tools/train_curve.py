#!/usr/bin/env python3
"""Training-dynamics check of the split-precision mode: the same synthetic speaker-classification task trained for a few
hundred steps with precision="f16x3" (opt-in) and precision="f32" (default) from identical initial variables and batches.
Per-step GEMM results agree to ~1e-6 relative (tests/), so the two loss curves must stay together until chaotic
divergence of SGD itself - measured by a yardstick run (f32 against f32 started from variables perturbed by 1e-6
relative noise); the script prints both curves, their largest relative gap and the final training accuracy
of nearest-speaker-weight classification, and writes a JSON summary.

  python tools/train_curve.py [steps] > profiles/rNN_train_curve.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from tf_kaldi_speaker_amd import _lib, engine as E

D, NSPK, B, T = 30, 200, 64, 120


def batches(steps, seed=0):
    """Speaker s = a fixed spectral pattern [D] plus a slow per-utterance drift; frames = pattern + noise."""
    rs = np.random.RandomState(seed)
    pattern = rs.randn(NSPK, D).astype(np.float32)
    for _ in range(steps):
        y = rs.randint(0, NSPK, B)
        drift = 0.3 * rs.randn(B, 1, D)
        x = pattern[y][:, None, :] + drift + 1.5 * rs.randn(B, T, D)
        yield x.astype(np.float32), y.astype(np.int32)


def run(precision, steps, perturb=0.0):
    cfg = E.make_config(D, NSPK, loss_func="additive_margin_softmax", margin_m=0.15, lambda_min=0.0, lambda_base=1000.0,
                        lambda_gamma=1e-2, lambda_power=5.0, last_layer_linear=True, weight_l2_regularizer=1e-3,
                        batchnorm_momentum=0.99, optimizer="momentum", momentum=0.9, max_batch=B, max_frames=T, precision=precision)
    eng = E.Engine(cfg, device="cuda:0")
    eng.init_variables(seed=3)
    if perturb:       # the same run from initial variables perturbed by `perturb` relative noise: the divergence yardstick
        g = torch.Generator(device="cpu").manual_seed(11)
        noise = torch.randn(eng.variables.numel(), generator=g).to(eng.variables.device)
        eng.variables.mul_(1.0 + perturb * noise)
        _lib.check(eng.lib.xv_engine_invalidate_weights(eng.h), "xv_engine_invalidate_weights")
    losses, acc = [], []
    for i, (x, y) in enumerate(batches(steps)):
        raw, reg = eng.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), 0.02, i, fetch_losses=True)
        losses.append(float(raw))
        if i >= steps - 20:
            logits = eng.endpoint("logits").cpu().numpy()[:, :NSPK]
            acc.append(float((logits.argmax(1) == y).mean()))
    emb = eng.endpoint("tdnn6_dense").cpu().numpy()
    return np.array(losses), float(np.mean(acc)), emb


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    l16, a16, e16 = run("f16x3", steps)
    l32, a32, e32 = run("f32", steps)
    l32p, a32p, e32p = run("f32", steps, perturb=1e-6)
    gap = np.abs(l16 - l32) / np.maximum(np.abs(l32), 1e-6)
    gap_p = np.abs(l32p - l32) / np.maximum(np.abs(l32), 1e-6)
    out = {"task": "synthetic %d-speaker classification, %d chunks x %d frames x %d-dim, AM-Softmax m=0.15, momentum SGD lr 0.02" % (NSPK, B, T, D),
           "steps": steps,
           "loss_f16x3": [round(float(v), 5) for v in l16[:: max(1, steps // 30)]],
           "loss_f32": [round(float(v), 5) for v in l32[:: max(1, steps // 30)]],
           "max_rel_gap_first_10_steps": float(gap[:10].max()),
           "max_rel_gap_first_50_steps": float(gap[:50].max()), "max_rel_gap_all_steps": float(gap.max()),
           "yardstick_f32_vs_f32_with_1e-6_relative_noise_on_the_initial_variables": {
               "max_rel_gap_first_10_steps": float(gap_p[:10].max()), "max_rel_gap_first_50_steps": float(gap_p[:50].max()),
               "max_rel_gap_all_steps": float(gap_p.max()),
               "last_batch_embedding_rel_diff": float(np.linalg.norm(e32p - e32) / np.linalg.norm(e32))},
           "final_loss": {"f16x3": float(l16[-10:].mean()), "f32": float(l32[-10:].mean())},
           "train_accuracy_last_20_steps": {"f16x3": a16, "f32": a32},
           "last_batch_embedding_rel_diff": float(np.linalg.norm(e16 - e32) / np.linalg.norm(e32))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
