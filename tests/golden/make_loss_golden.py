#!/usr/bin/env python3
"""Generate tests/golden/loss_golden.npz from the REFERENCE's own NumPy loss
oracles (model/test_utils.py:157-318: compute_asoftmax / compute_amsoftmax /
compute_arcsoftmax).  Runs only in the build container (needs /root/reference);
the resulting .npz holds inputs + expected outputs only and is what travels.

Case design follows the reference self-tests (model/tdnn.py:254-343): random
embeddings with the adversarial rows  row0 = w[:,y0]+1e-5 (theta~0),
row1 = -w[:,y1]+1e-5 (theta~pi), row2 *= 1e-4 (tiny norm), row3 *= 10 (large
norm); feature_norm in {False, True} with feature_scaling_factor 0.1; plus the
margin values and lambda schedules of the shipped configs
(egs/voxceleb/v1/nnet_conf/tdnn_{a,am,arc}softmax_*.json).

Usage:  python tests/golden/make_loss_golden.py [/root/reference]
"""
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, REF)
from model.test_utils import compute_amsoftmax, compute_arcsoftmax, compute_asoftmax  # noqa: E402


class P(object):
    pass


def main():
    rs = np.random.RandomState(20240611)
    n, dim, ncls = 12, 32, 10
    cases = []
    # (kind, m, lambda_min, lambda_base, lambda_gamma, lambda_power)
    grid = []
    for m in (1, 2, 4):
        grid.append(("asoftmax", m, 10.0, 1000.0, 1e-5, 5.0))      # shipped schedule
        grid.append(("asoftmax", m, 0.0, 1000.0, 1e-4, 5.0))
    for m in (0.0, 0.1, 0.15, 0.2, 0.25, 0.3, 0.5):
        grid.append(("additive_margin_softmax", m, 0.0, 1000.0, 1e-4, 5.0))
    for m in (0.0, 0.1, 0.2, 0.25, 0.3, 0.35, 0.5):
        grid.append(("additive_angular_margin_softmax", m, 0.0, 1000.0, 1e-5, 5.0))

    for kind, m, lmin, lbase, lgamma, lpower in grid:
        for step in (1, 1000, 1000000):
            for fnorm in (False, True):
                w = (rs.rand(dim, ncls).astype(np.float32) - 0.5) * 0.6
                labels = rs.randint(0, ncls, n).astype(np.int32)
                emb = rs.rand(n, dim).astype(np.float32)
                emb[0, :] = w[:, labels[0]] + 1e-5
                emb[1, :] = -1 * w[:, labels[1]] + 1e-5
                emb[2, :] = 1e-4 * emb[2, :]
                if step == 1:
                    # the reference applies the x10 row only at global_step=1 (tdnn.py:254,276):
                    # "The norm cannot be too large, since the precision in softmax and log will
                    # screw things up" - its oracle adds 1e-16 inside the log (test_utils.py:181),
                    # which saturates once the margin is fully annealed in.
                    emb[3, :] = 10 * emb[3, :]
                p = P()
                p.feature_norm = fnorm
                p.feature_scaling_factor = 0.1
                p.global_step = step
                if kind == "asoftmax":
                    p.asoftmax_m = m
                    p.asoftmax_lambda_min, p.asoftmax_lambda_base = lmin, lbase
                    p.asoftmax_lambda_gamma, p.asoftmax_lambda_power = lgamma, lpower
                    loss = compute_asoftmax(emb.astype(np.float64), labels, p, w.astype(np.float64))
                elif kind == "additive_margin_softmax":
                    p.amsoftmax_m = m
                    p.amsoftmax_lambda_min, p.amsoftmax_lambda_base = lmin, lbase
                    p.amsoftmax_lambda_gamma, p.amsoftmax_lambda_power = lgamma, lpower
                    loss = compute_amsoftmax(emb.astype(np.float64), labels, p, w.astype(np.float64))
                else:
                    p.arcsoftmax_m = m
                    p.arcsoftmax_lambda_min, p.arcsoftmax_lambda_base = lmin, lbase
                    p.arcsoftmax_lambda_gamma, p.arcsoftmax_lambda_power = lgamma, lpower
                    loss = compute_arcsoftmax(emb.astype(np.float64), labels, p, w.astype(np.float64))
                cases.append(dict(kind=kind, m=m, lmin=lmin, lbase=lbase, lgamma=lgamma, lpower=lpower,
                                  step=step, fnorm=fnorm, w=w, labels=labels, emb=emb, loss=float(loss)))

    out = {
        "kind": np.array([c["kind"] for c in cases]),
        "m": np.array([c["m"] for c in cases], np.float64),
        "sched": np.array([[c["lmin"], c["lbase"], c["lgamma"], c["lpower"]] for c in cases], np.float64),
        "step": np.array([c["step"] for c in cases], np.int64),
        "feature_norm": np.array([c["fnorm"] for c in cases], np.bool_),
        "w": np.stack([c["w"] for c in cases]),
        "labels": np.stack([c["labels"] for c in cases]),
        "emb": np.stack([c["emb"] for c in cases]),
        "loss": np.array([c["loss"] for c in cases], np.float64),
    }
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loss_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote %s: %d cases" % (dst, len(cases)))


if __name__ == "__main__":
    main()
