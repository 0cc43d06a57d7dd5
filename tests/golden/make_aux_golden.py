"""Generates tests/golden/aux_golden.npz from the reference's NumPy oracles for the auxiliary losses
(model/test_utils.py:855-884 compute_ring_loss / compute_mhe - what model/loss.py:1040-1087 checks the graph against).
Run in the build container only (needs /root/reference):  python tests/golden/make_aux_golden.py"""
import os
import sys
import types

import numpy as np

sys.path.insert(0, "/root/reference")
from model import test_utils  # noqa: E402

rs = np.random.RandomState(20261003)
out = {}
i = 0
for (b, d, n, r, lam) in ((16, 64, 20, 20.0, 0.01), (5, 512, 37, 1.5, 0.1), (128, 32, 300, 10.0, 0.01)):
    feats = rs.randn(b, d) * rs.choice([0.1, 1.0, 10.0])
    w = rs.randn(d, n)
    labels = rs.randint(0, n, b)
    p = types.SimpleNamespace(ring_loss_lambda=lam, mhe_lambda=lam)
    out["features_%d" % i], out["w_%d" % i], out["labels_%d" % i] = feats, w, labels
    out["r_%d" % i], out["lambda_%d" % i] = np.array(r), np.array(lam)
    out["ring_%d" % i] = np.array(test_utils.compute_ring_loss(feats, p, r))
    out["mhe_%d" % i] = np.array(test_utils.compute_mhe(labels, p, w.copy()))      # the reference normalises w in place: pass a copy
    i += 1
out["num_cases"] = np.array(i)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aux_golden.npz")
np.savez_compressed(path, **out)
print("wrote %s (%d cases)" % (path, i))
