"""Generates tests/golden/pooling_golden.npz: statistics pooling (model/pooling.py:9-34) pinned by the reference's own NumPy
code.  The reference ships no oracle for statistics pooling itself, but its self-attention oracle
model/test_utils.py:compute_self_attention (:320-372) degenerates to it exactly: with `query = 0` every score is 0, the
softmax over time is uniform (1/T), and the function returns concat(mean_t x, sqrt(mean_t (x - mean)^2 + 1e-12)) - the
two-pass biased variance of pooling.py:23-30.  The only difference from the graph is the `+ 1e-12` inside the sqrt where
the graph masks variances <= 1e-12 (pooling.py:28-29): < 5e-7 relative once the variance is above 1e-6, and both give
1e-6 for a constant chunk.

Run in the build container only (needs /root/reference):  python tests/golden/make_pooling_golden.py
The reference function is executed under py2 integer-division semantics for its reshape arguments exactly as
make_attention_golden.py does (source read at run time, nothing of it stored here).  Inputs: pooling_cases.py (seeded; they include
the adversarial rows of the reference's pooling self-test, model/pooling.py:503-506).
"""
import inspect
import os
import sys
import types

import numpy as np

sys.path.insert(0, "/root/reference")
from model import test_utils  # noqa: E402

src = inspect.getsource(test_utils.compute_self_attention).replace("/n_heads", "//n_heads")
ns = dict(vars(test_utils))
exec(compile(src, "<compute_self_attention under py2 division>", "exec"), ns)
compute_self_attention = ns["compute_self_attention"]

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pooling_cases import pooling_cases  # noqa: E402

params = types.SimpleNamespace(att_split_key=False, att_use_scale=True, att_penalty_term=0.0)
out = {}
i = 0
for x in pooling_cases():
    b, t, c = x.shape
    key = np.ones((b, t, 4))
    query = np.zeros((1, 4))                          # every score 0: uniform attention over time
    pooled, _ = compute_self_attention(x.astype(np.float64), key, query, params)
    out["pool_%d" % i] = pooled
    out["x_sum_%d" % i] = np.array([x.astype(np.float64).sum(), np.abs(x.astype(np.float64)).sum()])
    i += 1
out["num_cases"] = np.array(i)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pooling_golden.npz")
np.savez_compressed(path, **out)
print("wrote %s (%d cases)" % (path, i))
