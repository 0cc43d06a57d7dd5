"""Generates tests/golden/attention_golden.npz from the reference's own NumPy oracle model/test_utils.py:compute_self_attention
(the function the reference's pooling self-test checks the TF graph against, model/pooling.py:515-560).

Run in the build container only (needs /root/reference):  python tests/golden/make_attention_golden.py
The reference function was written for Python 2: its reshape arguments `value_dim/n_heads` are floats under Python 3 and
numpy refuses them.  The function is therefore executed with py2 integer-division semantics for exactly those
expressions (its source is read at run time, `X/n_heads` -> `X//n_heads`, nothing of it is stored here).
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

out = {}
rs = np.random.RandomState(20261002)
i = 0
for use_scale in (True, False):
    for (b, l, vd, kd) in ((4, 13, 24, 16), (3, 40, 10, 6), (2, 1, 8, 8)):
        value = rs.rand(b, l, vd)                    # post-ReLU like
        if b > 2:
            value[1] = 0.7                           # constant chunk: weighted variance 0
            value[2] *= 100.0
        key = np.tanh(rs.randn(b, l, kd) * 2)
        query = rs.randn(1, kd) * 0.1
        params = types.SimpleNamespace(att_split_key=False, att_use_scale=use_scale, att_penalty_term=0.0)
        att, penalty = compute_self_attention(value, key, query, params)
        out["value_%d" % i], out["key_%d" % i], out["query_%d" % i] = value, key, query
        out["att_%d" % i], out["use_scale_%d" % i] = att, np.array(use_scale)
        i += 1
out["num_cases"] = np.array(i)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "attention_golden.npz")
np.savez_compressed(path, **out)
print("wrote %s (%d cases)" % (path, i))
