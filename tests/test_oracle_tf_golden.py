"""oracle/xvector_oracle.py against vectors dumped from the reference's own TensorFlow graph (tests/golden/make_tf_golden.py):
endpoints, loss, every gradient, one and two optimiser steps (SGD / momentum / Nesterov / Adam), the BN moving statistics and the
inference-mode embedding.  The generator needs TensorFlow 1.x and the reference checkout, neither of which exists in the build
container - until someone runs it and commits tests/golden/tf_golden.npz this module SKIPS and the oracle's conv / BN / dense /
pooling / optimiser semantics stay "parity unpinned vs TF1" (oracle header)."""
import json
import os

import numpy as np
import pytest

from oracle import xvector_oracle as O
from tests.oracle_params import oracle_kw_from_params

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_golden.npz")
pytestmark = pytest.mark.skipif(not os.path.isfile(GOLDEN), reason="tests/golden/tf_golden.npz absent: run tests/golden/make_tf_golden.py on a box "
                                "with TensorFlow 1.x + the reference checkout (cannot run in the build container)")


def _cases():
    if not os.path.isfile(GOLDEN):
        return ["<no golden file>"]
    return json.loads(str(np.load(GOLDEN)["__cases__"]))


def _close(got, want, tol, name):
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64).reshape(want.shape)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
    assert err < tol, "%s: max error %.3g of the tensor maximum (tolerance %.1g)" % (name, err, tol)


@pytest.mark.parametrize("case", _cases())
def test_oracle_matches_the_tensorflow_graph(case):
    data = np.load(GOLDEN)
    sub = lambda kind: {k.split("/", 2)[2]: data[k] for k in data.files if k.startswith("%s/%s/" % (case, kind))}      # noqa: E731
    d = json.loads(str(data[case + "/params"]))
    d.setdefault("pooling_type", "statistics_pooling")
    x, y = data[case + "/x"].astype(np.float64), data[case + "/y"]
    lr, step = float(data[case + "/lr"]), int(data[case + "/step"])
    var0, var1, var2 = sub("var0"), sub("var1"), sub("var2")
    cfg = O.Config(feat_dim=x.shape[2], num_speakers=int(var0["softmax/output/kernel"].shape[-1]), **oracle_kw_from_params(d))
    shapes = O.variable_shapes(cfg)
    assert set(shapes) == set(var0), "variable names differ: %s" % sorted(set(shapes) ^ set(var0))
    V = O.OrderedDict((k, var0[k].astype(np.float64).reshape(shapes[k])) for k in shapes)
    newV, state, info = O.train_step(V, {}, cfg, x, y, lr, step)
    # TensorFlow computed in fp32: 2e-5 on forward values, 1e-4 on gradients (their own rounding noise, a few hundred rows per sum)
    _close(info["raw_loss"], data[case + "/raw_loss"], 2e-5, "raw_loss")
    _close(info["total_loss"], data[case + "/total_loss"], 2e-5, "total_loss")
    for k, want in sub("ep").items():
        if k in info["endpoints"]:
            _close(info["endpoints"][k], want, 5e-5, "endpoint " + k)
    grads = sub("grad")
    assert set(grads) == {k for k in shapes if O.is_trainable(k)}
    for k, want in grads.items():
        _close(info["grads"][k], want, 2e-4, "gradient " + k)
    for k, want in var1.items():
        _close(newV[k], want, 2e-5, "after one step: " + k)
    newV2, _, _ = O.train_step(newV, state, cfg, x, y, lr, step + 1)
    for k, want in var2.items():
        _close(newV2[k], want, 5e-5, "after two steps: " + k)
    # inference mode on the moving statistics (this is where the fused-BN Bessel switch of TF_SEMANTICS, SURVEY N4, would show)
    _, ep, _ = O.tdnn_forward(newV2, x, cfg, False)
    _close(ep[d.get("embedding_node", "tdnn6_dense")], data[case + "/emb_after"], 5e-5, "inference embedding")
