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
needs_golden = pytest.mark.skipif(not os.path.isfile(GOLDEN), reason="tests/golden/tf_golden.npz absent: run tests/golden/make_tf_golden.py on a box "
                                  "with TensorFlow 1.x + the reference checkout (cannot run in the build container)")


def _cases():
    if not os.path.isfile(GOLDEN):
        return ["<no golden file>"]
    return json.loads(str(np.load(GOLDEN)["__cases__"]))


def _close(got, want, tol, name, floor=1e-30, slack=None):
    """Maximum error as a fraction of the tensor's maximum; `floor` bounds that scale from below for tensors that are ZERO analytically
    (the gradient of a bias in front of a BatchNorm: TensorFlow's fp32 gives rounding noise there, the fp64 oracle 1e-17); `slack` is a
    per-element allowance subtracted from the error first (Adam's ill-conditioned elements, see check_case)."""
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64).reshape(want.shape)
    diff = np.abs(got - want)
    if slack is not None:
        diff = np.maximum(diff - np.asarray(slack, np.float64).reshape(want.shape), 0.0)
    err = diff.max() / max(np.abs(want).max(), floor)
    assert err < tol, "%s: max error %.3g of the tensor maximum (tolerance %.1g)" % (name, err, tol)


def _no_kinks(endpoints, case, which):
    """A ReLU input within fp32 rounding of zero makes the comparison ill-posed (its sign, and every gradient upstream of it, is decided by
    rounding): the generator redraws such batches (make_tf_golden.KINK = 2e-6 on its fp32 values), so half that on the fp64 ones here."""
    n = 0
    for k in endpoints:
        if k.endswith("_relu"):
            pre = [k[:-5] + s for s in ("_bn", "_dense", "_conv") if k[:-5] + s in endpoints]
            n += int((np.abs(endpoints[pre[0]]) < 1e-6).sum()) if pre else 0
    assert n == 0, ("case %s, %s step: %d ReLU input(s) within 1e-6 of zero - the comparison is ill-posed, not wrong; regenerate the golden "
                    "file (the generator's redraw rule should have caught it: raise make_tf_golden.KINK)" % (case, which, n))


@needs_golden
@pytest.mark.parametrize("case", _cases())
def test_oracle_matches_the_tensorflow_graph(case):
    check_case(np.load(GOLDEN), case)


def check_case(data, case):
    """One case of a golden file in make_tf_golden.py's layout (an open .npz or a dict of arrays) against the oracle."""
    files = data.files if hasattr(data, "files") else list(data)
    sub = lambda kind: {k.split("/", 2)[2]: data[k] for k in files if k.startswith("%s/%s/" % (case, kind))}      # noqa: E731
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
    _no_kinks(info["endpoints"], case, "first")
    # TensorFlow computed in fp32: 2e-5 on forward values, 1e-4 on gradients (their own rounding noise, a few hundred rows per sum)
    _close(info["raw_loss"], data[case + "/raw_loss"], 2e-5, "raw_loss")
    _close(info["total_loss"], data[case + "/total_loss"], 2e-5, "total_loss")
    for k, want in sub("ep").items():
        if k in info["endpoints"]:
            _close(info["endpoints"][k], want, 5e-5, "endpoint " + k)
    grads = sub("grad")
    assert set(grads) == {k for k in shapes if O.is_trainable(k)}
    for k, want in grads.items():
        _close(info["grads"][k], want, 2e-4, "gradient " + k, floor=1e-2)      # GTOL, GFLOOR below
    # Adam divides the step by |g| + 1e-8 (its first step is lr * g / (|g| + eps)): where a gradient element is within the gradient tolerance
    # of zero the update is anything in +-lr whatever the arithmetic (a sign flip moves it by 2 lr), so those elements get that much slack (a handful per tensor); SGD and
    # momentum are linear in g and get none
    GTOL, GFLOOR = 2e-4, 1e-2
    slack = {}
    if d.get("optimizer") == "adam":
        for k, g in info["grads"].items():
            g = np.abs(np.asarray(g, np.float64).reshape(shapes[k]))
            slack[k] = 2.0 * lr * np.minimum(1.0, GTOL * max(g.max(), GFLOOR) / (g + 1e-8))
    for k, want in var1.items():
        _close(newV[k], want, 2e-5, "after one step: " + k, slack=slack.get(k))
    # the second step starts from the GOLDEN variables after the first (with the oracle's own slot state: TensorFlow's slots are not dumped),
    # so that one step's allowance does not leak into the next
    V1 = O.OrderedDict((k, var1[k].astype(np.float64).reshape(shapes[k])) for k in shapes)
    newV2, _, info2 = O.train_step(V1, state, cfg, x, y, lr, step + 1)
    _no_kinks(info2["endpoints"], case, "second")
    for k, want in var2.items():
        s2 = None
        if k in slack:
            g = np.abs(np.asarray(info2["grads"][k], np.float64).reshape(shapes[k]))
            s2 = slack[k] + 2.0 * lr * np.minimum(1.0, GTOL * max(g.max(), GFLOOR) / (g + 1e-8))
        _close(newV2[k], want, 5e-5, "after two steps: " + k, slack=s2)
    # inference mode on the moving statistics (this is where the fused-BN Bessel switch of TF_SEMANTICS, SURVEY N4, would show), from the
    # GOLDEN variables after the two steps: the inference arithmetic alone, not the optimiser's conditioning again
    V2 = O.OrderedDict((k, var2[k].astype(np.float64).reshape(shapes[k])) for k in shapes)
    _, ep, _ = O.tdnn_forward(V2, x, cfg, False)
    _close(ep[d.get("embedding_node", "tdnn6_dense")], data[case + "/emb_after"], 5e-5, "inference embedding")


def test_generator_cases_and_layout_run_through_the_checker():
    """NOT a parity pin (the numbers below come from the oracle itself): what it proves is that the generator's case list and key layout
    and this module's checker agree before anyone spends a TensorFlow box on them - every config of make_tf_golden.CASES builds an
    oracle Config, its variable names are the ones the checker expects, and check_case() reads every key the generator writes
    (make_tf_golden.run_case's `key(...)` scheme, restated here without TensorFlow)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_tf_golden", os.path.join(os.path.dirname(GOLDEN), "make_tf_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)                          # TensorFlow is only imported inside gen.main()
    out = {"__cases__": np.array(json.dumps([c[0] for c in gen.CASES]))}
    for name, cfg_d in gen.CASES:
        d = dict(cfg_d)
        cfg = O.Config(feat_dim=gen.D, num_speakers=gen.N, **oracle_kw_from_params(d))
        for attempt in range(200):
            arrays, on_kink = _oracle_case_in_generator_layout(gen, name, d, cfg, attempt)
            if on_kink == 0:
                break
        assert on_kink == 0, "no kink-free batch for %s" % name
        out.update(arrays)
    for name, _ in gen.CASES:
        check_case(out, name)


def _oracle_case_in_generator_layout(gen, name, d, cfg, attempt):
    """make_tf_golden.run_case with the oracle (in float32 arithmetic) where the TensorFlow session is: same keys, same redraw rule."""
    out = {}
    rs = gen.case_rng(name, attempt)
    V = O.init_variables(cfg, seed=1, dtype=np.float32)
    for k in V:
        if k.endswith(("gamma", "beta", "bias")):
            V[k] = (V[k] + 0.1 * rs.randn(*V[k].shape)).astype(np.float32)
        elif k.endswith("moving_mean"):
            V[k] = (0.2 * rs.randn(*V[k].shape)).astype(np.float32)
        elif k.endswith("moving_variance"):
            V[k] = (0.5 + rs.rand(*V[k].shape)).astype(np.float32)
    x, y = rs.randn(gen.B, gen.T, gen.D).astype(np.float32), rs.randint(0, gen.N, gen.B).astype(np.int32)
    key = lambda *p: "/".join((name,) + p)      # noqa: E731
    out[key("params")] = np.array(json.dumps(d))
    out[key("x")], out[key("y")], out[key("lr")], out[key("step")] = x, y, np.float64(gen.LR), np.int64(gen.STEP)
    out[key("attempt")] = np.int64(attempt)
    V1, state, info = O.train_step(V, {}, cfg, x, y, gen.LR, gen.STEP)          # float32 arithmetic, like the TensorFlow graph
    V2, _, info2 = O.train_step(V1, state, cfg, x, y, gen.LR, gen.STEP + 1)
    on_kink = gen.kinks(info["endpoints"]) + gen.kinks(info2["endpoints"])
    for k in V:
        out[key("var0", k)], out[key("var1", k)], out[key("var2", k)] = (np.asarray(v, np.float32) for v in (V[k], V1[k], V2[k]))
    for k, v in info["endpoints"].items():
        if np.asarray(v).dtype.kind == "f":
            out[key("ep", k)] = np.asarray(v, np.float32)
    out[key("raw_loss")], out[key("total_loss")] = np.float64(info["raw_loss"]), np.float64(info["total_loss"])
    for k, g in info["grads"].items():
        out[key("grad", k)] = np.asarray(g, np.float32)
    _, ep, _ = O.tdnn_forward(V2, x, cfg, False)
    out[key("emb_after")] = np.asarray(ep[d["embedding_node"]], np.float32)
    return out, on_kink
