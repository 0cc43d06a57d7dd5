"""Oracle parity at the BASELINE shapes, full size, on a real MI355X (S1-S4, and configs[4] as S5).

Every other engine test compares with the float64 oracle at a few chunks x a few dozen frames; the reductions over
~25 000 (chunk, frame) rows per BatchNorm / weight gradient, the 7 351-column loss head and the per-tensor fp16 plane
scales of the split-precision path only meet their worst case at the shapes the benchmark runs.  Here the engine does one
optimiser step at

  S1  128 chunks x 200 frames x 30-dim, statistics pooling, AM-Softmax m = 0.2        (BASELINE configs[1])
  S2  128 x 400, A-Softmax m = 4                                                      (chunk length / loss of configs[4])
  S3  64 chunks x one length drawn from U{200..400} (the shipped sampler), ArcFace    (configs[2])
  S4  S1 + the self-attention head of nnet_conf/*_tdnn4_att.json (1500/1500 keys)     (configs[3])
  S5  128 x 400, the extended 10-layer frame stack, A-Softmax m = 4                   (configs[4]; no reference counterpart)

with 7 351 speakers, in both precisions, and is compared with the oracle's float64 step from the same fp32 variables,
features and labels: loss, regulariser, 15-19 endpoints incl. the `tdnn6_dense` embedding and the logits, every gradient
(oracle evaluated on the GPU's ReLU on/off pattern, which is first asserted to differ from the float64 pattern only at
rounding-level pre-activations - see tests/test_gpu_engine.py::oracle_step_with_gpu_relu_pattern), and every variable after
the update.  Tolerances are those of the small-shape tests: endpoints 5e-5 (north_star: embeddings within 1e-4), gradients
1e-4 of each tensor's largest entry, updated variables 2e-5.

The oracle forward of a shape (tens of seconds of NumPy on the host) is computed once and shared by the two precisions.
"""
import numpy as np
import pytest

from tests.test_gpu_engine import (_make, oracle_forward, gpu_relu_pattern, oracle_backward_and_update,
                                   compare_step_with_oracle)

pytestmark = pytest.mark.gpu

NSPK = 7351
SHAPES = {
    "S1": dict(B=128, T=200, step=100000,      # lambda = 1000 * 11^-5: the margin branch carries 99 % of the logit
               kw=dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True)),
    "S2": dict(B=128, T=400, step=5000,
               kw=dict(loss_func="asoftmax", margin_m=4, lambda_min=10, lambda_gamma=1e-5, last_layer_linear=True)),
    "S3": dict(B=64, T=None, step=300000,      # T: one draw of the shipped sampler's U{min_segment_len..max_segment_len}
               kw=dict(loss_func="additive_angular_margin_softmax", margin_m=0.3, lambda_gamma=1e-5, last_layer_linear=True)),
    "S4": dict(B=128, T=200, step=0,
               kw=dict(loss_func="softmax", pooling_type="self_attention", att_key_num_nodes=(1500, 1500),
                       att_key_network_type=3, att_use_scale=True)),
    # BASELINE configs[4] "Deep TDNN, extended context, 10 layers, A-Softmax, 400-frame chunks" (no reference counterpart):
    # the table `bench.py --extended` times
    "S5": dict(B=128, T=400, step=20000,
               kw=dict(loss_func="asoftmax", margin_m=4, lambda_min=10, lambda_gamma=1e-5, last_layer_linear=True,
                       frame_layers=((5, 512), (1, 512), (3, 512), (1, 512), (3, 512), (1, 512), (3, 512), (1, 512), (1, 512), (1, 1500)))),
}
_ORACLE = {"shape": None}      # one shape's float64 forward at a time (S2's caches are ~5 GB)


def _inputs(name):
    c = SHAPES[name]
    rs = np.random.RandomState({"S1": 11, "S2": 12, "S3": 13, "S4": 14, "S5": 15}[name])
    T = c["T"] if c["T"] else int(rs.randint(200, 401))
    x = rs.randn(c["B"], T, 30).astype(np.float32)
    labels = rs.randint(0, NSPK, c["B"]).astype(np.int32)
    return c["B"], T, x, labels


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("name", list(SHAPES))
def test_full_size_step_matches_oracle(name, precision, monkeypatch):
    monkeypatch.setenv("XV_PRECISION", precision)
    c = SHAPES[name]
    B, T, x, labels = _inputs(name)
    eng, cfg_o, V = _make(c["kw"], B, T, N=NSPK, seed=3)
    lr = 0.05
    eng.forward(x, True)
    eng.loss(labels, c["step"], True)
    eng.backward(-1)
    if _ORACLE["shape"] != name:
        _ORACLE.clear()
        _ORACLE.update(shape=name, fwd=oracle_forward(V, cfg_o, x.astype(np.float64)), V=V)
    else:       # same fp32 variables on both runs (set_variables -> get_variables of identically seeded values)
        assert all(np.array_equal(V[k], _ORACLE["V"][k]) for k in V)
    fwd = _ORACLE["fwd"]
    ep_gpu = gpu_relu_pattern(eng, fwd)
    newV, _, info = oracle_backward_and_update(V, cfg_o, fwd, ep_gpu, labels, lr, c["step"], {})
    report = {}
    try:
        compare_step_with_oracle(eng, cfg_o, newV, info, lr, report=report)
    finally:
        worst = sorted(report.items(), key=lambda kv: -kv[1])[:6]
        print("\n[%s %s B=%d T=%d] loss %.6f  worst: %s" % (name, precision, B, T, info["raw_loss"],
                                                          ", ".join("%s %.2e" % kv for kv in worst)))
    eng.close()
    if precision == "f16x3":      # last user of this shape's oracle forward
        _ORACLE.clear()
        _ORACLE["shape"] = None
