"""Whole-graph parity on a real MI355X: the native engine (one C call per phase) against the
CPU oracle's train_step on the same seeded variables / features / labels.

Tolerances: endpoints (incl. the tdnn6_dense embedding) within 5e-5 relative of the float64
oracle (north_star asks 1e-4 for embeddings; measured 1e-6..2e-5); gradients within 1e-4 of the
largest entry of each variable, evaluated on the GPU's ReLU on/off pattern (see
oracle_step_with_gpu_relu_pattern); updated variables within 2e-5."""
import numpy as np
import pytest
import torch

from oracle import xvector_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f32", "f16x3"], autouse=True)
def xv_precision(request, monkeypatch):
    """Every test of this module runs twice: fp32-input MFMA and the split-precision (f16x3) path,
    against the same oracle and the same tolerances."""
    monkeypatch.setenv("XV_PRECISION", request.param)
    return request.param


def rel_err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)


CASES = [
    dict(loss_func="softmax"),
    dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True),
    dict(loss_func="additive_angular_margin_softmax", margin_m=0.25, lambda_gamma=1e-2, last_layer_linear=True),
    dict(loss_func="asoftmax", margin_m=4, lambda_min=10, lambda_gamma=1e-5, last_layer_linear=True),
    dict(loss_func="asoftmax", margin_m=2, lambda_min=0, lambda_gamma=1.0, last_layer_linear=True, last_layer_no_bn=True),
    dict(loss_func="additive_margin_softmax", margin_m=0.2, feature_norm=True, feature_scaling_factor=30.0,
         last_layer_linear=True),
    dict(loss_func="softmax", optimizer="momentum", momentum=0.9, use_nesterov=True),
    dict(loss_func="softmax", optimizer="adam"),
    # self-attention pooling (pooling.py:37-192), the shipped form of nnet_conf/*_tdnn4_att.json with smaller key layers,
    # and its affine-key / unscaled variant
    dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, pooling_type="self_attention",
         att_key_num_nodes=(300, 200)),
    dict(loss_func="softmax", pooling_type="self_attention", att_key_num_nodes=(64, 48), att_key_network_type=0, att_use_scale=False),
    dict(loss_func="softmax", pooling_type="self_attention", att_key_num_nodes=(64, 48), att_key_network_type=1),      # fisher *_att_2.json
    dict(loss_func="softmax", pooling_type="self_attention", att_key_num_nodes=(64, 48), att_key_network_type=2),      # fisher *_att_3.json
    # auxiliary losses (loss.py:985-1036) as in nnet_conf/*_r0.01.json and *_mhe0.01.json, stronger weights to make them count
    dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, aux_loss_func=("ring_loss", "mhe_loss"),
         ring_loss_init=3.0, ring_loss_lambda=0.05, mhe_lambda=0.05),
]


def _make(kw, B, T, N=37, P=1500, seed=0, max_batch=None, max_frames=None, D=30, L=512, engine_cfg=None):
    from tf_kaldi_speaker_amd import engine as E
    kw = dict(kw)
    cfg_o = O.Config(feat_dim=D, num_speakers=N, num_nodes_pooling_layer=P, num_nodes_last_layer=L, **kw)
    ekw = dict(kw)
    ekw.pop("loss_func", None)
    c = E.make_config(D, N, loss_func=kw.get("loss_func", "softmax"), num_nodes_pooling_layer=P, num_nodes_last_layer=L,
                      max_batch=max_batch or B, max_frames=max_frames or T, **ekw)
    eng = E.Engine(engine_cfg if engine_cfg is not None else c)       # engine_cfg: built by the product's own Params mapping
    V = O.init_variables(cfg_o, seed=seed, dtype=np.float64)
    rs = np.random.RandomState(seed + 100)
    for k in V:   # move BN parameters / biases / moving stats off their trivial init
        if k.endswith(("gamma", "beta", "bias")):
            V[k] = V[k] + 0.1 * rs.randn(*V[k].shape)
        if k.endswith("alpha"):
            V[k] = V[k] + 0.3 * rs.rand(*V[k].shape)
        if k.endswith("moving_mean"):
            V[k] = 0.2 * rs.randn(*V[k].shape)
        if k.endswith("moving_variance"):
            V[k] = 0.5 + rs.rand(*V[k].shape)
    eng.set_variables({k: v.astype(np.float32) for k, v in V.items()})
    V = OrderedDictF64(eng.get_variables())   # oracle starts from the exact fp32 values
    return eng, cfg_o, V


def OrderedDictF64(d):
    from collections import OrderedDict
    return OrderedDict((k, v.astype(np.float64)) for k, v in d.items())


def test_variable_table_matches_reference_names():
    eng, cfg_o, V = _make(dict(loss_func="softmax"), 2, 20)
    assert list(eng.table.keys())[:4] == ["tdnn/tdnn1_conv/kernel", "tdnn/tdnn1_conv/bias", "tdnn/tdnn1_bn/gamma",
                                          "tdnn/tdnn1_bn/beta"]
    shapes = O.variable_shapes(cfg_o)
    assert set(eng.table) == set(shapes)
    for k, (shape, off, tr) in eng.table.items():
        assert tuple(shape) == tuple(shapes[k]) and off % 4 == 0 and tr == O.is_trainable(k)
    eng.close()


RELU_LAYERS = tuple("tdnn%d" % i for i in range(1, 15)) + ("att_key0", "att_key1")


def oracle_forward(V, cfg_o, x):
    """The float64 forward half of an oracle step: reusable for several GPU runs from the same variables and inputs
    (the full-size tests compare both precisions against one oracle forward)."""
    bn_new = {}
    feats, ep, caches = O.tdnn_forward(V, x, cfg_o, True, bn_new)
    return {"bn_new": bn_new, "feats": feats, "ep": ep, "caches": caches}


def gpu_relu_pattern(eng, fwd, max_flip_fraction=1e-4, cfg_o=None):
    """Endpoints of the oracle forward with every ReLU output replaced by the GPU's (see oracle_step_with_gpu_relu_pattern).
    Asserts that the two on/off patterns differ only at rounding-level pre-activations.
    prelu / lrelu (cfg_o.network_relu_type): the oracle differentiates those on the SIGN OF THE ACTIVATION'S INPUT, so the rounding-level
    entries whose sign the GPU sees differently get the GPU's sign in fwd["caches"] (patched in place; positive slopes assumed)."""
    ep = fwd["ep"]
    ep_gpu = dict(ep)
    leaky = cfg_o is not None and cfg_o.network_relu_type != "relu"
    for prefix in RELU_LAYERS:
        key = prefix + "_relu"
        if key not in ep:
            continue
        got = eng.endpoint(key).cpu().numpy().reshape(ep[key].shape)
        flips = (got > 0) != (ep[key] > 0)
        assert flips.mean() < max_flip_fraction, (key, flips.mean())
        pre = ep[prefix + "_bn"] if prefix + "_bn" in ep else ep[prefix + "_dense"]       # att_key1 with a plain ReLU (type 1)
        assert np.all(np.abs(pre[flips]) < 2e-5 * max(1.0, np.abs(pre).max())), key
        ep_gpu[key] = got.astype(np.float64)
        if leaky and flips.any():
            x = fwd["caches"][prefix + "_act_in"].copy()
            x[flips] = np.where(got[flips] > 0, 1.0, -1.0) * np.abs(x[flips])
            fwd["caches"][prefix + "_act_in"] = x
    return ep_gpu


def oracle_backward_and_update(V, cfg_o, fwd, ep_for_backward, labels, lr, step, opt_state):
    """Loss, every gradient (incl. the regulariser) and the optimiser step of the oracle, with the ReLU masks of
    `ep_for_backward` (same code path as O.train_step)."""
    ep, caches, bn_new = fwd["ep"], fwd["caches"], fwd["bn_new"]
    raw_loss, logits, dfeat, Gl = O.loss_forward_backward(V, cfg_o, fwd["feats"], labels, step)
    reg, Gr = O.regularization(V, cfg_o)
    G = O.tdnn_backward(V, cfg_o, ep_for_backward, caches, dfeat)
    G.update(Gl)
    for k, g in Gr.items():
        G[k] = G[k] + g
    ep = dict(ep)
    ep["logits"] = logits
    newV, new_state = {}, {}
    t = opt_state.get("__t__", 0) + 1
    for name, p in V.items():
        if not O.is_trainable(name):
            newV[name] = bn_new.get(name, p)
            continue
        g = G[name].reshape(p.shape)
        if cfg_o.optimizer == "sgd":
            newV[name] = O.sgd_update(p, g, lr)
        elif cfg_o.optimizer == "momentum":
            newV[name], new_state[name] = O.momentum_update(p, g, opt_state.get(name, np.zeros_like(p)), lr, cfg_o.momentum,
                                                            cfg_o.use_nesterov)
        else:
            m, v = opt_state.get(name, (np.zeros_like(p), np.zeros_like(p)))
            pn, m, v = O.adam_update(p, g, m, v, t, lr)
            newV[name], new_state[name] = pn, (m, v)
    new_state["__t__"] = t
    info = {"raw_loss": raw_loss, "reg_loss": reg, "grads": G, "endpoints": ep}
    return newV, new_state, info


def oracle_step_with_gpu_relu_pattern(eng, V, cfg_o, x, labels, lr, step, opt_state, fwd=None):
    """Oracle train step whose ReLU on/off pattern is taken from the GPU forward.

    ReLU makes the gradient a discontinuous function of the forward values: a pre-activation
    that is -3e-6 in float64 and +4e-8 in fp32 flips one mask bit, and with only ~200 rows per
    BatchNorm in these unit-test shapes one flipped bit moves a per-channel gradient by ~1 %.
    That is rounding, not arithmetic, so the test (1) asserts that the GPU pattern differs from
    the float64 pattern only where the float64 pre-activation is within 2e-5 of zero and in
    fewer than 1e-4 of the positions, then (2) checks all gradients at 1e-4 against the oracle
    evaluated on the GPU's pattern."""
    fwd = fwd if fwd is not None else oracle_forward(V, cfg_o, x)
    return oracle_backward_and_update(V, cfg_o, fwd, gpu_relu_pattern(eng, fwd, cfg_o=cfg_o), labels, lr, step, opt_state)


@pytest.mark.parametrize("kw", CASES, ids=lambda d: "-".join(str(v).replace(" ", "") for v in d.values()))
def test_train_step_matches_oracle(kw):
    _check_train_step(kw, 6, 40)


# The segment-level layers run fused (GEMM + split-K sum + BatchNorm in one launch, xv_skinny.hip) for batches of <= 128 chunks - every
# other test of this module - and as separate launches beyond that or with XV_SEGMENT_FUSED=0: both forms against the oracle, for the
# configurations that take different branches there (BN / no BN / linear last layer, l2_scaling in between, prelu)
SEGMENT_PATH_CASES = [CASES[0], dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, feature_norm=True),
                      dict(loss_func="asoftmax", margin_m=2, lambda_min=5, lambda_gamma=1e-3, last_layer_no_bn=True),
                      dict(loss_func="softmax", network_relu_type="prelu")]


@pytest.mark.parametrize("kw", SEGMENT_PATH_CASES, ids=lambda d: "-".join(str(v).replace(" ", "") for v in d.values()))
def test_train_step_matches_oracle_unfused_segment_layers(kw):
    _check_train_step(kw, 130, 21, N=53)


def test_train_step_matches_oracle_under_the_alternate_switches(xv_precision):
    """XV_SEGMENT_FUSED=0 (separate launches for <= 128 chunks too) and XV_DZ_SLOTS=2 (the two-slot dz ring in fp32 mode): the library
    reads its switches once per process, so the train-step oracle comparisons of this module run again in a child process under them."""
    import os, subprocess, sys
    if xv_precision != "f32":
        pytest.skip("the child process runs both precisions")
    env = dict(os.environ, XV_SEGMENT_FUSED="0", XV_DZ_SLOTS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "test_train_step_matches_oracle and not alternate_switches"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


# feature / layer widths of the other shipped recipes and ragged everything: 23-dim MFCCs (egs/sre, egs/fisher: not a multiple
# of 4 - padded to 24 / 32 channels inside), 40-dim, pooling layers of 600 / 3000 nodes, a 256-node last layer, odd batch,
# frame and speaker counts (but at least 4 rows per BatchNorm: with 2 the normalised values are +-1/sqrt(1 + eps/var) and a
# rounding-level difference of two nearly equal pre-activations is amplified past any fixed tolerance)
ODD_DIMS = [
    dict(D=23, P=600, L=256, N=101, B=5, T=33, kw=dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True)),
    dict(D=40, P=3000, L=512, N=53, B=5, T=27, kw=dict(loss_func="softmax")),
    dict(D=30, P=600, L=128, N=20011, B=7, T=25, kw=dict(loss_func="additive_angular_margin_softmax", margin_m=0.2, last_layer_linear=True)),   # 626 column tiles of logits
    dict(D=23, P=1500, L=128, N=19, B=4, T=22, kw=dict(loss_func="asoftmax", margin_m=2, lambda_min=5, lambda_gamma=1e-3,
                                                         last_layer_linear=True, pooling_type="self_attention",
                                                         att_key_num_nodes=(100, 60))),
]


@pytest.mark.parametrize("c", ODD_DIMS, ids=lambda c: "D%d-P%d-L%d-N%d-B%d-T%d" % (c["D"], c["P"], c["L"], c["N"], c["B"], c["T"]))
def test_train_step_matches_oracle_odd_dimensions(c):
    _check_train_step(c["kw"], c["B"], c["T"], N=c["N"], P=c["P"], D=c["D"], L=c["L"])


# Extended frame-layer tables (BASELINE configs[4] "Deep TDNN, extended context, 10 layers"; no reference counterpart - model/tdnn.py
# hard-codes its five layers): the oracle and the engine build the stack from the same (context, width) table
EXTENDED = [
    dict(B=6, T=40, kw=dict(loss_func="asoftmax", margin_m=4, lambda_min=10, lambda_gamma=1e-5, last_layer_linear=True,
                            frame_layers=((5, 512), (1, 512), (3, 512), (1, 512), (3, 512), (1, 512), (3, 512), (1, 512), (1, 512), (1, 1500)))),
    dict(B=5, T=24, kw=dict(loss_func="softmax", frame_layers=((3, 64), (2, 128), (1, 1500)))),
    dict(B=6, T=36, kw=dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, pooling_type="self_attention",
                            att_key_num_nodes=(96, 64), frame_layers=((5, 256), (3, 512), (1, 384), (7, 512), (1, 128), (1, 1500)))),
    dict(B=6, T=60, kw=dict(loss_func="softmax", optimizer="momentum", momentum=0.9,
                            frame_layers=((5, 512), (5, 512), (7, 512), (1, 512), (1, 512), (3, 256), (1, 256), (3, 512), (1, 512), (1, 512), (1, 512),
                                          (1, 1500)))),
]


@pytest.mark.parametrize("c", EXTENDED, ids=lambda c: "%dlayers" % len(c["kw"]["frame_layers"]))
def test_train_step_matches_oracle_extended_frame_layers(c):
    _check_train_step(c["kw"], c["B"], c["T"])


def test_extended_table_is_validated():
    from tf_kaldi_speaker_amd import engine as E
    with pytest.raises(ValueError):
        E.make_config(30, 10, frame_layers=((5, 512), (1, 512)))                         # fewer than 3 layers
    with pytest.raises(ValueError):
        E.make_config(30, 10, frame_layers=((5, 512), (1, 512), (1, 512)))               # last width != num_nodes_pooling_layer
    with pytest.raises(Exception):
        E.Engine(E.make_config(30, 10, max_batch=2, max_frames=10, frame_layers=((5, 512), (5, 512), (5, 512), (1, 1500))))   # receptive field 13 > 10


# network_relu_type (tdnn.py:24-30, common.py:27-42; SURVEY.md N3 - no shipped config sets it): prelu with its trainable per-channel
# alpha variables, tf.nn.leaky_relu; statistics and attention pooling, a ReLU'd last layer with and without BatchNorm, an extended table
RELU_VARIANTS = [
    dict(loss_func="softmax", network_relu_type="lrelu"),
    dict(loss_func="softmax", network_relu_type="prelu"),
    dict(loss_func="softmax", network_relu_type="prelu", last_layer_no_bn=True),
    dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, network_relu_type="prelu", pooling_type="self_attention",
         att_key_num_nodes=(64, 48), att_key_network_type=3),
    dict(loss_func="softmax", network_relu_type="lrelu", pooling_type="self_attention", att_key_num_nodes=(64, 48), att_key_network_type=2),
    dict(loss_func="asoftmax", margin_m=2, lambda_min=5, lambda_gamma=1e-3, last_layer_linear=True, network_relu_type="prelu",
         frame_layers=((5, 128), (1, 256), (3, 128), (1, 1500))),
]


@pytest.mark.parametrize("kw", RELU_VARIANTS, ids=lambda d: d["network_relu_type"] + ("_att%d" % d["att_key_network_type"] if "pooling_type" in d else "")
                         + ("_nobn" if d.get("last_layer_no_bn") else "") + ("_ext" if "frame_layers" in d else ""))
def test_train_step_matches_oracle_prelu_lrelu(kw):
    _check_train_step(kw, 6, 40)


def test_prelu_with_fused_relu_key_layer_is_refused():
    from tf_kaldi_speaker_amd import engine as E
    with pytest.raises(Exception):
        E.Engine(E.make_config(30, 10, network_relu_type="prelu", pooling_type="self_attention", att_key_num_nodes=(64, 48), att_key_network_type=1,
                               max_batch=2, max_frames=30))


def _shipped_combinations():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shipped_config_switches.json")
    return json.load(open(path))


from tests.oracle_params import oracle_kw_from_params as _oracle_kw_from_params      # noqa: E402


@pytest.mark.parametrize("combo", _shipped_combinations(), ids=lambda c: c["example"].split("/", 1)[1].replace("/nnet_conf/", ":").replace(".json", ""))
def test_every_shipped_switch_combination_steps_like_the_oracle(combo, tmp_path):
    """One full parity step (loss, endpoints, every gradient, update) for each of the 59 distinct combinations of hot-path switches
    among the reference's 81 single-task nnet_conf/*.json (tests/golden/make_config_switches.py), with the engine configured
    by the product's own Params -> engine mapping."""
    import json
    from tf_kaldi_speaker_amd.misc.utils import Params
    from tf_kaldi_speaker_amd.model.tdnn import engine_config
    d = dict(combo["params"])
    f = tmp_path / "config.json"
    f.write_text(json.dumps(d))
    params = Params(str(f))
    B, T, N, D = 4, 24, 13, 30
    if d.get("clip_gradient", False):
        pytest.skip("clip_gradient is false in every shipped config; covered by test_gpu_ops")
    cfg = engine_config(params, D, N, d["loss_func"], B, T)
    kw = _oracle_kw_from_params(params.dict)
    _check_train_step(kw, B, T, N=N, P=params.dict["num_nodes_pooling_layer"], D=D, L=params.dict["num_nodes_last_layer"], engine_cfg=cfg)


def _check_train_step(kw, B, T, **dims):
    eng, cfg_o, V = _make(kw, B, T, **dims)
    rs = np.random.RandomState(42)
    x = rs.randn(B, T, cfg_o.feat_dim).astype(np.float32)
    labels = rs.randint(0, cfg_o.num_speakers, B).astype(np.int32)
    step, lr = 1234, 0.05

    eng.forward(x, True)
    eng.loss(labels, step, True)
    eng.backward(-1)
    newV, _, info = oracle_step_with_gpu_relu_pattern(eng, V, cfg_o, x.astype(np.float64), labels, lr, step, {})
    compare_step_with_oracle(eng, cfg_o, newV, info, lr)
    eng.close()


def compare_step_with_oracle(eng, cfg_o, newV, info, lr, endpoint_tol=5e-5, grad_tol=1e-4, var_tol=2e-5, report=None):
    """Loss values, endpoints, every gradient and (after eng.apply) every variable of the engine against the oracle step
    `info` / `newV`.  `report`: optional dict that receives the observed errors."""
    raw, reg = eng.losses()
    assert abs(raw - info["raw_loss"]) <= 2e-5 * abs(info["raw_loss"]) + 1e-6, (raw, info["raw_loss"])
    assert abs(reg - info["reg_loss"]) <= 2e-5 * abs(info["reg_loss"])
    fl = O.frame_layers(cfg_o)
    nf = len(fl)
    names = ["%s_%s" % (p, kind) for p, kind, _, _ in fl] + ["tdnn1_relu", "tdnn3_relu", "tdnn%d_bn" % nf, "tdnn%d_relu" % nf, "pooling",
                                                            "tdnn%d_dense" % (nf + 1), "tdnn%d_relu" % (nf + 1), "tdnn%d_dense" % (nf + 2), "output", "logits"]
    if cfg_o.pooling_type == "self_attention":
        names += ["att_key0_dense", "att_key0_relu", "att_key1_dense", "attention_weights"]
    for name in names:
        ref = info["endpoints"][name]
        got = eng.endpoint(name).cpu().numpy()
        # north_star: embeddings within 1e-4 relative; measured ~1e-6..2e-5
        err = rel_err(got, ref.reshape(got.shape))
        if report is not None:
            report["endpoint:" + name] = err
        assert err <= endpoint_tol, (name, err)
    grads = eng.get_gradients()
    for name, g in grads.items():
        ref = info["grads"][name].reshape(g.shape)
        if "att_key1_dense/bias" in name:
            # no BN behind it: a real gradient for the tanh key, 0 + rounding noise for the affine key (the frame gradients
            # of a chunk sum to zero through the softmax) - absolute criterion on the scale of the layer's kernel gradient
            scale = max(np.abs(ref).max(), 1e-2 * np.abs(info["grads"][name.replace("/bias", "/kernel")]).max())
            assert np.abs(g - ref).max() <= grad_tol * scale, (name, np.abs(g - ref).max(), scale)
            continue
        last = "tdnn/tdnn%d_" % (len(O.frame_layers(cfg_o)) + 2)
        if name.endswith("_conv/bias") or (name.endswith("_dense/bias") and not (name.startswith(last) and cfg_o.last_layer_no_bn)):
            # bias in front of a BatchNorm: the true gradient is exactly 0, both sides hold rounding noise
            assert np.abs(g).max() <= 1e-4 * max(1.0, np.abs(info["grads"][name.replace("/bias", "/kernel")]).max())
            continue
        assert np.all(np.isfinite(g)), name
        err = rel_err(g, ref)
        if report is not None:
            report["grad:" + name] = err
        assert err <= grad_tol, (name, err)
    # optimiser + BN moving averages
    eng.apply(lr, 1.0)
    after = eng.get_variables()
    for name, v in after.items():
        ref = newV[name]
        if name.endswith("/bias") and not name.startswith("softmax"):
            continue
        denom = max(np.abs(ref).max(), 1e-12)
        diff = np.abs(v - ref)
        if cfg_o.optimizer == "adam" and name in info["grads"]:
            # Adam's first step is lr*g/(|g|+1e-8): entries whose gradient is rounding noise (|g| << 1e-8-ish
            # relative to the tensor) get a step whose sign is noise on both sides; the kernel itself is
            # pinned on identical inputs in test_gpu_ops.py::test_optimizers_and_reductions.
            gref = np.abs(info["grads"][name].reshape(v.shape))
            diff = diff[gref >= 1e-3 * gref.max()]
        # var_tol of the variable, plus - for SGD and the first momentum step - what the gradient tolerance above admits through lr*g
        # (tiny batches have gradients that are large against the weights)
        slack = 0.0
        if cfg_o.optimizer in ("sgd", "momentum") and name in info["grads"]:      # first momentum step: lr*g, lr*(1+m)*g with nesterov
            slack = lr * (1.0 + (cfg_o.momentum if cfg_o.optimizer == "momentum" else 0.0)) * grad_tol * np.abs(info["grads"][name]).max()
        if report is not None:
            report["var:" + name] = diff.max() / denom
        assert diff.max() <= var_tol * denom + slack, (name, diff.max() / denom)


def test_second_step_uses_updated_weights_and_staged_backward():
    """Two consecutive steps (kernel-layout weight copies must be rebuilt) with the staged backward
    used for all-reduce overlap; gradients must equal the single-call backward bit for bit."""
    kw = dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True)
    B, T = 6, 33      # 6 rows per segment-level BatchNorm: with 4 the two-step error of either precision sits AT the 1e-4 bound
    eng, cfg_o, V = _make(kw, B, T)      # (5e-5 ... 1.2e-4 depending on the summation order; tests/tools/two_step_error.py)
    rs = np.random.RandomState(1)
    opt = {}
    for it in range(2):
        x = rs.randn(B, T, 30).astype(np.float32)
        labels = rs.randint(0, cfg_o.num_speakers, B).astype(np.int32)
        eng.forward(x, True)
        eng.loss(labels, it, True)
        eng.backward(-1)
        V, opt, info = oracle_step_with_gpu_relu_pattern(eng, V, cfg_o, x.astype(np.float64), labels, 0.1, it, opt)
        g_all = eng.grads.clone()
        eng.grads.zero_()
        covered = 0
        for st in range(4):
            eng.backward(st)
            b, e = eng.stage_grad_range(st)
            covered += e - b
        assert covered == eng.n_train
        assert torch.equal(g_all, eng.grads), "staged backward differs from the single-call backward"
        eng.apply(0.1, 1.0)
    after = eng.get_variables()
    for name in ("tdnn/tdnn2_conv/kernel", "tdnn/tdnn5_dense/kernel", "softmax/output/kernel", "tdnn/tdnn3_bn/moving_variance"):
        ref = V[name]
        assert np.abs(after[name] - ref).max() / np.abs(ref).max() <= 1e-4, name
    eng.close()


def test_inference_mode_embeddings_variable_length():
    """Trainer.predict path: is_training=False (moving statistics), B=1, any T >= 15, capacity reuse."""
    kw = dict(loss_func="softmax")
    eng, cfg_o, V = _make(kw, 1, 50, max_batch=4, max_frames=700)
    rs = np.random.RandomState(3)
    for T in (15, 25, 200, 700):
        x = rs.randn(1, T, 30).astype(np.float32)
        _, ep, _ = O.tdnn_forward(V, x.astype(np.float64), cfg_o, False)
        eng.forward(x, False)
        for name in ("tdnn6_dense", "tdnn7_dense", "output"):
            got = eng.endpoint(name).cpu().numpy()
            assert rel_err(got, ep[name].reshape(got.shape)) <= 1e-4, (T, name, rel_err(got, ep[name].reshape(got.shape)))
    with pytest.raises(Exception):
        eng.forward(rs.randn(1, 14, 30).astype(np.float32), False)     # shorter than the receptive field
    with pytest.raises(Exception):
        eng.forward(rs.randn(5, 20, 30).astype(np.float32), False)     # over capacity
    eng.close()


def test_inference_mode_long_utterances():
    """extract.py's shapes: one 10 000-frame utterance (its default --chunk-size, the longest single forward) and a stack of
    equal-length chunks of a split utterance (B = 3 x 5 000), attention-pooled as well.  The fp64 oracle needs minutes for
    these, so the long forward is tied to the oracle-checked short one by size-independent properties: with moving statistics
    every frame-level output row depends only on its own context window (valid convolutions), so rows of the long forward
    equal rows of 700-frame windows cut from it; pooling and tdnn6 are then re-derived in fp64 from the fetched tdnn5 output;
    and a chunk's embedding does not depend on what else is in the batch."""
    rs = np.random.RandomState(11)
    relu = lambda v: np.maximum(v, 0)
    for kw in (dict(loss_func="softmax"), dict(loss_func="softmax", pooling_type="self_attention", att_key_num_nodes=(200, 120))):
        eng, cfg_o, V = _make(kw, 1, 50, max_batch=3, max_frames=10000)
        att = "pooling_type" in kw
        T = 10000
        x = rs.randn(1, T, 30).astype(np.float32)
        eng.forward(x, False)
        z5 = eng.endpoint("tdnn5_bn").cpu().numpy().astype(np.float64)            # [T-14, 1500]
        pool = eng.endpoint("pooling").cpu().numpy()[0]
        emb = eng.endpoint("tdnn6_dense").cpu().numpy()[0]
        w = eng.endpoint("attention_weights").cpu().numpy().reshape(-1).astype(np.float64) if att else np.full(T - 14, 1.0 / (T - 14))
        assert z5.shape == (T - 14, 1500) and abs(w.sum() - 1.0) < 1e-5
        a5 = relu(z5)
        mean = (a5 * w[:, None]).sum(0)
        var = (((a5 - mean) ** 2) * w[:, None]).sum(0)
        ref_pool = np.concatenate([mean, np.sqrt(np.where(var <= 1e-12, 1e-12, var))])
        assert rel_err(pool, ref_pool) <= 1e-5, rel_err(pool, ref_pool)
        ref_emb = ref_pool @ V["tdnn/tdnn6_dense/kernel"] + V["tdnn/tdnn6_dense/bias"]
        assert rel_err(emb, ref_emb) <= 1e-5, rel_err(emb, ref_emb)
        for s0 in (0, 4321, T - 700):                                              # windows: rows s0 .. s0+685 of the long forward
            eng.forward(x[:, s0:s0 + 700], False)
            zw = eng.endpoint("tdnn5_bn").cpu().numpy()
            assert rel_err(zw, z5[s0:s0 + 686]) <= 2e-6, (s0, rel_err(zw, z5[s0:s0 + 686]))
        # the 700-frame window itself against the oracle (the anchor of the chain)
        _, ep, _ = O.tdnn_forward(V, x[:, :700].astype(np.float64), cfg_o, False)
        eng.forward(x[:, :700], False)
        for name in ("tdnn5_bn", "pooling", "tdnn6_dense"):
            got = eng.endpoint(name).cpu().numpy()
            assert rel_err(got, ep[name].reshape(got.shape)) <= 5e-5, (name, rel_err(got, ep[name].reshape(got.shape)))
        # three 5 000-frame chunks in one batch == each alone
        xb = rs.randn(3, 5000, 30).astype(np.float32)
        eng.forward(xb, False)
        stacked = eng.endpoint("tdnn6_dense").cpu().numpy()
        for i in range(3):
            eng.forward(xb[i:i + 1], False)
            alone = eng.endpoint("tdnn6_dense").cpu().numpy()[0]
            assert rel_err(stacked[i], alone) <= 2e-6, (i, rel_err(stacked[i], alone))
        eng.close()


def test_valid_mode_zeroes_margin():
    """build('valid') forces asoftmax_m=1 / amsoftmax_m=0 / arcsoftmax_m=0 (trainer.py:261-271)."""
    kw = dict(loss_func="additive_angular_margin_softmax", margin_m=0.3, lambda_gamma=1.0, last_layer_linear=True)
    B, T = 5, 30
    eng, cfg_o, V = _make(kw, B, T)
    rs = np.random.RandomState(8)
    x = rs.randn(B, T, 30).astype(np.float32)
    labels = rs.randint(0, cfg_o.num_speakers, B).astype(np.int32)
    feats, ep, _ = O.tdnn_forward(V, x.astype(np.float64), cfg_o, False)
    ref, _, _ = O.margin_softmax_loss("additive_angular_margin_softmax", feats, labels, V["softmax/output/kernel"], 0.0, 0.0)
    eng.forward(x, False)
    eng.loss(labels, 10 ** 6, with_margin=False)
    raw, _ = eng.losses()
    assert abs(raw - ref) <= 1e-4 * abs(ref)
    eng.close()


def test_full_size_properties():
    """BASELINE shape S1 (128 x 200 x 30, 7351 speakers): too slow for the oracle's full backward in a
    unit test, so check size-independent properties: finite NaN-free gradients (the reference's own
    assertion, tdnn.py:282), determinism (bitwise-equal repeat), BN-bias gradients ~ 0, gradient of the
    regulariser alone (lr-scaled weight decay identity), and loss == log(N) scale at init."""
    from tf_kaldi_speaker_amd import engine as E
    c = E.make_config(30, 7351, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True,
                      max_batch=128, max_frames=200)
    eng = E.Engine(c)
    eng.init_variables(seed=0)
    rs = np.random.RandomState(0)
    x = rs.randn(128, 200, 30).astype(np.float32)
    labels = rs.randint(0, 7351, 128).astype(np.int32)
    eng.forward(x, True)
    eng.loss(labels, 0, True)
    eng.backward(-1)
    raw, reg = eng.losses()
    assert np.isfinite(raw) and abs(raw - np.log(7351)) < 1.5
    g1 = eng.grads.clone()
    assert torch.isfinite(g1).all()
    eng.forward(x, True)
    eng.loss(labels, 0, True)
    eng.backward(-1)
    assert torch.equal(g1, eng.grads), "training step is not deterministic"
    grads = eng.get_gradients()
    assert np.abs(grads["tdnn/tdnn2_conv/bias"]).max() < 1e-4
    # stats pooling output is [mean, std]: std half must be strictly positive
    pool = eng.endpoint("pooling").cpu().numpy()
    assert pool.shape == (128, 3000) and (pool[:, 1500:] > 0).all()
    eng.close()


def test_out_of_range_label_poisons_the_loss_instead_of_faulting():
    eng, cfg_o, V = _make(dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True), 4, 30)
    rs = np.random.RandomState(0)
    x = rs.randn(4, 30, 30).astype(np.float32)
    eng.forward(x, True)
    eng.loss(np.array([1, 2, cfg_o.num_speakers + 5, -3], np.int32), 0, True)
    raw, _ = eng.losses()
    assert np.isnan(raw)
    eng.loss(np.array([1, 2, 3, 4], np.int32), 0, True)
    raw, _ = eng.losses()
    assert np.isfinite(raw)
    eng.close()


@pytest.mark.parametrize("scale", [1e3, 1e-3], ids=["x1e3", "x1e-3"])
def test_full_size_split_precision_operand_ranges(scale, xv_precision):
    """BASELINE shape S1 at full size with the features scaled by 1e3 / 1e-3 and a 40-sigma outlier: every operand range of the
    first layers moves (BatchNorm brings the rest back), which is what the per-tensor power-of-two plane scales of the split
    precision path have to follow.  Both precisions are held to the float64 oracle at this size and unit scale in
    tests/test_gpu_full_size.py (loss, endpoints, every gradient, update); this test adds the two scaled inputs, for which
    the fp32-input MFMA path is the reference: forward quantities to rounding, and the gradients behind the last ReLU (no
    on/off pattern in them to flip between two roundings)."""
    if xv_precision == "f32":
        pytest.skip("compares the two precisions itself")
    from tf_kaldi_speaker_amd import engine as E
    rs = np.random.RandomState(3)
    x = (rs.randn(128, 200, 30) * scale).astype(np.float32)
    x[5, 17, 3] = 40.0 * scale                                  # an outlier sets the input plane scale
    labels = rs.randint(0, 7351, 128).astype(np.int32)

    def run(prec):
        eng = E.Engine(E.make_config(30, 7351, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True,
                                     max_batch=128, max_frames=200, precision=prec))
        eng.init_variables(seed=1)
        eng.forward(x, True)
        eng.loss(labels, 1000, True)
        eng.backward(-1)
        raw, reg = eng.losses()
        res = (raw, reg, eng.endpoint("tdnn6_dense").cpu().numpy(), eng.endpoint("tdnn5_bn").cpu().numpy()[:4096],
               {k: v.copy() for k, v in eng.get_gradients().items() if k.startswith(("tdnn/tdnn7", "softmax"))})
        eng.close()
        return res

    a, b = run("f32"), run("f16x3")
    assert abs(a[0] - b[0]) <= 2e-6 * abs(a[0]) and abs(a[1] - b[1]) <= 2e-6 * abs(a[1]), (a[0], b[0])
    assert rel_err(b[2], a[2]) <= 2e-5 and rel_err(b[3], a[3]) <= 2e-5, (rel_err(b[2], a[2]), rel_err(b[3], a[3]))
    for k in a[4]:
        if k.endswith("/bias") and not k.startswith("softmax"):
            continue
        assert rel_err(b[4][k], a[4][k]) <= 5e-5, (k, rel_err(b[4][k], a[4][k]))
