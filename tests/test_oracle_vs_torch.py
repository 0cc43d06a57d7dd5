"""Second, independent implementation of the hot path (torch CPU ops + autograd,
float64) used to pin the oracle's hand-written forward AND backward
(SURVEY.md section 7 step 1-ii).  Nothing here touches the GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import xvector_oracle as O

torch.set_num_threads(4)


def _t(a, grad=True):
    return torch.tensor(np.asarray(a, np.float64), requires_grad=grad)


def torch_forward(V, x, cfg, labels, step, training=True):
    """tdnn + loss + regulariser written with torch ops only."""
    tv = {k: _t(v, O.is_trainable(k)) for k, v in V.items()}
    h = _t(x, False)
    eps = 1e-3
    ep = {}

    def act(prefix, y, scope=""):
        """network_relu_type (tdnn.py:24-30): relu | tf.nn.leaky_relu (alpha 0.2) | prelu = relu(x) + alpha (x - |x|) / 2."""
        if cfg.network_relu_type == "lrelu":
            return F.leaky_relu(y, 0.2)
        if cfg.network_relu_type == "prelu":
            return torch.relu(y) + tv["tdnn/%s%s_relu/alpha" % (scope, prefix)] * (y - y.abs()) * 0.5
        return torch.relu(y)

    def bn(prefix, z, scope=""):
        base = "tdnn/%s%s_bn/" % (scope, prefix)
        g, b = tv[base + "gamma"], tv[base + "beta"]
        mm, mv = tv[base + "moving_mean"], tv[base + "moving_variance"]
        z2 = z.reshape(-1, z.shape[-1])
        if training:
            y = F.batch_norm(z2, None, None, g, b, True, 0.0, eps)
        else:
            y = F.batch_norm(z2, mm, mv, g, b, False, 0.0, eps)
        return y.reshape(z.shape)

    # the torch side builds the stack from the table itself (reference: 5/5/7/1/1; extended: cfg.frame_layers), not from the oracle's helper
    table = cfg.frame_layers if cfg.frame_layers else ((5, 512), (5, 512), (7, 512), (1, 512), (1, cfg.num_nodes_pooling_layer))
    nf = len(table)
    s0, s1 = "tdnn%d" % (nf + 1), "tdnn%d" % (nf + 2)
    for li, (ctx, _) in enumerate(table):
        prefix, kind = "tdnn%d" % (li + 1), ("conv" if ctx > 1 else "dense")
        name = "%s_%s" % (prefix, kind)
        K, b = tv["tdnn/%s/kernel" % name], tv["tdnn/%s/bias" % name]
        if kind == "conv":
            w = K[0].permute(2, 1, 0)            # [k,C,O] -> [O,C,k]
            z = F.conv1d(h.transpose(1, 2), w, b).transpose(1, 2)
        else:
            z = F.linear(h, K.t(), b)
        ep[name] = z
        h = act(prefix, bn(prefix, z))
        ep[prefix + "_relu"] = h
    if cfg.pooling_type == "self_attention":
        a0, a1 = "tdnn/attention/att_key0/att_key0_dense/", "tdnn/attention/att_key1/att_key1_dense/"
        k = act("att_key0", bn("att_key0", F.linear(ep["tdnn%d_relu" % (nf - 1)], tv[a0 + "kernel"].t(), tv[a0 + "bias"]), "attention/att_key0/"),
                "attention/att_key0/")
        k = F.linear(k, tv[a1 + "kernel"].t(), tv[a1 + "bias"])
        ep["att_key1_dense"] = k
        if cfg.att_key_network_type == 3:
            k = torch.tanh(k)
        elif cfg.att_key_network_type == 1:
            k = act("att_key1", k, "attention/att_key1/")
        elif cfg.att_key_network_type == 2:
            k = act("att_key1", bn("att_key1", k, "attention/att_key1/"), "attention/att_key1/")
        score = torch.einsum("btd,hd->bth", k, tv["tdnn/attention/query"])[:, :, 0]
        if cfg.att_use_scale:
            score = score / np.sqrt(k.shape[-1])
        w = torch.softmax(score, dim=1)
        ep["attention_weights"] = w[:, None, :]
        mean = torch.einsum("btc,bt->bc", h, w)
        var = torch.einsum("btc,bt->bc", (h - mean[:, None, :]) ** 2, w)
    else:
        mean = h.mean(dim=1)
        var = ((h - mean[:, None, :]) ** 2).mean(dim=1)
    mask = (var <= 1e-12).double()
    var = (1 - mask) * var + mask * 1e-12
    h = torch.cat([mean, var.sqrt()], dim=1)
    ep["pooling"] = h
    z = F.linear(h, tv["tdnn/%s_dense/kernel" % s0].t(), tv["tdnn/%s_dense/bias" % s0])
    ep[s0 + "_dense"] = z
    h = act(s0, bn(s0, z))
    z = F.linear(h, tv["tdnn/%s_dense/kernel" % s1].t(), tv["tdnn/%s_dense/bias" % s1])
    ep[s1 + "_dense"] = z
    h = z
    if not cfg.last_layer_no_bn:
        h = bn(s1, h)
    if not cfg.last_layer_linear:
        h = act(s1, h)
    if cfg.feature_norm:
        ss = (h * h).sum(dim=-1, keepdim=True)
        h = h * torch.rsqrt(torch.clamp(ss, min=1e-12)) * cfg.feature_scaling_factor
    ep["output"] = h
    if labels is None:
        return ep, tv, None, None
    lab = torch.tensor(labels, dtype=torch.long)
    W = tv["softmax/output/kernel"]
    if cfg.loss_func == "softmax":
        logits = h @ W + tv["softmax/output/bias"]
        loss = F.cross_entropy(logits, lab)
    else:
        wn = W * torch.rsqrt(torch.clamp((W * W).sum(dim=0, keepdim=True), min=1e-12))
        logits = h @ wn
        if cfg.loss_func == "asoftmax" and cfg.margin_m == 1:
            loss = F.cross_entropy(logits, lab)
        else:
            idx = torch.arange(h.shape[0])
            sel = logits[idx, lab]
            fn = torch.clamp(h.norm(dim=1), min=1e-12)
            c = torch.clamp(sel / fn, -1 + 1e-12, 1 - 1e-12)
            m = cfg.margin_m
            if cfg.loss_func == "asoftmax":
                if m == 2:
                    phi = 2 * torch.sign(c) * c * c - 1
                else:
                    c2, c4 = c * c, c ** 4
                    s0 = torch.sign(c)
                    s3 = torch.sign(2 * c2 - 1) * s0
                    phi = s3 * (8 * c4 - 8 * c2 + 1) + (2 * s0 + s3 - 3)
            elif cfg.loss_func == "additive_margin_softmax":
                phi = c - m
            else:
                sin = torch.sqrt(torch.clamp(1 - c * c, min=1e-12))
                cpm = c * np.cos(m) - sin * np.sin(m)
                phi = torch.where(c > np.cos(np.pi - m), cpm, -cpm - 2)
            lam = O.margin_lambda(cfg.lambda_min, cfg.lambda_base, cfg.lambda_gamma, cfg.lambda_power, step)
            fa = 1.0 / (1.0 + lam)
            delta = torch.zeros_like(logits)
            delta[idx, lab] = phi * fn - sel
            updated = (1 - fa) * logits + fa * (logits + delta)
            loss = F.cross_entropy(updated, lab)
    for aux in cfg.aux_loss_func:
        if aux == "ring_loss":
            loss = loss + cfg.ring_loss_lambda * ((h.norm(dim=1) - tv["softmax_ringloss/r"]) ** 2).mean()
        else:
            wn2 = W * torch.rsqrt(torch.clamp((W * W).sum(dim=0, keepdim=True), min=1e-12))
            loss = loss + cfg.mhe_lambda / ((2.0 - 2.0 * (wn2.t()[lab] @ wn2)).mean() + 1e-6)
    reg = 0
    for k, v in tv.items():
        if k.endswith("/kernel"):
            s = cfg.weight_l2_regularizer
            if k.startswith("softmax/") and cfg.output_weight_l2_regularizer is not None:
                s = cfg.output_weight_l2_regularizer
            reg = reg + s * (v * v).sum() / 2
    return ep, tv, loss, reg


CASES = [
    dict(loss_func="softmax"),
    dict(loss_func="asoftmax", margin_m=1, lambda_min=10, lambda_gamma=1e-5),
    dict(loss_func="asoftmax", margin_m=2, lambda_min=10, lambda_gamma=1e-5, last_layer_linear=True),
    dict(loss_func="asoftmax", margin_m=4, lambda_min=0, lambda_gamma=1e-2, last_layer_linear=True),
    dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True),
    dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, last_layer_no_bn=True),
    dict(loss_func="additive_angular_margin_softmax", margin_m=0.25, lambda_gamma=1e-2, last_layer_linear=True),
    dict(loss_func="additive_margin_softmax", margin_m=0.2, feature_norm=True, feature_scaling_factor=30.0,
         last_layer_linear=True),
    # self-attention pooling, the shipped form (nnet_conf/*_tdnn4_att.json) and its affine-key / unscaled variants
    dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, pooling_type="self_attention",
         att_key_num_nodes=(24, 20)),
    dict(loss_func="softmax", pooling_type="self_attention", att_key_num_nodes=(16, 12), att_key_network_type=0, att_use_scale=False),
    dict(loss_func="softmax", pooling_type="self_attention", att_key_num_nodes=(16, 12), att_key_network_type=1),
    dict(loss_func="softmax", pooling_type="self_attention", att_key_num_nodes=(16, 12), att_key_network_type=2),
    # auxiliary losses of the shipped *_r0.01.json / *_mhe0.01.json configs (loss.py:985-1036)
    dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, aux_loss_func=("ring_loss", "mhe_loss"),
         ring_loss_init=3.0, ring_loss_lambda=0.05, mhe_lambda=0.05),
    # network_relu_type variants (tdnn.py:24-30) incl. the relu'd last layer and the attention key networks
    dict(loss_func="softmax", network_relu_type="lrelu"),
    dict(loss_func="softmax", network_relu_type="prelu", last_layer_no_bn=True),
    dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, network_relu_type="prelu",
         pooling_type="self_attention", att_key_num_nodes=(16, 12), att_key_network_type=1),
    dict(loss_func="softmax", network_relu_type="lrelu", pooling_type="self_attention", att_key_num_nodes=(16, 12), att_key_network_type=2),
    # extended frame-layer tables (no reference counterpart, BASELINE configs[4]): 10 layers of mixed contexts, a 3-layer stack,
    # and a 6-layer one under the attention head (key input = the last-but-one frame layer)
    dict(loss_func="asoftmax", margin_m=4, lambda_min=10, lambda_gamma=1e-5, last_layer_linear=True,
         frame_layers=((5, 16), (1, 16), (3, 24), (1, 16), (3, 16), (1, 16), (3, 16), (1, 16), (1, 16), (1, 20))),
    dict(loss_func="softmax", frame_layers=((3, 8), (2, 12), (1, 20))),
    dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, pooling_type="self_attention", att_key_num_nodes=(16, 12),
         frame_layers=((5, 16), (3, 16), (1, 24), (3, 16), (1, 12), (1, 20))),
]


@pytest.mark.parametrize("kw", CASES, ids=lambda d: d["loss_func"] + "_" + str(d.get("margin_m", "")) + ("_%dlayers" % len(d["frame_layers"]) if "frame_layers" in d else "") + ("_att%d" % d["att_key_network_type"] if "att_key_network_type" in d else "_att" if "pooling_type" in d else "") + ("_aux" if "aux_loss_func" in d else "") + ("_" + d["network_relu_type"] if "network_relu_type" in d else ""))
def test_full_step_forward_backward(kw):
    cfg = O.Config(feat_dim=6, num_speakers=11, num_nodes_pooling_layer=20, num_nodes_last_layer=16, **kw)
    # the layer widths 512 are fixed by tdnn.py; keep B,T small instead
    V = O.init_variables(cfg, seed=3, dtype=np.float64)
    rs = np.random.RandomState(7)
    for k in V:   # perturb BN params / biases so every gradient path is exercised
        if k.endswith(("gamma", "beta", "bias", "alpha")):
            V[k] = V[k] + 0.1 * rs.randn(*V[k].shape)
    B, T = 5, 22
    x = rs.randn(B, T, cfg.feat_dim)
    labels = rs.randint(0, cfg.num_speakers, B)
    step = 777

    _, _, info = O.train_step(V, {}, cfg, x, labels, 0.0, step)
    ep, tv, loss, reg = torch_forward(V, x, cfg, labels, step)
    (loss + reg).backward()

    assert abs(float(loss.detach()) - float(info["raw_loss"])) < 1e-10 * max(1, abs(float(loss.detach())))
    assert abs(float((loss + reg).detach()) - float(info["total_loss"])) < 1e-10 * max(1, abs(float((loss + reg).detach())))
    nf = len(cfg.frame_layers) if cfg.frame_layers else 5
    names = ["tdnn1_conv", "tdnn3_conv" if not cfg.frame_layers else "tdnn1_relu", "pooling", "tdnn%d_dense" % (nf + 1), "tdnn%d_dense" % (nf + 2), "output"]
    if cfg.pooling_type == "self_attention":
        names += ["att_key1_dense", "attention_weights"]
    for name in names:
        a, b = info["endpoints"][name], ep[name].detach().numpy()
        assert np.allclose(a, b, rtol=1e-9, atol=1e-11), name
    for k, v in tv.items():
        if not O.is_trainable(k):
            continue
        g = info["grads"][k].reshape(v.shape)
        tg = v.grad.numpy()
        scale = max(np.abs(tg).max(), 1e-12)
        assert np.abs(g - tg).max() <= 1e-8 * scale + 1e-12, (k, np.abs(g - tg).max(), scale)


def test_inference_mode_uses_moving_stats():
    cfg = O.Config(feat_dim=5, num_speakers=7, num_nodes_pooling_layer=12)
    V = O.init_variables(cfg, seed=1, dtype=np.float64)
    rs = np.random.RandomState(2)
    for k in V:
        if k.endswith("moving_mean"):
            V[k] = 0.3 * rs.randn(*V[k].shape)
        if k.endswith("moving_variance"):
            V[k] = 0.5 + rs.rand(*V[k].shape)
    x = rs.randn(2, 30, 5)
    _, ep, _ = O.tdnn_forward(V, x, cfg, False)
    ept, _, _, _ = torch_forward(V, x, cfg, None, 0, training=False)
    for name in ("tdnn6_dense", "tdnn7_dense", "output"):
        assert np.allclose(ep[name], ept[name].detach().numpy(), rtol=1e-9, atol=1e-11)


def test_finite_difference_spot_check():
    """fp64 finite differences on a few weights."""
    cfg = O.Config(feat_dim=4, num_speakers=6, num_nodes_pooling_layer=8, loss_func="additive_angular_margin_softmax",
                   margin_m=0.3, lambda_gamma=1.0, last_layer_linear=True)
    V = O.init_variables(cfg, seed=5, dtype=np.float64)
    rs = np.random.RandomState(11)
    x = rs.randn(4, 20, 4)
    labels = rs.randint(0, 6, 4)
    _, _, info = O.train_step(V, {}, cfg, x, labels, 0.0, 50)

    def total(Vp):
        return float(O.train_step(Vp, {}, cfg, x, labels, 0.0, 50)[2]["total_loss"])

    for name, idx in (("tdnn/tdnn2_conv/kernel", (0, 2, 100, 7)), ("tdnn/tdnn5_dense/kernel", (33, 3)),
                      ("tdnn/tdnn6_bn/gamma", (10,)), ("softmax/output/kernel", (100, 2)),
                      ("tdnn/tdnn1_conv/kernel", (0, 1, 2, 300))):
        h = 1e-6
        Vp = dict(V); Vm = dict(V)
        Vp[name] = V[name].copy(); Vp[name][idx] += h
        Vm[name] = V[name].copy(); Vm[name][idx] -= h
        fd = (total(Vp) - total(Vm)) / (2 * h)
        an = info["grads"][name].reshape(V[name].shape)[idx]
        assert abs(fd - an) < 1e-6 * max(1.0, abs(an)) + 1e-8, (name, fd, an)


def test_optimizers_match_torch():
    rs = np.random.RandomState(0)
    p0, g = rs.randn(50), rs.randn(50)
    # momentum (tf semantics == torch SGD with dampening 0 when lr constant)
    p = torch.tensor(p0.copy(), requires_grad=True)
    opt = torch.optim.SGD([p], lr=0.1, momentum=0.9)
    pn, acc = p0.copy(), np.zeros(50)
    for _ in range(3):
        p.grad = torch.tensor(g)
        opt.step()
        pn, acc = O.momentum_update(pn, g, acc, 0.1, 0.9, False)
    assert np.allclose(pn, p.detach().numpy(), rtol=1e-12, atol=1e-12)
    # adam: TF's epsilon sits outside the bias-corrected sqrt ("epsilon hat"); check a hand formula
    pn, m, v = O.adam_update(p0, g, np.zeros(50), np.zeros(50), 1, 0.001)
    lr_t = 0.001 * np.sqrt(1 - 0.999) / (1 - 0.9)
    ref = p0 - lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    assert np.allclose(pn, ref, rtol=1e-12)
    assert np.allclose(O.sgd_update(p0, g, 0.5), p0 - 0.5 * g)


def test_bn_moving_average_switch():
    z = np.random.RandomState(0).randn(10, 3)
    _, (mean, var, _, _) = O.batchnorm_train_fwd(z, np.ones(3), np.zeros(3))
    mm, mv = O.batchnorm_moving_update(np.zeros(3), np.ones(3), mean, var, 10, 0.99, False)
    assert np.allclose(mv, 0.99 + 0.01 * z.var(axis=0))
    _, mvu = O.batchnorm_moving_update(np.zeros(3), np.ones(3), mean, var, 10, 0.99, True)
    assert np.allclose(mvu, 0.99 + 0.01 * z.var(axis=0, ddof=1))
    assert np.allclose(mm, 0.01 * z.mean(axis=0))
