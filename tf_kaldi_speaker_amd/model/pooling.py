"""Pooling layers with the reference's signatures (model/pooling.py:9,37,195):

    f(features, aux_features, endpoints, params, is_training) -> [batch, 2 * dim]

`statistics_pooling` and `self_attention` run on the GPU through the op-level C-ABI (csrc/xv_elementwise.hip,
csrc/xv_attention.hip); inside the training engine the same kernels are driven natively (csrc/xv_engine.hip), so these
callables are the evaluation forms a caller of the reference API reaches.  `self_attention` covers the single-head form
every shipped attention config uses (key network on `endpoints[att_key_input]`, value = `endpoints[att_value_input]`, key
not split, no value network, no penalty term, no post non-linearity); the other options of the reference function are
refused by name.  `ghost_vlad` is outside the hot path.

The attention variables live in a module-level store under the reference's scope names ("attention/att_key0/att_key0_dense/kernel",
..., "attention/query"), like the loss functions' `softmax/output/kernel` (model/loss.py): the first call creates them,
a second creating call raises as tf.get_variable would, `reuse_variables()` re-enters them.
"""
from collections import OrderedDict

import numpy as np
import torch

try:
    from .. import ops
    from .common import to_device, shape_list
except (ImportError, ValueError):
    import ops
    from model.common import to_device, shape_list

VAR2STD_EPSILON = 1e-12
BN_EPSILON = 1e-3            # tf.layers.batch_normalization default

_VARIABLES = OrderedDict()   # "attention/..." -> device tensor
_STATE = {"seed": 0, "reuse": False}


def reset_variables(seed=0):
    _VARIABLES.clear()
    _STATE["seed"], _STATE["reuse"] = seed, False


def reuse_variables(flag=True):
    """tf.get_variable_scope().reuse_variables() for the attention scope."""
    _STATE["reuse"] = bool(flag)


def set_variable(name, value):
    _VARIABLES[name] = to_device(np.asarray(value, np.float32))


def get_variables():
    return _VARIABLES


def _get(name, shape, init):
    if name in _VARIABLES:
        v = _VARIABLES[name]      # created by an earlier call or preset with set_variable(): re-entered like a reused scope
        assert tuple(v.shape) == tuple(shape), "variable %s has shape %s, requested %s" % (name, tuple(v.shape), tuple(shape))
        return v
    if _STATE["reuse"]:
        raise ValueError("Variable %s does not exist, or was not created with tf.get_variable()." % name)
    rs = np.random.RandomState(_STATE["seed"] + len(_VARIABLES))
    if init == "glorot":
        lim = np.sqrt(6.0 / (shape[0] + shape[1]))
        val = rs.uniform(-lim, lim, size=shape)
    elif init == "query":        # tf.initializers.truncated_normal(stddev=0.1), pooling.py:131-132
        val = np.clip(rs.randn(*shape) * 0.1, -0.2, 0.2)
    else:
        val = np.full(shape, float(init))
    _VARIABLES[name] = to_device(val.astype(np.float32))
    return _VARIABLES[name]


def statistics_pooling(features, aux_features, endpoints, params, is_training):
    """[batch, length, dim] -> [batch, 2*dim] = concat(mean, stddev) with the variance floor of
    reference pooling.py:9-34 (wave-shuffle Welford kernel, csrc/xv_elementwise.hip)."""
    x = to_device(features)
    assert x.dim() == 3
    if x.shape[2] % 4 != 0:
        raise ValueError("statistics_pooling: the channel count must be a multiple of 4 (got %d)" % x.shape[2])
    return ops.stat_pool_forward(x)


def _dense(x2d, num_nodes, scope):
    """tf.layers.dense inside variable_scope(name): <name>/<name>_dense/{kernel,bias} (common.py dense_* helpers)."""
    c = x2d.shape[1]
    kernel = _get("attention/%s/%s_dense/kernel" % (scope, scope), (c, num_nodes), "glorot")
    bias = _get("attention/%s/%s_dense/bias" % (scope, scope), (num_nodes,), 0.0)
    c_pad = (c + 3) // 4 * 4
    xin = x2d if c_pad == c else ops.pad_channels(x2d, c_pad)
    wt = ops.prep_weight_fwd(kernel.view(1, c, num_nodes), c_pad)
    return ops.affine_forward(xin.view(xin.shape[0], 1, c_pad), 1, wt, bias, num_nodes)


def _bn_relu(z, scope, params, is_training, relu=True):
    """tf.layers.batch_normalization(momentum=params.batchnorm_momentum, training=is_training) [+ relu]; in training
    mode the moving statistics are updated in place (the UPDATE_OPS the reference's train_op depends on)."""
    n = z.shape[1]
    gamma = _get("attention/%s/%s_bn/gamma" % (scope, scope), (n,), 1.0)
    beta = _get("attention/%s/%s_bn/beta" % (scope, scope), (n,), 0.0)
    mmean = _get("attention/%s/%s_bn/moving_mean" % (scope, scope), (n,), 0.0)
    mvar = _get("attention/%s/%s_bn/moving_variance" % (scope, scope), (n,), 1.0)
    if is_training:
        part = ops.col_stats(z)
        _, _, scale, shift = ops.bn_finalize(part, z.shape[0], gamma, beta, BN_EPSILON, float(params.batchnorm_momentum), 0, mmean, mvar)
    else:
        scale, shift = ops.bn_inference_scale(gamma, beta, mmean, mvar, BN_EPSILON)
    return ops.bn_apply(z, scale, shift, relu)


def self_attention(features, aux_features, endpoints, params, is_training=None):
    """Self-attentive statistics pooling, reference pooling.py:37-192, in the single-head form of the shipped configs
    (egs/voxceleb/v1/nnet_conf/tdnn_softmax_1e-2_tdnn4_att.json, egs/fisher/v1/nnet_conf/*_att*.json).

    `features` is not used (as in the reference): the key comes from endpoints[params.att_key_input] through the key
    network (att_key_num_nodes[:-1] as dense+bn+relu, the last layer by att_key_network_type 0 affine | 1 +relu |
    2 +bn+relu | 3 +tanh), the value is endpoints[params.att_value_input].  weights = softmax_t(key.query [/ sqrt(dk)]),
    output = concat(weighted mean, sqrt of the weighted variance floored at 1e-12).  Side effects as in the reference:
    endpoints gains the key layers' "<name>_dense/_bn/_relu/_tanh", "attention_weights" [b, 1, t] and
    "att_output_before_nonlinear".
    """
    d = params.dict
    if d.get("network_relu_type", "relu") != "relu":
        raise NotImplementedError("self_attention: network_relu_type %r (relu only)" % d.get("network_relu_type"))
    if len(d.get("att_value_num_nodes", [])) != 0:
        raise NotImplementedError("self_attention on the MI355X kernels: no value network (att_value_num_nodes must be empty)")
    if int(d.get("att_num_heads", 1)) != 1 or d.get("att_split_key", False):
        raise NotImplementedError("self_attention on the MI355X kernels: one head, key not split")
    if d.get("att_apply_nonlinear", False):
        raise NotImplementedError("self_attention on the MI355X kernels: att_apply_nonlinear is not supported")
    if float(d.get("att_penalty_term", 0) or 0) != 0.0:
        raise NotImplementedError("self_attention on the MI355X kernels: att_penalty_term must be 0 (one head has no penalty)")
    key_nodes = list(d["att_key_num_nodes"])
    if len(key_nodes) < 1:
        raise NotImplementedError("self_attention: att_key_num_nodes needs at least the key layer")
    key_type = int(d["att_key_network_type"])
    if key_type not in (0, 1, 2, 3):
        raise NotImplementedError("self_attention: att_key_network_type %r is not one of 0..3" % key_type)

    value = to_device(endpoints[params.att_value_input])
    key_in = to_device(endpoints[params.att_key_input])
    assert value.dim() == 3 and key_in.dim() == 3 and value.shape[:2] == key_in.shape[:2], \
        "the key and the value must be [batch, length, dim] with the same batch and length"
    b, t, vdim = shape_list(value)
    k = key_in.reshape(b * t, key_in.shape[2])

    for index, num_nodes in enumerate(key_nodes[:-1]):        # intermediate layers: affine + bn + relu
        name = "att_key%d" % index
        z = _dense(k, num_nodes, name)
        endpoints[name + "_dense"] = z.view(b, t, num_nodes)
        bn = _bn_relu(z, name, params, is_training, relu=False)
        endpoints[name + "_bn"] = bn.view(b, t, num_nodes)
        k = ops.key_activation(bn, 1)
        endpoints[name + "_relu"] = k.view(b, t, num_nodes)
    name = "att_key%d" % (len(key_nodes) - 1)
    kdim = key_nodes[-1]
    z = _dense(k, kdim, name)
    endpoints[name + "_dense"] = z.view(b, t, kdim)
    act = 0
    if key_type == 1:
        act = 1
        endpoints[name + "_relu"] = ops.key_activation(z, 1).view(b, t, kdim)
    elif key_type == 2:
        bn = _bn_relu(z, name, params, is_training, relu=False)
        endpoints[name + "_bn"] = bn.view(b, t, kdim)
        z = ops.key_activation(bn, 1)
        endpoints[name + "_relu"] = z.view(b, t, kdim)
    elif key_type == 3:
        act = 3
        endpoints[name + "_tanh"] = ops.key_activation(z, 3).view(b, t, kdim)      # endpoint only; the score kernel applies tanh itself

    query = _get("attention/query", (1, kdim), "query")
    scale = 1.0 / np.sqrt(float(kdim)) if d.get("att_use_scale", False) else 1.0
    weights = ops.softmax_segments(ops.att_score(z, act, query.view(-1), scale), b, t)
    endpoints["attention_weights"] = weights.view(b, 1, t)

    # weighted mean / stddev of the value: the fused BN+ReLU pooling kernel with the identity transform
    vpad = (vdim + 3) // 4 * 4
    v2 = value.reshape(b * t, vdim)
    if vpad != vdim:
        v2 = ops.pad_channels(v2, vpad)
    one = torch.ones(vpad, dtype=torch.float32, device=v2.device)
    zero = torch.zeros(vpad, dtype=torch.float32, device=v2.device)
    pooled = ops.stat_pool_forward_bn(v2, b, t, one, zero, relu=False, weights=weights)
    att = torch.cat([pooled[:, :vdim], pooled[:, vpad:vpad + vdim]], dim=1) if vpad != vdim else pooled
    endpoints["att_output_before_nonlinear"] = att
    return att


def ghost_vlad(features, aux_features, endpoints, params, is_training=None):
    raise NotImplementedError("Not implement ghost_vlad pooling on the MI355X engine (no shipped single-task config selects it, "
                              "SURVEY.md section 2)")
