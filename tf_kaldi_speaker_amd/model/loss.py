"""Softmax loss family with the reference's signatures (model/loss.py:9,51,172,260):

    f(features[B,E], labels[B], num_outputs, params, is_training=None, reuse_variables=None, name="softmax")
        -> (loss, endpoints{"logits", "labels"})        side effect: params.dict["softmax_w"] = w

These are the *forward* (evaluation) forms on GPU buffers; gradients of the same kernels are
produced inside the engine (csrc/xv_engine.hip) that Trainer drives.  Variables live in a module-level
store keyed by "<name>/output/kernel" exactly like the TF variable scope, so a second call with
reuse_variables=True sees the same weight (loss.py:96-102).
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

_VARIABLES = OrderedDict()      # name -> torch tensor (device)
_SEED = [0]


def reset_variables(seed=0):
    _VARIABLES.clear()
    _SEED[0] = seed


def get_variable(name, shape, reuse, init="xavier"):
    if name in _VARIABLES:
        if reuse is None or reuse is False:
            raise ValueError("Variable %s already exists, disallowed. Did you mean to set reuse=True?" % name)
        v = _VARIABLES[name]
        assert tuple(v.shape) == (tuple(shape) or (1,)), "variable %s has shape %s, requested %s" % (name, tuple(v.shape), shape)
        return v
    if reuse is True:
        raise ValueError("Variable %s does not exist, or was not created with get_variable()." % name)
    rs = np.random.RandomState(_SEED[0] + len(_VARIABLES))
    if init == "xavier":                         # tf.contrib.layers.xavier_initializer(): uniform, loss.py:100-102
        lim = np.sqrt(6.0 / (shape[0] + shape[1]))
        val = rs.uniform(-lim, lim, size=shape).astype(np.float32)
    elif isinstance(init, float):                # constant initialiser (ring loss r, loss.py:1009-1011)
        val = np.full(shape if shape else (1,), init, np.float32)
    else:
        val = np.zeros(shape, np.float32)
    _VARIABLES[name] = to_device(val)
    return _VARIABLES[name]


def set_variable(name, value):
    _VARIABLES[name] = to_device(value)


def _lambda(params, prefix):
    lmin = float(params.dict[prefix + "_lambda_min"])
    base = float(params.dict[prefix + "_lambda_base"])
    gamma = float(params.dict[prefix + "_lambda_gamma"])
    power = float(params.dict[prefix + "_lambda_power"])
    params.dict[prefix + "_lambda_min"], params.dict[prefix + "_lambda_base"] = lmin, base
    params.dict[prefix + "_lambda_gamma"], params.dict[prefix + "_lambda_power"] = gamma, power
    step = float(params.dict.get("global_step", 0))
    return max(lmin, base * (1.0 + gamma * step) ** (-power))      # loss.py:144-145


def _run(kind, features, labels, num_outputs, params, reuse_variables, name, m, lam, with_bias):
    x = to_device(features)
    y = to_device(labels, torch.int32)
    assert len(shape_list(x)) == len(shape_list(y)) + 1
    e = x.shape[1]
    w = get_variable(name + "/output/kernel", (e, num_outputs), reuse_variables)
    params.dict["softmax_w"] = w
    bias = get_variable(name + "/output/bias", (num_outputs,), reuse_variables, init="zeros") if with_bias else None
    inv, wn, wnt = ops.loss_prep_weight(w, kind != 0)
    ldl = wn.shape[1]
    logits = torch.zeros((x.shape[0], ldl), dtype=torch.float32, device=x.device)
    logits[:, :num_outputs] = ops.affine_forward(x.view(x.shape[0], 1, e), 1, wnt, bias, num_outputs)
    loss, _, _, _ = ops.margin_softmax_rows(kind, logits, num_outputs, x, y, m, lam)
    endpoints = OrderedDict()
    endpoints["logits"] = logits[:, :num_outputs]
    endpoints["labels"] = y
    total = loss[0]
    aux_funcs = params.dict.get("aux_loss_func") or []
    if kind == 1 and m == 1.0:       # asoftmax, m = 1: the reference returns before the auxiliary losses (loss.py:110-115)
        aux_funcs = []
    for aux in aux_funcs:            # loss.py:985-1036, added by every other loss function (loss.py:40,161,249,347)
        if aux == "ring_loss":
            r = get_variable(name + "_ringloss/r", (), reuse_variables, init=float(params.ring_loss_init))
            total = total + ops.ring_loss(x, r, float(params.ring_loss_lambda))
            endpoints["ring_loss_r"] = r
        elif aux == "mhe_loss":
            wn_unit = wn if kind != 0 else ops.loss_prep_weight(w, True)[1]      # MHE normalises the weights itself (loss.py:1026)
            total = total + ops.mhe_loss(wn_unit, num_outputs, y, float(params.mhe_lambda))
            endpoints["w"] = w
        else:
            raise NotImplementedError("Unsupported loss function %s" % aux)
    return total, endpoints


def softmax(features, labels, num_outputs, params, is_training=None, reuse_variables=None, name="softmax"):
    return _run(0, features, labels, num_outputs, params, reuse_variables, name, 0.0, 0.0, True)


def asoftmax(features, labels, num_outputs, params, is_training=None, reuse_variables=None, name="softmax"):
    m = int(params.asoftmax_m)
    if m not in (1, 2, 4):
        raise NotImplementedError("[ERROR] m=%d is not unsupported." % m)
    return _run(1, features, labels, num_outputs, params, reuse_variables, name, float(m), _lambda(params, "asoftmax"), False)


def additive_margin_softmax(features, labels, num_outputs, params, is_training=None, reuse_variables=None, name="softmax"):
    params.amsoftmax_m = float(params.amsoftmax_m)
    return _run(2, features, labels, num_outputs, params, reuse_variables, name, params.amsoftmax_m,
                _lambda(params, "amsoftmax"), False)


def additive_angular_margin_softmax(features, labels, num_outputs, params, is_training=None, reuse_variables=None,
                                    name="softmax"):
    params.arcsoftmax_m = float(params.arcsoftmax_m)
    return _run(3, features, labels, num_outputs, params, reuse_variables, name, params.arcsoftmax_m,
                _lambda(params, "arcsoftmax"), False)


def _not_hot_path(*a, **k):
    raise NotImplementedError("triplet / end-to-end losses are outside the hot path (no shipped config selects them, SURVEY.md section 2)")


semihard_triplet_loss = angular_triplet_loss = e2e_valid_loss = generalized_angular_triplet_loss = _not_hot_path
