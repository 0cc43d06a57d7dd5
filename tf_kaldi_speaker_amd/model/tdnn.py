"""tdnn(features, params, is_training, reuse_variables, aux_features) -> (features, endpoints)
with the reference's endpoint names and order (model/tdnn.py:8-191), executed by the native engine.

The variables of the "tdnn" scope are owned by an engine held in a module-level graph (the
analogue of TF's default graph): the first call creates them (Glorot-uniform, seed = params.seed),
`reuse_variables=True` re-enters the same variables, a second creating call raises like TF does.
Trainer builds its own engine (with the loss head) and does not go through this module-level graph.
"""
from collections import OrderedDict

import torch

try:
    from .. import engine as E
    from .common import to_device
except (ImportError, ValueError):
    import engine as E
    from model.common import to_device

_GRAPH = {"engine": None, "key": None}

# network_type "extended_tdnn": the layer recipe of tdnn() on a caller-supplied table of (context, width) frame layers
# (params.tdnn_layers).  It has NO counterpart in the reference (model/tdnn.py:39-189 hard-codes 5 + 2 layers; BASELINE configs[4]
# "extended context, 10 layers"); the default is the 10-layer shape of the extended x-vector recipe with contiguous contexts.
DEFAULT_EXTENDED_LAYERS = ((5, 512), (1, 512), (3, 512), (1, 512), (3, 512), (1, 512), (3, 512), (1, 512), (1, 512), (1, None))


def frame_layer_table(params):
    """None for the reference network, else [(context, width)] with the pooling width filled in."""
    if params.dict.get("network_type", "tdnn") != "extended_tdnn":
        return None
    table = params.dict.get("tdnn_layers") or DEFAULT_EXTENDED_LAYERS
    pool = int(params.dict.get("num_nodes_pooling_layer", 1500))
    return tuple((int(k), int(w) if w else pool) for k, w in table)


def endpoint_order(table=None):
    """Endpoint names in the reference's insertion order (tdnn.py:45-189) for a frame-layer table (None = the reference's)."""
    table = table or ((5, 512), (5, 512), (7, 512), (1, 512), (1, 1500))
    names = []
    for i, (k, _) in enumerate(table):
        names += ["tdnn%d_%s" % (i + 1, "conv" if k > 1 else "dense"), "tdnn%d_bn" % (i + 1), "tdnn%d_relu" % (i + 1)]
    # self_attention only (pooling.py:78-149); absent names are skipped for statistics pooling
    names += ["att_key0_dense", "att_key0_bn", "att_key0_relu", "att_key1_dense", "att_key1_bn", "att_key1_relu", "attention_weights", "pooling"]
    n = len(table)
    names += ["tdnn%d_dense" % (n + 1), "tdnn%d_bn" % (n + 1), "tdnn%d_relu" % (n + 1), "tdnn%d_dense" % (n + 2), "tdnn%d_bn" % (n + 2),
              "tdnn%d_relu" % (n + 2)]
    return names


ENDPOINT_ORDER = endpoint_order()


def reset_default_graph():
    if _GRAPH["engine"] is not None:
        _GRAPH["engine"].close()
    _GRAPH["engine"], _GRAPH["key"] = None, None


def check_params(params):
    """The static checks of model/tdnn.py:24-30,111-113,133-142,162-184 (defaults are inserted into params)."""
    if params.dict.get("network_relu_type", "relu") not in ("relu", "prelu", "lrelu"):
        raise NotImplementedError("network_relu_type %s (relu, prelu or lrelu: tdnn.py:24-30)" % params.network_relu_type)
    if "num_nodes_pooling_layer" not in params.dict:
        params.dict["num_nodes_pooling_layer"] = 1500
    if params.pooling_type == "self_attention":
        # model/pooling.py:37-192 in the form the shipped attention config uses; everything else the function offers
        # (value network, several heads, split keys, penalty term, post non-linearity) is refused by name
        d = params.dict
        unsupported = []
        nf = len(frame_layer_table(params) or ()) or 5       # key = the last-but-one frame layer, value = the last (tdnn4_relu / tdnn5_relu)
        if d.get("att_key_input") != "tdnn%d_relu" % (nf - 1):
            unsupported.append("att_key_input=%r (tdnn%d_relu)" % (d.get("att_key_input"), nf - 1))
        if d.get("att_value_input") != "tdnn%d_relu" % nf:
            unsupported.append("att_value_input=%r (tdnn%d_relu)" % (d.get("att_value_input"), nf))
        if len(d.get("att_value_num_nodes", [])) != 0:
            unsupported.append("att_value_num_nodes (no value network)")
        if len(d.get("att_key_num_nodes", [])) != 2:
            unsupported.append("att_key_num_nodes (two key layers)")
        if int(d.get("att_key_network_type", -1)) not in (0, 1, 2, 3):
            unsupported.append("att_key_network_type=%r (0..3)" % d.get("att_key_network_type"))
        if int(d.get("att_key_network_type", -1)) == 1 and d.get("network_relu_type", "relu") != "relu":
            unsupported.append("att_key_network_type 1 with network_relu_type %s (the score kernel's fused ReLU is a plain one)" % d["network_relu_type"])
        if int(d.get("att_num_heads", 1)) != 1 or d.get("att_split_key", False):
            unsupported.append("att_num_heads / att_split_key (one head)")
        if float(d.get("att_penalty_term", 0) or 0) != 0.0:
            unsupported.append("att_penalty_term (0)")
        if d.get("att_apply_nonlinear", False):
            unsupported.append("att_apply_nonlinear (false)")
        if unsupported:
            raise NotImplementedError("self_attention on the MI355X engine supports the shipped single-head form only; not: "
                                      + ", ".join(unsupported))
    elif params.pooling_type != "statistics_pooling":
        if params.pooling_type == "ghost_vlad":
            raise NotImplementedError("Not implement %s pooling on the MI355X engine yet" % params.pooling_type)
        raise NotImplementedError("Not implement %s pooling" % params.pooling_type)
    if "num_nodes_last_layer" not in params.dict:
        params.dict["num_nodes_last_layer"] = 512
    if "last_layer_no_bn" not in params.dict:
        params.last_layer_no_bn = False
    if "last_layer_linear" not in params.dict:
        params.last_layer_linear = False


def engine_config(params, dim, num_speakers=0, loss_type="softmax", max_batch=128, max_frames=400, max_rows=0):
    check_params(params)
    d = params.dict
    kw = dict(num_nodes_pooling_layer=d["num_nodes_pooling_layer"], num_nodes_last_layer=d["num_nodes_last_layer"],
              last_layer_no_bn=d["last_layer_no_bn"], last_layer_linear=d["last_layer_linear"],
              feature_norm=bool(d.get("feature_norm", False)), feature_scaling_factor=float(d.get("feature_scaling_factor", 1.0)),
              weight_l2_regularizer=float(d["weight_l2_regularizer"]),
              output_weight_l2_regularizer=d.get("output_weight_l2_regularizer", None),
              batchnorm_momentum=float(d["batchnorm_momentum"]), optimizer=d.get("optimizer", "sgd"),
              momentum=float(d.get("momentum", 0.0) or 0.0), use_nesterov=bool(d.get("use_nesterov", False)),
              clip_gradient_norm=float(d["clip_gradient_norm"]) if d.get("clip_gradient", False) else 0.0,
              max_batch=max_batch, max_frames=max_frames, max_rows=max_rows,
              frame_layers=frame_layer_table(params), network_relu_type=d.get("network_relu_type", "relu"),
              precision=d.get("precision", None),      # engine extension: "f32" (default) | "f16x3" (opt-in fast mode); absent from reference configs
              pooling_type=d["pooling_type"])
    if num_speakers and d.get("aux_loss_func"):      # loss.py:985-1036; every loss function adds them (loss.py:40,161,249,347)
        kw.update(aux_loss_func=tuple(d["aux_loss_func"]))
        if "ring_loss" in d["aux_loss_func"]:
            kw.update(ring_loss_init=float(d["ring_loss_init"]), ring_loss_lambda=float(d["ring_loss_lambda"]))
        if "mhe_loss" in d["aux_loss_func"]:
            kw.update(mhe_lambda=float(d["mhe_lambda"]))
    if d["pooling_type"] == "self_attention":
        kw.update(att_key_num_nodes=tuple(d["att_key_num_nodes"]), att_key_network_type=int(d["att_key_network_type"]),
                  att_use_scale=bool(d.get("att_use_scale", False)))
    if d.get("feature_norm", False):
        assert "feature_scaling_factor" in d, "If feature normalization is applied, scaling factor is necessary."
    prefix = {"asoftmax": "asoftmax", "additive_margin_softmax": "amsoftmax", "additive_angular_margin_softmax": "arcsoftmax"}.get(loss_type)
    if prefix is not None and num_speakers:
        kw.update(margin_m=float(d[prefix + "_m"]), lambda_min=float(d[prefix + "_lambda_min"]),
                  lambda_base=float(d[prefix + "_lambda_base"]), lambda_gamma=float(d[prefix + "_lambda_gamma"]),
                  lambda_power=float(d[prefix + "_lambda_power"]))
    return E.make_config(dim, num_speakers, loss_func=loss_type if num_speakers else "softmax", **kw)


def collect_endpoints(eng, b, names=None):
    """OrderedDict of device tensors in the reference's insertion order; frame-level ones are [B,T_l,C]."""
    out = OrderedDict()
    if names is None:
        nf = int(eng.config.num_frame_layers)
        names = endpoint_order([(int(eng.config.frame_context[i]), int(eng.config.frame_width[i])) for i in range(nf)] if nf else None)
    for name in names:
        try:
            t = eng.endpoint(name)
        except Exception:
            continue
        if t.shape[0] != b:
            t = t.view(b, t.shape[0] // b, t.shape[1])
        out[name] = t
    return out


def tdnn(features, params, is_training=None, reuse_variables=None, aux_features=None):
    x = to_device(features)
    assert x.dim() == 3, "features must be [batch, length, dim]"
    b, t, dim = x.shape
    check_params(params)
    # everything that changes the graph tdnn() builds: two graphs in one process that differ in any of these must not share an engine
    key = (dim, params.dict["num_nodes_pooling_layer"], params.dict["num_nodes_last_layer"],
           bool(params.last_layer_no_bn), bool(params.last_layer_linear), frame_layer_table(params),
           params.dict.get("network_relu_type", "relu"), params.dict.get("pooling_type", "statistics_pooling"),
           tuple(sorted((k, repr(v)) for k, v in params.dict.items() if k.startswith("att_"))))
    eng = _GRAPH["engine"]
    if eng is None:
        if reuse_variables is True:
            raise ValueError("Variable tdnn/tdnn1_conv/kernel does not exist, or was not created with tf.get_variable().")
        eng = E.Engine(engine_config(params, dim, 0, "softmax", max_batch=max(b, 1), max_frames=max(t, 15)))
        eng.init_variables(seed=int(params.dict.get("seed", 0)))
        _GRAPH["engine"], _GRAPH["key"] = eng, key
    else:
        if not reuse_variables:
            raise ValueError("Variable tdnn/tdnn1_conv/kernel already exists, disallowed. Did you mean to set reuse=True?")
        assert key == _GRAPH["key"], "tdnn() re-entered with a different architecture"
        if b > eng.config.max_batch or t > eng.config.max_frames:     # grow: new capacity, same variables
            values = eng.get_variables()
            eng.close()
            eng = E.Engine(engine_config(params, dim, 0, "softmax", max_batch=max(b, eng.config.max_batch),
                                         max_frames=max(t, eng.config.max_frames)))
            eng.set_variables(values)
            _GRAPH["engine"] = eng
    eng.forward(x, bool(is_training))
    endpoints = collect_endpoints(eng, b)
    nf = len(frame_layer_table(params) or ()) or 5
    last = [k for k in endpoints if k.startswith("tdnn%d_" % (nf + 2))][-1]
    return endpoints[last], endpoints


def extended_tdnn(features, params, is_training=None, reuse_variables=None, aux_features=None):
    """tdnn() on params.tdnn_layers (network_type "extended_tdnn"); same signature, same endpoint naming scheme."""
    if params.dict.get("network_type") != "extended_tdnn":
        params.dict["network_type"] = "extended_tdnn"
    return tdnn(features, params, is_training, reuse_variables, aux_features)
