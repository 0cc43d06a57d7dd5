"""A config's hot-path keys (reference nnet_conf/*.json) in the oracle's vocabulary - the tests' own reading,
independent of tf_kaldi_speaker_amd/model/tdnn.py::engine_config, which builds the engine side from the same dict."""


def oracle_kw_from_params(d):
    """The test's own reading of a config's hot-path keys in the oracle's vocabulary - independent of
    model/tdnn.py::engine_config, which builds the engine side from the same dict."""
    kw = dict(loss_func=d["loss_func"], last_layer_no_bn=bool(d.get("last_layer_no_bn", False)),
              last_layer_linear=bool(d.get("last_layer_linear", False)), feature_norm=bool(d.get("feature_norm", False)),
              weight_l2_regularizer=d["weight_l2_regularizer"], batchnorm_momentum=d["batchnorm_momentum"],
              optimizer=d.get("optimizer", "sgd"), pooling_type=d["pooling_type"])
    if kw["feature_norm"]:
        kw["feature_scaling_factor"] = d["feature_scaling_factor"]
    if kw["optimizer"] == "momentum":
        kw.update(momentum=d["momentum"], use_nesterov=bool(d.get("use_nesterov", False)))
    prefix = {"asoftmax": "asoftmax", "additive_margin_softmax": "amsoftmax", "additive_angular_margin_softmax": "arcsoftmax"}.get(d["loss_func"])
    if prefix:
        kw.update(margin_m=d[prefix + "_m"], lambda_min=d[prefix + "_lambda_min"], lambda_base=d[prefix + "_lambda_base"],
                  lambda_gamma=d[prefix + "_lambda_gamma"], lambda_power=d[prefix + "_lambda_power"])
    if d["pooling_type"] == "self_attention":
        kw.update(att_key_num_nodes=tuple(d["att_key_num_nodes"]), att_key_network_type=d["att_key_network_type"],
                  att_use_scale=bool(d.get("att_use_scale", False)))
    if d.get("aux_loss_func"):
        kw["aux_loss_func"] = tuple(d["aux_loss_func"])
        for k in ("ring_loss_init", "ring_loss_lambda", "mhe_lambda"):
            if k in d:
                kw[k] = d[k]
    return kw
