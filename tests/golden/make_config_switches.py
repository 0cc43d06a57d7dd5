#!/usr/bin/env python3
"""Derives tests/golden/shipped_config_switches.json: the distinct combinations of graph / loss / optimiser switches among
the reference's 81 single-task nnet_conf/*.json (voxceleb, sre, fisher recipes), each with how many files use it and one
example path.  Only the keys the hot path reads are kept; data-pipeline keys (batch sizes, epochs,
learning-rate schedule, segment lengths) are dropped, which is what folds 81 files into a handful of combinations.
Run in the build container (needs /root/reference); tests/test_gpu_engine.py runs one parity step per combination."""
import glob
import json
import os

REF = "/root/reference/egs"
KEYS = ["network_type", "loss_func", "pooling_type", "last_layer_no_bn", "last_layer_linear", "feature_norm", "feature_scaling_factor",
        "num_nodes_pooling_layer", "num_nodes_last_layer", "weight_l2_regularizer", "output_weight_l2_regularizer", "batchnorm_momentum",
        "optimizer", "momentum", "use_nesterov", "clip_gradient", "clip_gradient_norm", "network_relu_type",
        "att_key_input", "att_key_num_nodes", "att_key_network_type", "att_value_input", "att_value_num_nodes", "att_value_network_type",
        "att_apply_nonlinear", "att_use_scale", "att_num_heads", "att_split_key", "att_penalty_term",
        "aux_loss_func", "ring_loss_init", "ring_loss_lambda", "mhe_lambda"]
LOSS_PREFIX = {"asoftmax": "asoftmax", "additive_margin_softmax": "amsoftmax", "additive_angular_margin_softmax": "arcsoftmax"}


def main():
    combos = {}
    files = sorted(glob.glob(os.path.join(REF, "*", "*", "nnet_conf", "*.json")))
    for f in files:
        d = json.load(open(f))
        if "network_type" not in d:      # multitask configs: outside the hot path
            continue
        keep = {k: d[k] for k in KEYS if k in d}
        prefix = LOSS_PREFIX.get(d["loss_func"])
        if prefix:                        # only the margin family the config actually selects
            keep.update({k: v for k, v in d.items() if k.startswith(prefix + "_")})
        if d.get("pooling_type") != "self_attention":
            keep = {k: v for k, v in keep.items() if not k.startswith("att_")}
        if not d.get("aux_loss_func"):
            keep = {k: v for k, v in keep.items() if k not in ("ring_loss_init", "ring_loss_lambda", "mhe_lambda")}
        key = json.dumps(keep, sort_keys=True)
        c = combos.setdefault(key, {"params": keep, "count": 0, "example": os.path.relpath(f, "/root/reference")})
        c["count"] += 1
    out = sorted(combos.values(), key=lambda c: -c["count"])
    assert sum(c["count"] for c in out) == 81, sum(c["count"] for c in out)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shipped_config_switches.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("%d single-task configs -> %d distinct switch combinations -> %s" % (81, len(out), path))


if __name__ == "__main__":
    main()
