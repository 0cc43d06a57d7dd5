#!/usr/bin/env python
"""Golden vectors FROM THE REFERENCE ITSELF (TensorFlow 1.x) for the part of the hot path nothing else pins:
conv / BatchNorm / dense / statistics pooling / the optimisers / the BN moving statistics.

CANNOT RUN IN THE BUILD CONTAINER (no TensorFlow, no wheel, no network): run it once on any box that has the reference's
environment (README.md:23,31-34: Python 2.7 or 3.x + TensorFlow 1.12 ... 1.15, CPU is enough), then commit the two outputs:

    export PYTHONPATH=/path/to/tf-kaldi-speaker          # the reference checkout
    python tests/golden/make_tf_golden.py                 # -> tests/golden/tf_golden.npz, tests/golden/tf_golden_ckpt.*

`tests/test_oracle_tf_golden.py` then compares oracle/xvector_oracle.py with these vectors (it skips while the file is absent), which
lifts the "parity unpinned vs TF1" status of the oracle in one command; `tests/test_tf_checkpoint.py` checks the TensorFlow-free
checkpoint reader against the Saver-written `tf_golden_ckpt.*`.

What is dumped, per case (a seeded 4 x 40 x 30 batch, 13 speakers; graph built by the reference's own Trainer.build("train"),
model/trainer.py:190-449, i.e. model/tdnn.py:33-191 + model/loss.py + the optimiser of trainer.py:332-346):
    <case>/params            the JSON the Trainer was built from
    <case>/x, y, lr, step    the fed batch, learning rate and global_step
    <case>/var0/<name>       every global variable before the step (BN parameters / biases moved off their trivial initial values)
    <case>/ep/<endpoint>     every float endpoint of the training graph (tdnn*_conv/_bn/_relu, pooling, tdnn6_dense, ..., logits)
    <case>/raw_loss, total_loss
    <case>/grad/<name>       d total_loss / d variable for every trainable variable (tf.gradients on trainer.total_loss)
    <case>/var1/<name>       every model variable after ONE sess.run(train_op)   (optimiser step + BN moving-average update)
    <case>/var2/<name>       ... after a SECOND one on the same batch (momentum / Adam slot state, global_step + 1)
    <case>/emb_after         the predict graph's embedding (is_training=False: moving statistics) of x after the two steps
    <case>/attempt           which data seed was used: a batch that puts any ReLU input within KINK of zero in either step is thrown away and
                             redrawn (case_rng), because there the sign - and with it every gradient upstream - is decided by fp32 rounding,
                             and no tolerance makes that comparison meaningful (about one element in a million lands there)
Only data is written: no reference source text.
"""
from __future__ import print_function

import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

BASE = {
    "seed": 0, "network_type": "tdnn", "last_layer_no_bn": False, "last_layer_linear": False, "feature_norm": False,
    "pooling_type": "statistics_pooling", "embedding_node": "tdnn6_dense", "batch_type": "softmax",
    "weight_l2_regularizer": 1e-2, "batchnorm_momentum": 0.99, "clip_gradient": False, "clip_gradient_norm": 3,
    "keep_checkpoint_max": 3, "use_nesterov": False,
}
MARGIN = {"lambda_min": 0, "lambda_base": 1000, "lambda_gamma": 0.0001, "lambda_power": 5}


def margin(prefix, m, **kw):
    d = dict(MARGIN, **kw)
    out = {prefix + "_m": m}
    out.update({prefix + "_" + k: v for k, v in d.items()})
    return out


CASES = [
    ("softmax_sgd", dict(BASE, loss_func="softmax", optimizer="sgd")),
    ("amsoftmax_momentum", dict(BASE, loss_func="additive_margin_softmax", last_layer_linear=True, optimizer="momentum", momentum=0.9,
                                **margin("amsoftmax", 0.2))),
    ("arcsoftmax_adam_fn", dict(BASE, loss_func="additive_angular_margin_softmax", last_layer_linear=True, optimizer="adam",
                                feature_norm=True, feature_scaling_factor=30, **margin("arcsoftmax", 0.3, lambda_gamma=0.00001))),
    ("asoftmax_m4_nesterov", dict(BASE, loss_func="asoftmax", last_layer_linear=True, optimizer="momentum", momentum=0.9, use_nesterov=True,
                                  **margin("asoftmax", 4, lambda_min=10, lambda_gamma=0.00001))),
    ("softmax_nobn_last", dict(BASE, loss_func="softmax", optimizer="sgd", last_layer_no_bn=True)),
]
B, T, D, N = 4, 40, 30, 13
STEP, LR = 1234, 0.05
KINK = 2e-6                 # |ReLU input| below this counts as "on the kink": fp32 errors of these endpoints are <= 1.7e-6 (measured on the oracle)


def case_rng(name, attempt):
    """The batch / perturbation generator of a case: a seed that does not depend on PYTHONHASHSEED, redrawn per attempt."""
    return np.random.RandomState(sum(bytearray(name.encode())) + 1000 * attempt)


def kinks(endpoints, thr=KINK):
    """How many ReLU inputs of a {name: array} endpoint dictionary lie within thr of zero.  The input of <layer>_relu is <layer>_bn where
    the layer has a BatchNorm, else its _dense / _conv output (model/tdnn.py:60-152 names them so)."""
    n = 0
    for k in endpoints:
        if k.endswith("_relu"):
            pre = [k[:-5] + s for s in ("_bn", "_dense", "_conv") if k[:-5] + s in endpoints]
            if pre:
                n += int((np.abs(np.asarray(endpoints[pre[0]])) < thr).sum())
    return n


def run_case(tf, Trainer, Params, name, cfg, workdir, out, save_ckpt, attempt=0):
    """One case into `out`; returns the number of ReLU inputs on the kink over the two steps (main() redraws the batch while it is > 0)."""
    tf.reset_default_graph()
    case_dir = os.path.join(workdir, "%s_%d" % (name, attempt))
    os.makedirs(os.path.join(case_dir, "nnet"))
    json_path = os.path.join(case_dir, "config.json")
    with open(json_path, "w") as f:
        json.dump(cfg, f)
    params = Params(json_path)
    trainer = Trainer(params, case_dir)
    trainer.build("train", dim=D, loss_type=params.loss_func, num_speakers=N)
    trainer.build("predict", dim=D)
    sess = trainer.sess
    sess.run(tf.global_variables_initializer())
    sess.run(tf.local_variables_initializer())
    rs = case_rng(name, attempt)
    model_vars = [v for v in tf.global_variables() if "optimizer" not in v.op.name and not v.op.name.endswith(("/Momentum", "/Adam", "/Adam_1"))
                  and v.op.name not in ("beta1_power", "beta2_power")]
    # BN parameters, biases and moving statistics off their trivial initial values (so that every term of the arithmetic shows)
    for v in model_vars:
        n, val = v.op.name, sess.run(v)
        if n.endswith(("gamma", "beta", "bias")):
            val = val + 0.1 * rs.randn(*val.shape).astype(np.float32)
        elif n.endswith("moving_mean"):
            val = (0.2 * rs.randn(*val.shape)).astype(np.float32)
        elif n.endswith("moving_variance"):
            val = (0.5 + rs.rand(*val.shape)).astype(np.float32)
        else:
            continue
        v.load(val, sess)
    x = rs.randn(B, T, D).astype(np.float32)
    y = rs.randint(0, N, B).astype(np.int32)
    feed = {trainer.train_features: x, trainer.train_labels: y, trainer.global_step: STEP, trainer.learning_rate: LR}
    key = lambda *p: "/".join((name,) + p)      # noqa: E731
    out[key("params")] = np.array(json.dumps(cfg))
    out[key("x")], out[key("y")], out[key("lr")], out[key("step")] = x, y, np.float64(LR), np.int64(STEP)
    out[key("attempt")] = np.int64(attempt)
    for v in model_vars:
        out[key("var0", v.op.name)] = sess.run(v)
    trainable = tf.trainable_variables()
    grads = tf.gradients(trainer.total_loss, trainable)
    ep_names = [k for k, t in trainer.endpoints.items() if hasattr(t, "dtype") and t.dtype.is_floating]
    res = sess.run({"ep": [trainer.endpoints[k] for k in ep_names], "total": trainer.train_ops["loss"], "raw": trainer.train_ops["raw_loss"],
                    "grads": [g for g in grads if g is not None]}, feed_dict=feed)
    for k, val in zip(ep_names, res["ep"]):
        out[key("ep", k)] = val
    on_kink = kinks(dict(zip(ep_names, res["ep"])))
    out[key("raw_loss")], out[key("total_loss")] = np.float64(res["raw"]), np.float64(res["total"])
    gi = iter(res["grads"])
    for v, g in zip(trainable, grads):
        if g is not None:
            out[key("grad", v.op.name)] = np.asarray(next(gi))
    sess.run(trainer.train_op, feed_dict=feed)
    for v in model_vars:
        out[key("var1", v.op.name)] = sess.run(v)
    feed[trainer.global_step] = STEP + 1
    on_kink += kinks(dict(zip(ep_names, sess.run([trainer.endpoints[k] for k in ep_names], feed_dict=feed))))      # the second step's forward
    sess.run(trainer.train_op, feed_dict=feed)
    for v in model_vars:
        out[key("var2", v.op.name)] = sess.run(v)
    out[key("emb_after")] = sess.run(trainer.embeddings, feed_dict={trainer.pred_features: x})
    if save_ckpt:
        # a small Saver-written checkpoint for tests/test_tf_checkpoint.py: a few variables of this graph, V2 format (the Saver default)
        few = [v for v in model_vars if "tdnn1" in v.op.name or "tdnn7" in v.op.name]
        saver = tf.train.Saver(var_list=few)
        prefix = saver.save(sess, os.path.join(case_dir, "tf_golden_ckpt"), write_meta_graph=False)
        for ext in (".index", ".data-00000-of-00001"):
            shutil.copyfile(prefix + ext, os.path.join(HERE, "tf_golden_ckpt" + ext))
        np.savez(os.path.join(HERE, "tf_golden_ckpt_values.npz"), **{v.op.name: sess.run(v) for v in few})
    trainer.close()
    return on_kink


def main():
    try:
        import tensorflow as tf
    except ImportError:
        sys.exit("make_tf_golden.py needs TensorFlow 1.x (the reference's environment, README.md:31-34); it cannot run in the build container")
    if not tf.__version__.startswith("1."):
        sys.exit("TensorFlow %s: the reference graph code (tf.layers, tf.contrib) needs TensorFlow 1.x" % tf.__version__)
    from misc.utils import Params              # the REFERENCE's modules: PYTHONPATH=<reference checkout>
    from model.trainer import Trainer
    tf.logging.set_verbosity(tf.logging.WARN)
    out = {"__tensorflow_version__": np.array(tf.__version__), "__cases__": np.array(json.dumps([c[0] for c in CASES]))}
    workdir = tempfile.mkdtemp(prefix="tf_golden_")
    try:
        for i, (name, cfg) in enumerate(CASES):
            for attempt in range(200):
                n = run_case(tf, Trainer, Params, name, cfg, workdir, out, save_ckpt=(i == 0), attempt=attempt)
                print("case", name, "attempt", attempt, "ReLU inputs on the kink:", n)
                if n == 0:
                    break
                for k in [k for k in out if k.startswith(name + "/")]:
                    del out[k]
            else:
                sys.exit("no kink-free batch for case %s in 200 attempts" % name)
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, "tf_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "tf_golden.npz"), "(%d arrays)" % len(out))


if __name__ == "__main__":
    main()
