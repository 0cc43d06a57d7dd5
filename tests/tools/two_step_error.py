"""Diagnostic (not a test): the relative error of a few variables against the float64 oracle after two optimiser steps, per precision
(what tests/test_gpu_engine.py::test_second_step_uses_updated_weights_and_staged_backward bounds at 1e-4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.test_gpu_engine import _make, oracle_step_with_gpu_relu_pattern

for prec in ("f32", "f16x3"):
    os.environ["XV_PRECISION"] = prec
    kw = dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True)
    B, T = int(os.environ.get('XV_DIAG_B', '6')), 33
    eng, cfg_o, V = _make(kw, B, T)
    rs = np.random.RandomState(1)
    opt = {}
    for it in range(2):
        x = rs.randn(B, T, 30).astype(np.float32)
        labels = rs.randint(0, cfg_o.num_speakers, B).astype(np.int32)
        eng.forward(x, True); eng.loss(labels, it, True); eng.backward(-1)
        V, opt, info = oracle_step_with_gpu_relu_pattern(eng, V, cfg_o, x.astype(np.float64), labels, 0.1, it, opt)
        eng.apply(0.1, 1.0)
    after = eng.get_variables()
    print(prec, " ".join("%s %.2e" % (n.split("/")[-2], np.abs(after[n] - V[n]).max() / np.abs(V[n]).max())
                         for n in ("tdnn/tdnn2_conv/kernel", "tdnn/tdnn5_dense/kernel", "tdnn/tdnn6_dense/kernel", "softmax/output/kernel", "tdnn/tdnn5_bn/gamma")))
    eng.close()
