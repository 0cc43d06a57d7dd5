"""Diagnostic (not a test): print the relative error of every endpoint / gradient of the
engine against the float64 oracle for one small training step."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import xvector_oracle as O
from tests.test_gpu_engine import _make, rel_err

kw = dict(loss_func=sys.argv[1] if len(sys.argv) > 1 else "softmax")
if kw["loss_func"] != "softmax":
    kw.update(margin_m=0.2, last_layer_linear=True)
B, T = 6, 40
eng, cfg_o, V = _make(kw, B, T)
rs = np.random.RandomState(42)
x = rs.randn(B, T, 30).astype(np.float32)
labels = rs.randint(0, cfg_o.num_speakers, B).astype(np.int32)
newV, _, info = O.train_step(V, {}, cfg_o, x.astype(np.float64), labels, 0.05, 1234)
eng.forward(x, True); eng.loss(labels, 1234, True); eng.backward(-1)
print("loss", eng.losses(), info["raw_loss"], info["reg_loss"])
for name in info["endpoints"]:
    try:
        got = eng.endpoint(name).cpu().numpy()
    except Exception as e:
        continue
    print("EP %-14s %.2e" % (name, rel_err(got, info["endpoints"][name].reshape(got.shape))))
for name, g in eng.get_gradients().items():
    print("GR %-30s %.2e" % (name, rel_err(g, info["grads"][name].reshape(g.shape))))
