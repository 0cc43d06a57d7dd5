import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xvector_oracle as O
from tests.test_gpu_engine import _make, rel_err
kw = dict(loss_func="softmax")
B, T = 6, 40
eng, cfg, V = _make(kw, B, T)
rs = np.random.RandomState(42)
x = rs.randn(B, T, 30).astype(np.float32)
labels = rs.randint(0, cfg.num_speakers, B).astype(np.int32)
x64 = x.astype(np.float64)
feats, ep, caches = O.tdnn_forward(V, x64, cfg, True, {})
loss, logits, dfeat, Gl = O.loss_forward_backward(V, cfg, feats, labels, 1234)
# manual backward down to da5
d = O.relu_bwd(ep["tdnn7_relu"], dfeat)
d, _, _ = O.batchnorm_train_bwd(d, caches["tdnn7_bn"], V["tdnn/tdnn7_bn/gamma"])
d, _, _ = O.dense_bwd(caches["tdnn7_dense_in"], V["tdnn/tdnn7_dense/kernel"], d)
d = O.relu_bwd(ep["tdnn6_relu"], d)
d, _, _ = O.batchnorm_train_bwd(d, caches["tdnn6_bn"], V["tdnn/tdnn6_bn/gamma"])
dpool, _, _ = O.dense_bwd(caches["tdnn6_dense_in"], V["tdnn/tdnn6_dense/kernel"], d)
da5 = O.statistics_pooling_bwd(caches["pool_in"], caches["pool"], dpool)
eng.forward(x, True); eng.loss(labels, 1234, True); eng.backward(0)
g = eng.endpoint("debug:dpool").cpu().numpy()
print("dpool", rel_err(g, dpool), "mean half", rel_err(g[:, :1500], dpool[:, :1500]), "std half", rel_err(g[:, 1500:], dpool[:, 1500:]))
g = eng.endpoint("debug:da5").cpu().numpy().reshape(da5.shape)
print("da5", rel_err(g, da5))
dd = np.abs(g - da5); i = np.unravel_index(dd.argmax(), dd.shape)
pool = ep["pooling"]
print("worst", i, g[i], da5[i], "std", pool[i[0], 1500 + i[2]], "mask", caches["pool"][2][i[0], i[2]], "masked frac", caches["pool"][2].mean())
gp = eng.endpoint("pooling").cpu().numpy()
print("gpu std there", gp[i[0], 1500 + i[2]], "a5 col", caches["pool_in"][i[0], :, i[2]])
print("---- full backward")
eng.forward(x, True); eng.loss(labels, 1234, True); eng.backward(-1)
G = eng.get_gradients()
_, _, info = O.train_step(V, {}, cfg, x64, labels, 0.05, 1234)
gb, rb = G["tdnn/tdnn5_bn/beta"], info["grads"]["tdnn/tdnn5_bn/beta"]
dd = np.abs(gb - rb); c = dd.argmax()
print("dbeta5 worst channel", c, gb[c], rb[c], "n bad (>1e-4 rel)", (dd > 1e-4 * np.abs(rb).max()).sum())
y5 = ep["tdnn5_bn"].reshape(-1, 1500)[:, c]
print("y5 min |y|", np.abs(y5).min(), "count y>0", (y5 > 0).sum())
order = np.argsort(np.abs(y5))[:5]
print("smallest |y5|", y5[order])
da = da5.reshape(-1, 1500)[:, c]
print("da5 at those", da[order], "sum dy", (da * (y5 > 0)).sum())
yg = eng.endpoint("tdnn5_bn").cpu().numpy()[:, c]
print("gpu y5 at those", yg[order])
