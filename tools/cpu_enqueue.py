"""How long does the host take to ENQUEUE one optimiser step (vs how long the GPU takes to run it)?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_kaldi_speaker_amd import engine as E
B, T, D, N = 128, 200, 30, 7351
cfg = E.make_config(D, N, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, max_batch=B, max_frames=T)
eng = E.Engine(cfg, device="cuda:0"); eng.init_variables(seed=0)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(B, T, D).astype(np.float32)).cuda(); y = torch.from_numpy(rs.randint(0, N, B).astype(np.int32)).cuda()
for i in range(10): eng.train_step(x, y, 0.01, i)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
parts = np.zeros(4)
for i in range(n):
    a = time.perf_counter(); eng.forward(x, True); b = time.perf_counter(); eng.loss(y, i, True); c = time.perf_counter()
    eng.backward(-1); d = time.perf_counter(); eng.apply(0.01, 1.0); e = time.perf_counter()
    parts += [b - a, c - b, d - c, e - d]
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.3f ms/step (forward %.3f, loss %.3f, backward %.3f, apply %.3f); GPU done after %.3f ms/step" %
      ((t1 - t0) / n * 1e3, *(parts / n * 1e3), (t2 - t0) / n * 1e3))
