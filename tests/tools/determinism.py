"""Two engines, same seed, same batches (variable lengths): every variable must be bit-identical after N steps, in both precisions."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tf_kaldi_speaker_amd import engine as E
def run(prec, steps=60):
    cfg = E.make_config(30, 7351, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, max_batch=128, max_frames=400, precision=prec)
    eng = E.Engine(cfg, device="cuda:0"); eng.init_variables(seed=3)
    rs = np.random.RandomState(11)
    for i in range(steps):
        b = [128, 64, 96][i % 3]; t = int(rs.randint(200, 401))
        x = torch.from_numpy(rs.randn(b, t, 30).astype(np.float32)).cuda(); y = torch.from_numpy(rs.randint(0, 7351, b).astype(np.int32)).cuda()
        eng.train_step(x, y, 0.01, i)
    torch.cuda.synchronize()
    v = {k: np.array(a, copy=True) for k, a in eng.get_variables().items()}
    eng.close(); return v
for prec in ("f32", "f16x3"):
    a, b = run(prec), run(prec)
    bad = [k for k in a if not np.array_equal(a[k], b[k])]
    fin = all(np.isfinite(a[k]).all() for k in a)
    print(prec, "bit-identical" if not bad else "DIFFERENT: %s" % bad[:5], "finite" if fin else "NOT FINITE")
