"""Soak run: N optimiser steps fed by the native loader; reports rate, host RSS and free device memory at the start and the end
(leak check) and that the loss stayed finite."""
import os, sys, time, tempfile, resource
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.kaldi_fixture import make_data_dir
from tf_kaldi_speaker_amd import engine as E
from tf_kaldi_speaker_amd.dataset.native_loader import NativeRandomQueue

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
root = tempfile.mkdtemp(prefix="xv_soak_")
root, spklist, _ = make_data_dir(root, num_spk=100, utts_per_spk=8, dim=30, min_frames=500, max_frames=1200, seed=0)
N = 7351
eng = E.Engine(E.make_config(30, N, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, max_batch=128, max_frames=400),
               device="cuda:0")
eng.init_variables(seed=0)
q = NativeRandomQueue(root, spklist, num_parallel=8, max_qsize=8, num_speakers=64, num_segments=2, min_len=200, max_len=400, seed=5)
q.start()
it = q.device_batches("cuda:0")
def mem():
    free, total = torch.cuda.mem_get_info()
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, (total - free) / 2**20
for i in range(50):
    x, y = next(it); eng.train_step(x, y % N, 0.01, i)
torch.cuda.synchronize()
rss0, dev0 = mem()
t0 = time.perf_counter(); chunks = 0
for i in range(steps):
    x, y = next(it)
    out = eng.train_step(x, y % N, 0.01, 50 + i, fetch_losses=(i % 500 == 0))
    if out is not None:
        print("step %5d  raw loss %.4f" % (i, out[0])); assert np.isfinite(out[0])
    chunks += x.shape[0]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
rss1, dev1 = mem()
raw, _ = eng.losses()
print("%d steps in %.1f s = %.0f chunks/s; final loss %.4f; host max RSS %.0f -> %.0f MiB; device memory in use %.0f -> %.0f MiB"
      % (steps, dt, chunks / dt, raw, rss0, rss1, dev0, dev1))
assert np.isfinite(raw) and dev1 - dev0 < 64 and rss1 - rss0 < 256
q.stop()
