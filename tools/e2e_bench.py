"""End-to-end step rate: native loader (CM decode on host threads -> pinned -> async H2D) feeding the engine, vs the same
engine on resident synthetic batches.  Shape S3-like: 128 chunks (64 speakers x 2), T ~ U[200,400], 30-dim, 7351 classes."""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.kaldi_fixture import make_data_dir
from tf_kaldi_speaker_amd import engine as E
from tf_kaldi_speaker_amd.dataset.native_loader import NativeRandomQueue

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
root = tempfile.mkdtemp(prefix="xv_e2e_")
root, spklist, _ = make_data_dir(root, num_spk=100, utts_per_spk=8, dim=30, min_frames=500, max_frames=1200, seed=0)
N = 7351
cfg = E.make_config(30, N, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, max_batch=128, max_frames=400)
eng = E.Engine(cfg, device="cuda:0")
eng.init_variables(seed=0)
q = NativeRandomQueue(root, spklist, num_parallel=threads, max_qsize=8, num_speakers=64, num_segments=2, min_len=200, max_len=400, seed=5,
                      packed=os.environ.get("XV_LOADER", "native") == "gpu_decode")      # XV_LOADER=gpu_decode: 'CM ' bytes over PCIe, xv_cm_decode on the GPU
q.start()
it = q.device_batches("cuda:0")
def run(batches, n):
    chunks = frames = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        x, y = next(batches)
        eng.train_step(x, y % N, 0.01, i)
        chunks += x.shape[0]; frames += x.shape[0] * x.shape[1]
    torch.cuda.synchronize()
    return chunks / (time.perf_counter() - t0), frames / chunks
run(it, 10)
rate, mean_t = run(it, steps)
print("loader -> engine : %8.0f chunks/s (mean T %.0f, %d decoder threads)" % (rate, mean_t, threads))
resident = [next(it) for _ in range(16)]
def cyc():
    i = 0
    while True:
        yield resident[i % 16]; i += 1
rate2, mean_t2 = run(cyc(), steps)
print("resident batches : %8.0f chunks/s (mean T %.0f)" % (rate2, mean_t2))
q.stop(); eng.close()
