"""Cost of the staged (multi-GPU) backward on one GPU: plain step vs four stages without a collective vs four stages with a
one-rank RCCL all-reduce per slice on the communication stream (not a test)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tf_kaldi_speaker_amd import engine as E
from tf_kaldi_speaker_amd.parallel import GradAllReduce
B, T, D, N = 128, 200, 30, 7351
cfg = E.make_config(D, N, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, max_batch=B, max_frames=T)
eng = E.Engine(cfg, device="cuda:0"); eng.init_variables(seed=0)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(B, T, D).astype(np.float32)).cuda(); y = torch.from_numpy(rs.randint(0, N, B).astype(np.int32)).cuda()
def run(ar, n=40):
    for i in range(5): eng.train_step(x, y, 0.01, i, allreduce=ar)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): eng.train_step(x, y, 0.01, i, allreduce=ar)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    print("plain  %.4f ms" % run(None))
    print("staged %.4f ms (4 stages, no collective)" % run(GradAllReduce(None, 1)))
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for rep in range(2):
    print("staged + one-rank RCCL all-reduce on the comm stream %.4f ms" % run(GradAllReduce(dist, 1, always=True)))
dist.destroy_process_group()
