"""Which host threads burn CPU in which phase of a run? (per-thread utime+stime from /proc/self/task)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_kaldi_speaker_amd import engine as E
B, T, D, N = 128, 200, 30, 7351
HZ = os.sysconf("SC_CLK_TCK")
def threads():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read().rsplit(")", 1)[1].split()
            out[int(t)] = (int(f[11]) + int(f[12])) / HZ
        except OSError:
            pass
    return out
def phase(name, fn):
    a, t0 = threads(), time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    b = threads()
    busy = sorted(((b[k] - a.get(k, 0.0)) / dt, k) for k in b if b[k] - a.get(k, 0.0) > 0.02 * dt)
    print("%-44s wall %.2f s  busy threads: %s" % (name, dt, ", ".join("%s%d=%.2f" % ("main:" if k == os.getpid() else "", k, v) for v, k in reversed(busy))), flush=True)
cfg = E.make_config(D, N, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, max_batch=B, max_frames=T)
eng = E.Engine(cfg, device="cuda:0"); eng.init_variables(seed=0)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(B, T, D).astype(np.float32)).cuda(); y = torch.from_numpy(rs.randint(0, N, B).astype(np.int32)).cuda()
for i in range(10): eng.train_step(x, y, 0.01, i)
torch.cuda.synchronize()
def free_run(n):
    for i in range(n): eng.train_step(x, y, 0.01, i)
    torch.cuda.synchronize()
def sync_each(n):
    for i in range(n):
        eng.train_step(x, y, 0.01, i); torch.cuda.synchronize()
def enqueue_then_wait():
    for i in range(40): eng.train_step(x, y, 0.01, i)
    torch.cuda.synchronize()
phase("idle sleep 2 s", lambda: time.sleep(2))
phase("400 steps, free run-ahead", lambda: free_run(400))
phase("400 steps, synchronize after each", lambda: sync_each(400))
phase("40 steps enqueued, then wait", enqueue_then_wait)
ev = [torch.cuda.Event(blocking=True) for _ in range(4)]
def bounded(n, depth=3):
    # run-ahead bounded to `depth` steps by a BLOCKING event recorded on torch's current stream after a cross-stream dependency?  The engine
    # runs on its own streams: use its fence + a host sleep-wait instead (poll with short sleeps)
    for i in range(n):
        eng.train_step(x, y, 0.01, i)
        if i % depth == depth - 1:
            torch.cuda.synchronize()
phase("400 steps, synchronize every 3rd", lambda: bounded(400))
