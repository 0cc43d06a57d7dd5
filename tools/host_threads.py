"""What does one rank cost the HOST?  400 free-running S1 steps of the engine: ms/step and which threads of the process were busy (utime + stime
per thread from /proc/self/task), under whatever environment it is started in.  XV_PROBE_FLAGS=<n>: hipSetDeviceFlags(n) first (4 = blocking
sync, 2 = yield); XV_PROBE_MODE=each: synchronise after every step.  -> profiles/r06_host_cpu.txt
    python tools/host_threads.py <label>"""
import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
fl = os.environ.get("XV_PROBE_FLAGS")
rc = None
if fl is not None:
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipSetDeviceFlags(ctypes.c_uint(int(fl)))
from tf_kaldi_speaker_amd import engine as E
B, T, D, N = 128, 200, 30, 7351
cfg = E.make_config(D, N, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, max_batch=B, max_frames=T)
eng = E.Engine(cfg, device="cuda:0"); eng.init_variables(seed=0)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(B, T, D).astype(np.float32)).cuda(); y = torch.from_numpy(rs.randint(0, N, B).astype(np.int32)).cuda()
for i in range(30): eng.train_step(x, y, 0.01, i)
torch.cuda.synchronize()
mode = os.environ.get("XV_PROBE_MODE", "free")
a, t0 = threads(), time.perf_counter()
n = 400
for i in range(n):
    eng.train_step(x, y, 0.01, i)
    if mode == "each": torch.cuda.synchronize()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
b = threads()
busy = sorted(((b[k] - a.get(k, 0.0)) / dt, k) for k in b if b[k] - a.get(k, 0.0) > 0.02 * dt)
print("%-60s %.3f ms/step  busy: %s%s" % (sys.argv[1] if len(sys.argv) > 1 else "", dt / n * 1e3, ", ".join("%s%.2f" % ("main=" if k == os.getpid() else "t=", v) for v, k in reversed(busy)),
      "" if rc is None else "  (hipSetDeviceFlags rc %d)" % rc), flush=True)
