"""The data-parallel wiring on real RCCL: a one-rank `nccl` process group on the GPU box (multi-GPU boxes are the driver's),
collectives forced on.  Checks that the staged backward + asynchronous all-reduce of the finished gradient slices + update
gives exactly the variables of the plain step (sum over one rank, scale 1) - i.e. stream ordering between the engine's
streams and RCCL's is right - and that bench.py's launch path under torch.distributed.run works."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from tf_kaldi_speaker_amd import engine as E
from tf_kaldi_speaker_amd.parallel import GradAllReduce, average_bn_statistics, broadcast_variables
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
B, T, N = 16, 80, 37
def run(use_dist):
    eng = E.Engine(E.make_config(30, N, loss_func="additive_angular_margin_softmax", margin_m=0.3, last_layer_linear=True,
                                 max_batch=B, max_frames=T), device="cuda:0")
    eng.init_variables(seed=3)
    if use_dist:
        broadcast_variables(dist, eng.variables, 0)
    rs = np.random.RandomState(5)
    ar = GradAllReduce(dist, 1, always=True) if use_dist else None
    for step in range(3):
        x = rs.randn(B, T, 30).astype(np.float32); y = rs.randint(0, N, B).astype(np.int32)
        eng.train_step(x, y, 0.05, step, allreduce=ar)
    if use_dist:
        average_bn_statistics(dist, eng.variables, eng.n_train, 2)     # sum + 1/2 on one rank: halves the tail
    torch.cuda.synchronize()
    v = eng.variables.cpu().numpy().copy(); nt = eng.n_train
    eng.close()
    return v, nt
a, nt = run(False)
b, _ = run(True)
dist.barrier(); dist.destroy_process_group()
print(json.dumps({"max_diff_trainable": float(np.abs(a[:nt] - b[:nt]).max()), "tail_ratio_err": float(np.abs(b[nt:] - 0.5 * a[nt:]).max()),
                  "finite": bool(np.isfinite(b).all())}))
"""


def test_staged_allreduce_on_rccl_matches_plain_step(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["finite"] and res["max_diff_trainable"] == 0.0 and res["tail_ratio_err"] < 1e-7, res


def test_bench_under_torch_distributed_run():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 1000 and line["scaling"] == "weak" and "roofline" in line
