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


TWO_RANK_WORKER = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from tf_kaldi_speaker_amd import engine as E
from tf_kaldi_speaker_amd.parallel import GradAllReduce, broadcast_variables
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)                       # both ranks share the one GPU of the box; RCCL refuses that, gloo stages through the host
dist.init_process_group("gloo", rank=rank, world_size=world)
B, T, N, D, lr, steps = 8, 60, 29, 30, 0.05, 2
kw = dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, max_batch=B, max_frames=T)
def batch(r, step):
    rs = np.random.RandomState(100 * r + step)
    return rs.randn(B, T, D).astype(np.float32), rs.randint(0, N, B).astype(np.int32)
eng = E.Engine(E.make_config(D, N, **kw), device="cuda:0")
eng.init_variables(seed=rank)                  # deliberately different: the broadcast must make the replicas equal
broadcast_variables(dist, eng.variables, 0)
v0 = eng.variables.clone()
ar = GradAllReduce(dist, world)
for step in range(steps):
    x, y = batch(rank, step)
    eng.train_step(x, y, lr, step, allreduce=ar)
torch.cuda.synchronize()
mine = eng.variables[:eng.n_train].cpu().numpy().copy()
gathered = [torch.zeros(eng.n_train) for _ in range(world)]
dist.all_gather(gathered, torch.from_numpy(mine))
res = {"rank": rank, "replicas_identical": bool(all(np.array_equal(g.numpy(), mine) for g in gathered))}
if rank == 0:
    # reference on one engine: per step, the gradients of both ranks' batches from the same weights, summed, update with 1/world.
    # BN moving statistics are per rank by design, so they follow rank 0's batch only.
    ref = E.Engine(E.make_config(D, N, **kw), device="cuda:0")
    ref.variables.copy_(v0)
    ref.lib.xv_engine_invalidate_weights(ref.h)
    for step in range(steps):
        keep = ref.variables.clone()
        total = None
        for r in (1, 0):                       # rank 0's batch last: its BN moving-average update is the one that stays
            ref.variables.copy_(keep); ref.lib.xv_engine_invalidate_weights(ref.h)
            x, y = batch(r, step)
            ref.forward(x, True); ref.loss(y, step, True); ref.backward(-1)
            g = ref.grads.clone()
            total = g if total is None else total + g
        ref.grads.copy_(total)
        ref.apply(lr, 1.0 / world)
    torch.cuda.synchronize()
    want = ref.variables[:ref.n_train].cpu().numpy()
    res["max_diff_vs_single_engine"] = float(np.abs(want - mine).max())
    res["moved"] = float(np.abs(mine - v0[:eng.n_train].cpu().numpy()).max())
    ref.close()
eng.close()
dist.barrier(); dist.destroy_process_group()
json.dump(res, open(os.path.join(sys.argv[1], "rank%%d.json" %% rank), "w"))      # (stdout of the two ranks may interleave)
"""


def test_two_ranks_sharing_the_gpu_average_gradients_like_one_engine(tmp_path):
    """World size 2 for real: two processes, each with its own engine and its own minibatch, on the one GPU of the box
    (gloo carries the device tensors; the 8-GPU RCCL runs are the driver's).  After two steps the replicas are bit-identical
    and equal the single-engine computation that sums both batches' gradients and applies them with 1/world."""
    script = tmp_path / "worker2.py"
    script.write_text(TWO_RANK_WORKER % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), str(script), str(tmp_path)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert len(lines) == 2 and all(l["replicas_identical"] for l in lines), lines
    r0 = [l for l in lines if l["rank"] == 0][0]
    assert r0["moved"] > 1e-4 and r0["max_diff_vs_single_engine"] == 0.0, r0


def test_bench_two_ranks_sharing_the_gpu():
    """bench.py's N > 1 path end to end (barriers, max-over-ranks timing, staged all-reduce, rank-0 JSON line) with two ranks
    on the one GPU of the box over gloo (XV_SHARE_GPU=1); the value itself is meaningless here."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", XV_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + os.getpid() % 90), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                       # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 100 and line["scaling"] == "weak" and line["config"]["parallelism"] == "dp2"
    assert "cpu_baseline" not in line and np.isfinite(line["loss"])
    # the per-rank communication report the scaling runs are read with: backend / world size as the process group reports them, one
    # all-reduce time per gradient slice (events on the communication stream), how much of it the backward pass did not cover
    comm = line["comm"]
    assert [c["rank"] for c in sorted(comm, key=lambda c: c["rank"])] == [0, 1]
    for c in comm:
        assert c["world_size"] == 2 and c["backend"] == "gloo", c
        rep = c["report"]
        assert "error" not in rep and rep["steps"] == 3 and len(rep["slices"]) == 4, rep
        assert abs(sum(sl["mbytes"] for sl in rep["slices"]) - 39.3) < 0.3               # the whole 9.83 M-float gradient buffer
        assert rep["allreduce_ms_per_step"] > 0 and 0.0 <= rep["overlap_frac"] <= 1.0


def test_bench_gpus8_launches_itself_with_eight_ranks_sharing_the_gpu():
    """`python bench.py --gpus 8` exactly as the driver calls it for N = 1 (no torch.distributed.run around it, no WORLD_SIZE): bench.py
    starts the 8 ranks itself as child processes.  On this 1-GPU box the ranks share the device over gloo (XV_SHARE_GPU=1): eight
    engines, eight seeded feeds, the staged all-reduce with world 8 - so the first real 8-GPU run is not the first 8-rank run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", XV_SHARE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["config"]["parallelism"] == "dp8" and line["scaling"] == "weak" and np.isfinite(line["loss"])
    assert line["value"] > 100 and abs(line["value"] - 8 * 128 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-2 * line["value"]
    comm = sorted(line["comm"], key=lambda c: c["rank"])
    assert [c["rank"] for c in comm] == list(range(8))
    assert len({c["data_seed"] for c in comm}) == 8                      # every rank draws its own minibatches
    for c in comm:
        assert c["world_size"] == 8 and c["backend"] == "gloo", c
        rep = c["report"]
        assert "error" not in rep and rep["steps"] == 3 and len(rep["slices"]) == 4, rep
        # the invariants a real 8-GPU run's report is read with: the whole 9.83 M-float gradient buffer in four slices, the
        # slice with the speaker matrix (15.05 MB of it) first and largest, the slice of the first two frame layers last; what is still
        # running when the compute stream has nothing left can only be (part of) that last slice's collective
        mb = [sl["mbytes"] for sl in rep["slices"]]
        assert abs(sum(mb) - 39.3) < 0.3 and mb[0] == max(mb) and mb[0] >= 15.05 and 5.0 < mb[-1] < 6.0, mb
        # (on real xGMI `exposed_ms` <= the last slice's time is the figure to read; eight ranks time-slicing ONE device over gloo can
        # start every slice late, so here only the report's own consistency is asserted)
        assert rep["exposed_ms_per_step"] is not None and rep["exposed_ms_per_step"] >= 0.0 and 0.0 <= rep["overlap_frac"] <= 1.0
        assert abs(rep["allreduce_ms_per_step"] - sum(sl["allreduce_ms"] for sl in rep["slices"])) < 1e-2 * max(rep["allreduce_ms_per_step"], 1.0)
        assert np.isfinite(c["loss"])
    # every rank trains on its own minibatches (different losses), yet the replicas stay bit-identical: same initial variables, the same
    # summed gradients, the same update
    assert len({round(c["loss"], 6) for c in comm}) > 1, [c["loss"] for c in comm]
    assert len({c["trainable_checksum"] for c in comm}) == 1, [c["trainable_checksum"] for c in comm]
