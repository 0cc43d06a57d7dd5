#!/usr/bin/env python3
"""Host-path benchmarks around the engine (loader, end-to-end feed, Trainer.train, extraction) - diagnostics, not tests.  One file, one
sub-command per question (they used to be eight scripts); every one builds its own synthetic Kaldi directory / archive under $TMPDIR.

  python tools/bench_host.py loader [workers=4] [batches=40]    native C++ loader threads vs batches planned in Python + native codec on a thread pool
  python tools/bench_host.py loader_scale [--procs 8] [--threads 8] [--batches 60] [--need 25800]
                                                                N loader processes at once (one per rank of an N-GPU job): aggregate chunks/s as JSON
  python tools/bench_host.py e2e [steps=60] [threads=8]         native loader -> pinned -> async H2D -> training steps, against resident batches
                                                                (XV_LOADER=gpu_decode: 'CM ' bytes over PCIe, decoded on the GPU)
  python tools/bench_host.py e2e_ab [--steps 100] [--rounds 3]  bench.py's e2e leg, resident vs host -> device copy inside the timed region, alternated
  python tools/bench_host.py trainer [steps=400]                Trainer.train itself: one epoch incl. checkpoint load + save, shipped batch shape
  python tools/bench_host.py soak [steps=2000]                  loader-fed steps: sustained rate, finite loss, host / device memory before and after
  python tools/bench_host.py extract [utterances=300]           extraction: engine one utterance at a time, then Trainer.predict_batch (host fp32 / 'CM ' + GPU decode)
  python tools/bench_host.py extract_driver                     nnet/lib/extract.py end to end on a 5 000-utterance 'CM ' ark, three times
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

PKG = os.path.join(ROOT, "tf_kaldi_speaker_amd")
N_SPK = 7351


def synthetic_dir(prefix, num_spk=100, utts_per_spk=8):
    from tests.kaldi_fixture import make_data_dir
    root = tempfile.mkdtemp(prefix=prefix)
    root, spklist, _ = make_data_dir(root, num_spk=num_spk, utts_per_spk=utts_per_spk, dim=30, min_frames=500, max_frames=1200, seed=0)
    return root, spklist


def bench_engine():
    from tf_kaldi_speaker_amd import engine as E
    eng = E.Engine(E.make_config(30, N_SPK, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, max_batch=128, max_frames=400), device="cuda:0")
    eng.init_variables(seed=0)
    return eng


def cmd_loader(argv):
    from tf_kaldi_speaker_amd.dataset.native_loader import NativeRandomQueue
    from tf_kaldi_speaker_amd.dataset.data_loader import PlannedRandomQueue
    workers = int(argv[0]) if argv else 4
    nbatch = int(argv[1]) if len(argv) > 1 else 40
    root, spklist = synthetic_dir("xv_loader_bench_")
    kw = dict(num_parallel=workers, max_qsize=10, num_speakers=64, num_segments=2, min_len=200, max_len=400, shuffle=True)
    for name, cls in (("native", NativeRandomQueue), ("planned", PlannedRandomQueue)):
        q = cls(root, spklist, **kw)
        q.start()
        q.fetch()
        t0 = time.time()
        chunks = 0
        for _ in range(nbatch):
            f, l = q.fetch()
            chunks += f.shape[0]
        dt = time.time() - t0
        q.stop()
        print("%-7s %d workers: %8.0f chunks/s  (%d batches of 128 chunks x T~U[200,400] x 30 in %.2f s)" % (name, workers, chunks / dt, nbatch, dt))


def _scale_worker(rank, root, spklist, threads, batches, chunks, start_evt, q):
    pin = int(os.environ.get("XV_LOADER_PIN", "0"))          # > 0: rank r keeps to CPUs [r * pin, (r + 1) * pin) (NUMA / CCD locality experiment)
    if pin > 0:
        os.sched_setaffinity(0, set(range(rank * pin, (rank + 1) * pin)))
    from tf_kaldi_speaker_amd.dataset.native_loader import NativeRandomQueue
    spk, seg = (chunks // 2, 2) if chunks % 2 == 0 else (chunks, 1)
    packed = os.environ.get("XV_LOADER", "native") == "gpu_decode"      # the threads only gather 'CM ' bytes (decode happens on the GPU)
    ld = NativeRandomQueue(root, spklist, num_parallel=threads, max_qsize=8, num_speakers=spk, num_segments=seg, min_len=200, max_len=400,
                           seed=100 + rank, packed=packed)
    ld.start()
    lab = np.empty(chunks, np.int32)
    if packed:
        from tf_kaldi_speaker_amd.dataset.native_loader import packed_chunk_bytes
        feat = np.empty(chunks * packed_chunk_bytes(ld.dim, 400), np.uint8)
        ld.fetch_into = ld.fetch_packed_into
    else:
        feat = np.empty(chunks * 400 * ld.dim, np.float32)
    ld.fetch_into(feat, lab)                      # warm: threads running, files open
    q.put(("ready", rank))
    start_evt.wait()
    t0 = time.perf_counter()
    n = 0
    for _ in range(batches):
        ld.fetch_into(feat, lab)
        n += chunks
    dt = time.perf_counter() - t0
    ld.stop()
    q.put(("done", rank, n, dt))


def loader_scale_run(procs, threads, batches, chunks=128, root=None, spklist=None):
    """(also called by tests/test_native_loader.py on its own small directory)"""
    import multiprocessing as mp
    if root is None:
        root, spklist = synthetic_dir("xv_loader_scale_", num_spk=max(chunks, 100), utts_per_spk=6)
    ctx = mp.get_context("spawn")
    q, start_evt = ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=_scale_worker, args=(r, root, spklist, threads, batches, chunks, start_evt, q)) for r in range(procs)]
    for p in ps:
        p.start()
    for _ in range(procs):
        assert q.get(timeout=300)[0] == "ready"
    t0 = time.perf_counter()
    start_evt.set()
    res = [q.get(timeout=600) for _ in range(procs)]
    wall = time.perf_counter() - t0
    for p in ps:
        p.join(30)
    per = sorted((r[1], r[2] / r[3]) for r in res)
    total = sum(r[2] for r in res)
    return {"procs": procs, "threads_per_proc": threads, "batches_per_proc": batches, "chunks_per_batch": chunks,
            "per_proc_chunks_per_s": [round(v, 1) for _, v in per], "aggregate_chunks_per_s": round(total / wall, 1), "host_cpus": os.cpu_count()}


def cmd_loader_scale(argv):
    """1 -> 8 GPU scaling of this workload is decided by the loader (the gradient all-reduce is < 0.5 ms of a 5 ms step), so
    the aggregate has to exceed N x the per-GPU step rate."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--batches", type=int, default=60)
    ap.add_argument("--need", type=float, default=25800.0, help="per-GPU step rate in chunks/s the loaders have to out-run (bench.py `value`)")
    a = ap.parse_args(argv)
    one = loader_scale_run(1, a.threads, a.batches)
    many = loader_scale_run(a.procs, a.threads, a.batches)
    many["single_proc_chunks_per_s"] = one["aggregate_chunks_per_s"]
    many["needed_chunks_per_s"] = a.procs * a.need
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import cgroup_cpus
    many["cgroup_cpus"] = cgroup_cpus()      # the aggregate is a figure for this many cores, whatever host_cpus says
    if many["cgroup_cpus"]:
        many["chunks_per_s_per_cpu"] = round(many["aggregate_chunks_per_s"] / min(many["cgroup_cpus"], a.procs * (a.threads + 1)), 1)
    many["headroom"] = round(many["aggregate_chunks_per_s"] / many["needed_chunks_per_s"], 2)
    print(json.dumps(many))


def _native_feed(threads, seed=5):
    from tf_kaldi_speaker_amd.dataset.native_loader import NativeRandomQueue
    root, spklist = synthetic_dir("xv_e2e_")
    q = NativeRandomQueue(root, spklist, num_parallel=threads, max_qsize=8, num_speakers=64, num_segments=2, min_len=200, max_len=400, seed=seed,
                          packed=os.environ.get("XV_LOADER", "native") == "gpu_decode")
    q.start()
    return q, q.device_batches("cuda:0")


def cmd_e2e(argv):
    import torch
    steps = int(argv[0]) if argv else 60
    threads = int(argv[1]) if len(argv) > 1 else 8
    eng = bench_engine()
    q, it = _native_feed(threads)

    def run(batches, n):
        chunks = frames = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            x, y = next(batches)
            eng.train_step(x, y % N_SPK, 0.01, i)
            chunks += x.shape[0]
            frames += x.shape[0] * x.shape[1]
        torch.cuda.synchronize()
        return chunks / (time.perf_counter() - t0), frames / chunks
    run(it, 10)
    rate, mean_t = run(it, steps)
    print("loader -> engine : %8.0f chunks/s (mean T %.0f, %d decoder threads)" % (rate, mean_t, threads))
    resident = [next(it) for _ in range(16)]

    def cyc():
        i = 0
        while True:
            yield resident[i % 16]
            i += 1
    rate2, mean_t2 = run(cyc(), steps)
    print("resident batches : %8.0f chunks/s (mean T %.0f)" % (rate2, mean_t2))
    q.stop()
    eng.close()


def cmd_e2e_ab(argv):
    """(the e2e leg of a default bench run comes last, on a chip two timed modes have warmed: here the two legs alternate in one process)"""
    import argparse
    import torch
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args(argv)
    a.extended, a.attention = False, False
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    for r in range(a.rounds):
        for h2d in (False, True):
            res = bench.run_mode(a.precision, a, dev, 0, 1, None, 128, 200, 200, h2d=h2d, light=True)
            print("round %d  %-8s %.4f ms/step" % (r, "h2d" if h2d else "resident", res["elapsed"] / a.steps * 1e3), flush=True)


def cmd_trainer(argv):
    from tests.kaldi_fixture import make_data_dir
    from tf_kaldi_speaker_amd.misc.utils import Params
    from tf_kaldi_speaker_amd.model.trainer import Trainer
    steps = int(argv[0]) if argv else 400
    root = tempfile.mkdtemp(prefix="xv_trainer_bench_")
    data, spklist, _ = make_data_dir(os.path.join(root, "train"), num_spk=100, utts_per_spk=8, dim=30, min_frames=500, max_frames=1200, seed=0)
    cfg = {"seed": 0, "network_type": "tdnn", "last_layer_no_bn": False, "last_layer_linear": True, "feature_norm": False,
           "loss_func": "additive_margin_softmax", "amsoftmax_m": 0.20, "amsoftmax_lambda_min": 0, "amsoftmax_lambda_base": 1000,
           "amsoftmax_lambda_gamma": 0.0001, "amsoftmax_lambda_power": 5, "batch_type": "softmax", "pooling_type": "statistics_pooling",
           "embedding_node": "tdnn6_dense", "learning_rate": 0.01, "use_nesterov": False, "clip_gradient": False, "clip_gradient_norm": 3,
           "weight_l2_regularizer": 1e-2, "batchnorm_momentum": 0.99, "num_epochs": 1, "num_steps_per_epoch": steps, "reduce_lr_epochs": 4,
           "show_training_progress": 100, "keep_checkpoint_max": 5, "save_summary_steps": 10000, "save_checkpoints_steps": 30000,
           "valid_max_iterations": 1000, "num_parallel_datasets": 8, "max_queue_size": 8, "num_speakers_per_batch": 64,
           "num_segments_per_speaker": 2, "min_segment_len": 200, "max_segment_len": 400}
    cfg_path = os.path.join(root, "config.json")
    json.dump(cfg, open(cfg_path, "w"))
    model = os.path.join(root, "exp")
    os.makedirs(os.path.join(model, "nnet"))
    tr = Trainer(Params(cfg_path), model)
    tr.build("train", dim=30, loss_type=cfg["loss_func"], num_speakers=N_SPK)
    tr.train(data, spklist, 0.01)          # epoch 1: includes engine warm-up
    t0 = time.perf_counter()
    tr.train(data, spklist, 0.01)          # epoch 2: load checkpoint, `steps` steps, save
    dt = time.perf_counter() - t0
    print("Trainer.train: %d steps x 128 chunks in %.2f s = %.0f chunks/s (checkpoint load + save of 39 MB variables included)" % (steps, dt, steps * 128 / dt))
    tr.close()


def cmd_soak(argv):
    import resource
    import torch
    steps = int(argv[0]) if argv else 2000
    eng = bench_engine()
    q, it = _native_feed(8)

    def mem():
        free, total = torch.cuda.mem_get_info()
        return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, (total - free) / 2 ** 20
    for i in range(50):
        x, y = next(it)
        eng.train_step(x, y % N_SPK, 0.01, i)
    torch.cuda.synchronize()
    rss0, dev0 = mem()
    t0 = time.perf_counter()
    chunks = 0
    for i in range(steps):
        x, y = next(it)
        out = eng.train_step(x, y % N_SPK, 0.01, 50 + i, fetch_losses=(i % 500 == 0))
        if out is not None:
            print("step %5d  raw loss %.4f" % (i, out[0]))
            assert np.isfinite(out[0])
        chunks += x.shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rss1, dev1 = mem()
    raw, _ = eng.losses()
    print("%d steps in %.1f s = %.0f chunks/s; final loss %.4f; host max RSS %.0f -> %.0f MiB; device memory in use %.0f -> %.0f MiB"
          % (steps, dt, chunks / dt, raw, rss0, rss1, dev0, dev1))
    assert np.isfinite(raw) and dev1 - dev0 < 64 and rss1 - rss0 < 256
    q.stop()


def make_extract_model(model):
    """A model directory nnet/lib/extract.py accepts: config, feature_dim, one .npz checkpoint, the checkpoint index."""
    from tf_kaldi_speaker_amd import engine as E
    nnet = os.path.join(model, "nnet")
    os.makedirs(nnet)
    cfg = {"network_type": "tdnn", "loss_func": "softmax", "pooling_type": "statistics_pooling", "embedding_node": "tdnn6_dense", "seed": 0,
           "last_layer_no_bn": False, "last_layer_linear": False, "weight_l2_regularizer": 1e-2, "batchnorm_momentum": 0.99,
           "optimizer": "sgd", "num_nodes_pooling_layer": 1500, "num_nodes_last_layer": 512, "feature_norm": False}
    json.dump(cfg, open(os.path.join(nnet, "config.json"), "w"))
    open(os.path.join(nnet, "feature_dim"), "w").write("30\n")
    eng = E.Engine(E.make_config(30, 10, max_batch=1, max_frames=100), device="cuda:0")
    eng.init_variables(seed=0)
    np.savez(os.path.join(nnet, "model-1.npz"), **eng.get_variables())
    eng.close()
    open(os.path.join(nnet, "checkpoint"), "w").write('model_checkpoint_path: "model-1"\nall_model_checkpoint_paths: "model-1"\n')


def cmd_extract(argv):
    """utterances/s and frames/s of Trainer.predict's inner sequence one utterance at a time as nnet/lib/extract.py's
    reference form does (uniform 400..2000 frames, 30-dim, and fixed lengths), then the same mix through Trainer.predict_batch (length-sorted
    padded batches, xv_engine_forward_lengths) in windows of 512 utterances as the driver feeds it."""
    import io
    import torch
    from tf_kaldi_speaker_amd import engine as E
    n_utts = int(argv[0]) if argv else 300
    eng = E.Engine(E.make_config(30, 0, max_batch=1, max_frames=10000), device="cuda:0")
    eng.init_variables(seed=0)
    rs = np.random.RandomState(0)

    def one(x):
        eng.forward(x, False)
        return eng.endpoint("tdnn6_dense").cpu().numpy()
    for t in (300, 1000, 3000, 10000):
        x = rs.randn(1, t, 30).astype(np.float32)
        for _ in range(3):
            one(x)
        t0 = time.perf_counter()
        for _ in range(20):
            one(x)
        dt = (time.perf_counter() - t0) / 20
        print("T=%5d  %.3f ms/utterance  %.2f M frames/s" % (t, dt * 1e3, t / dt / 1e6))
    lens = rs.randint(400, 2001, n_utts)
    utts = [rs.randn(1, int(t), 30).astype(np.float32) for t in lens]
    for x in utts[:5]:
        one(x)
    t0 = time.perf_counter()
    for x in utts:
        one(x)
    dt = time.perf_counter() - t0
    print("mixed 400..2000 frames: %d utterances in %.3f s = %.0f utterances/s, %.2f M frames/s" % (n_utts, dt, n_utts / dt, lens.sum() / dt / 1e6))
    eng.close()
    # batched
    sys.path.insert(0, PKG)
    from model.trainer import Trainer
    from misc.utils import Params
    from dataset import kaldi_io
    n_utts = 2048
    model = os.path.join(tempfile.mkdtemp(prefix="xv_extract_b_"), "exp")
    make_extract_model(model)
    tr = Trainer(Params(os.path.join(model, "nnet", "config.json")), model, single_cpu=True)
    tr.build("predict", dim=30)
    lens = rs.randint(400, 2001, n_utts)
    utts = [rs.randn(int(t), 30).astype(np.float32) for t in lens]
    buf = io.BytesIO()
    for i, u in enumerate(utts):
        kaldi_io.write_compressed_mat(buf, u, key="u%d" % i)
    buf.seek(0)
    packed = [m for _, m in kaldi_io.read_mat_ark_packed(buf)]
    for name, items in (("host fp32 matrices", utts), ("'CM ' matrices, GPU decode", packed)):
        tr.predict_batch(items[:512])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for w in range(0, n_utts, 512):
            tr.predict_batch(items[w:w + 512])
        dt = time.perf_counter() - t0
        print("batched, %s: %d utterances in %.3f s = %.0f utterances/s, %.2f M frames/s" % (name, n_utts, dt, n_utts / dt, lens.sum() / dt / 1e6))
    tr.close()


def cmd_extract_driver(argv):
    """Process start, checkpoint load, ark reading, forward, writing.  Prints the wall time of the whole process and the rate the driver itself logs
    at its end (reading + forward + writing after its first window, which carries the engine's creation and warm-up).  A difference of two wall
    times is dominated by the spread of the start-up."""
    import subprocess
    from tf_kaldi_speaker_amd.dataset import kaldi_io
    tmp = tempfile.mkdtemp(prefix="xv_extract_")
    model = os.path.join(tmp, "exp")
    make_extract_model(model)
    env = dict(os.environ, TF_KALDI_ROOT=PKG, PYTHONPATH=PKG)
    frames = 0
    for n in (5000, 5000, 5000):         # the same archive three times: run-to-run spread on this box
        ark = os.path.join(tmp, "in%d.ark" % n)
        if not os.path.isfile(ark):
            rs = np.random.RandomState(n)
            with open(ark, "wb") as f:
                for i in range(n):
                    t = int(rs.randint(400, 2001))
                    f.write(("utt%05d " % i).encode())
                    kaldi_io.write_compressed_mat(f, rs.randn(t, 30).astype(np.float32))
                    frames += t
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(PKG, "nnet", "lib", "extract.py"), "--node", "tdnn6_dense", model, "ark:" + ark,
                            "ark:" + os.path.join(tmp, "out%d.ark" % n)], env=env, cwd=PKG, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        wall = time.perf_counter() - t0
        last = [ln for ln in r.stderr.splitlines() if "Extracted" in ln]
        print("%4d utterances (%.2f M frames): %.2f s wall incl. process start-up, imports, checkpoint load; driver's own clock: %s"
              % (n, frames / 1e6, wall, last[-1].split("[INFO] ", 1)[-1] if last else "?"))


COMMANDS = {"loader": cmd_loader, "loader_scale": cmd_loader_scale, "e2e": cmd_e2e, "e2e_ab": cmd_e2e_ab, "trainer": cmd_trainer, "soak": cmd_soak,
            "extract": cmd_extract, "extract_driver": cmd_extract_driver}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in COMMANDS:
        sys.exit(__doc__)
    COMMANDS[sys.argv[1]](sys.argv[2:])
