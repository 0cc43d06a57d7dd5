"""Can the host feed N GPUs?  N processes (one per rank, as under torch.distributed.run), each with its own native loader
(libxvector_io.so, `threads` decoder threads), pull batches of the benchmark shape concurrently from one synthetic Kaldi
directory; prints the per-process and aggregate chunks/s as JSON.  SURVEY.md section 8e: 1 -> 8 GPU scaling of this workload is
decided by the loader (the gradient all-reduce is < 0.5 ms of a 6 ms step), so the aggregate has to exceed N x the per-GPU step rate.

  python tools/loader_scale.py [--procs 8] [--threads 8] [--batches 60] [--need 21500]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, root, spklist, threads, batches, chunks, start_evt, q):
    import numpy as np
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


def run(procs, threads, batches, chunks=128, root=None, spklist=None):
    from tests.kaldi_fixture import make_data_dir
    if root is None:
        root = tempfile.mkdtemp(prefix="xv_loader_scale_")
        root, spklist, _ = make_data_dir(root, num_spk=max(chunks, 100), utts_per_spk=6, dim=30, min_frames=500, max_frames=1200, seed=0)
    ctx = mp.get_context("spawn")
    q, start_evt = ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=_worker, args=(r, root, spklist, threads, batches, chunks, start_evt, q)) for r in range(procs)]
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
            "per_proc_chunks_per_s": [round(v, 1) for _, v in per], "aggregate_chunks_per_s": round(total / wall, 1),
            "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--batches", type=int, default=60)
    ap.add_argument("--need", type=float, default=21500.0, help="per-GPU step rate in chunks/s the loaders have to out-run (bench.py `value`)")
    a = ap.parse_args()
    one = run(1, a.threads, a.batches)
    many = run(a.procs, a.threads, a.batches)
    many["single_proc_chunks_per_s"] = one["aggregate_chunks_per_s"]
    many["needed_chunks_per_s"] = a.procs * a.need
    many["headroom"] = round(many["aggregate_chunks_per_s"] / many["needed_chunks_per_s"], 2)
    print(json.dumps(many))
