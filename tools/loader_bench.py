"""Throughput of the two minibatch streams on a synthetic Kaldi directory (CM-compressed 30-dim features):
native C++ threads (libxvector_io.so: planning + decoding) vs batches planned in Python and decoded by the native codec on a thread pool."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.kaldi_fixture import make_data_dir
from tf_kaldi_speaker_amd.dataset.native_loader import NativeRandomQueue
from tf_kaldi_speaker_amd.dataset.data_loader import PlannedRandomQueue

if __name__ == "__main__":
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    nbatch = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    root = tempfile.mkdtemp(prefix="xv_loader_bench_")
    root, spklist, _ = make_data_dir(root, num_spk=100, utts_per_spk=8, dim=30, min_frames=500, max_frames=1200, seed=0)
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
