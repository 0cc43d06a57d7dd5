#!/usr/bin/env python3
"""Extraction-path throughput: utterances/s and frames/s of Trainer.predict's inner sequence
(host features -> forward in inference mode -> embedding node back to the host), one utterance at a time as
nnet/lib/extract.py does, for VoxCeleb-like lengths (uniform 400..2000 frames, 30-dim) and for fixed lengths."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from tf_kaldi_speaker_amd import engine as E


def main():
    n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    cfg = E.make_config(30, 0, max_batch=1, max_frames=10000)
    eng = E.Engine(cfg, device="cuda:0")
    eng.init_variables(seed=0)
    rs = np.random.RandomState(0)
    node = "tdnn6_dense"

    def one(x):
        eng.forward(x, False)
        return eng.endpoint(node).cpu().numpy()

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
    batched(rs)


def batched(rs, n_utts=2048):
    """The same mix through Trainer.predict_batch (length-sorted padded batches, xv_engine_forward_lengths), host float32 matrices and
    'CM ' matrices decoded on the GPU, in windows of 512 utterances as nnet/lib/extract.py feeds it."""
    import io
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tf_kaldi_speaker_amd"))
    from extract_driver_bench import make_model
    from model.trainer import Trainer
    from misc.utils import Params
    from dataset import kaldi_io
    tmp = tempfile.mkdtemp(prefix="xv_extract_b_")
    model = os.path.join(tmp, "exp")
    make_model(model)
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


if __name__ == "__main__":
    main()
