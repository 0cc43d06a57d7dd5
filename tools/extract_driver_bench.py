#!/usr/bin/env python3
"""nnet/lib/extract.py itself, end to end: a 'CM '-compressed ark of VoxCeleb-like utterances (400..2000 frames, 30-dim) in, a float-vector
ark of embeddings out - process start, checkpoint load, ark reading, forward, writing.  Prints the wall time of the whole process and the
rate the driver itself logs at its end: utterances/s and frames/s of the reading + forward + writing loop after its first window (the
first window carries the engine's creation and warm-up).  A difference of two wall times is dominated by the spread of the start-up."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from tf_kaldi_speaker_amd.dataset import kaldi_io
from tf_kaldi_speaker_amd import engine as E

PKG = os.path.join(ROOT, "tf_kaldi_speaker_amd")


def make_model(model):
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


def make_ark(path, n, seed):
    rs = np.random.RandomState(seed)
    frames = 0
    with open(path, "wb") as f:
        for i in range(n):
            t = int(rs.randint(400, 2001))
            f.write(("utt%05d " % i).encode())
            kaldi_io.write_compressed_mat(f, rs.randn(t, 30).astype(np.float32))
            frames += t
    return frames


def main():
    tmp = tempfile.mkdtemp(prefix="xv_extract_")
    model = os.path.join(tmp, "exp")
    make_model(model)
    env = dict(os.environ, TF_KALDI_ROOT=PKG, PYTHONPATH=PKG)
    for n in (5000, 5000, 5000):         # the same archive three times: run-to-run spread on this box
        ark = os.path.join(tmp, "in%d.ark" % n)
        if not os.path.isfile(ark):
            frames = make_ark(ark, n, n)
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(PKG, "nnet", "lib", "extract.py"), "--node", "tdnn6_dense", model, "ark:" + ark,
                            "ark:" + os.path.join(tmp, "out%d.ark" % n)], env=env, cwd=PKG, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        wall = time.perf_counter() - t0
        last = [ln for ln in r.stderr.splitlines() if "Extracted" in ln]
        print("%4d utterances (%.2f M frames): %.2f s wall incl. process start-up, imports, checkpoint load; driver's own clock: %s"
              % (n, frames / 1e6, wall, last[-1].split("[INFO] ", 1)[-1] if last else "?"))


if __name__ == "__main__":
    main()
