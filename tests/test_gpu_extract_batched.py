"""Batched extraction on a real MI355X: many utterances of different lengths in one forward
(xv_engine_forward_lengths, Trainer.predict_batch, nnet/lib/extract.py) against the one-utterance-at-a-time path the reference
runs (egs/voxceleb/v1/nnet/lib/extract.py:64-93, model/trainer.py:708-726) and against the fp64 oracle; 'CM ' matrices decoded on
the GPU (xv_cm_decode_ragged) bit for bit like the reference codec."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import xvector_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tf_kaldi_speaker_amd")


def rel_err(got, ref):
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


def _engine(kw, max_batch, max_frames, max_rows=0, seed=0):
    from tf_kaldi_speaker_amd import engine as E
    eng = E.Engine(E.make_config(30, 0, max_batch=max_batch, max_frames=max_frames, max_rows=max_rows, **kw), device="cuda:0")
    eng.init_variables(seed=seed)
    V = {k: v.astype(np.float64) for k, v in eng.get_variables().items()}
    rs = np.random.RandomState(7)
    for k in V:      # moving statistics away from their initial (0, 1): inference mode must really use them
        if k.endswith("moving_mean"):
            V[k] = rs.randn(*V[k].shape) * 0.1
        if k.endswith("moving_variance"):
            V[k] = 0.5 + rs.rand(*V[k].shape)
    eng.set_variables({k: v.astype(np.float32) for k, v in V.items()})
    V = {k: v.astype(np.float64) for k, v in eng.get_variables().items()}
    return eng, V


@pytest.mark.parametrize("kw", [dict(), dict(pooling_type="self_attention", att_key_num_nodes=(200, 120))], ids=["statistics", "attention"])
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_forward_lengths_matches_one_at_a_time_and_the_oracle(kw, precision):
    """Padded batch with per-utterance lengths == each utterance alone (1e-6; the frame layers are valid convolutions, pooling sees
    the utterance's own frames only) == fp64 oracle (5e-5), incl. an utterance of exactly the receptive field and one that fills the batch."""
    lens = [15, 25, 137, 300, 64, 299]
    T = max(lens)
    eng, V = _engine(dict(kw, precision=precision), len(lens), T)
    cfg_o = O.Config(feat_dim=30, num_speakers=0, **{k: v for k, v in kw.items()})
    rs = np.random.RandomState(3)
    utts = [rs.randn(n, 30).astype(np.float32) for n in lens]
    x = np.full((len(lens), T, 30), 1e3, np.float32)              # the padding is NOT zero: it must not reach any utterance's embedding
    for i, u in enumerate(utts):
        x[i, :len(u)] = u
    eng.forward_lengths(x, np.asarray(lens, np.int32))
    got = {n: eng.endpoint(n).cpu().numpy() for n in ("pooling", "tdnn6_dense", "tdnn7_dense", "output")}
    for i, u in enumerate(utts):
        eng.forward(u[None], False)
        _, ep, _ = O.tdnn_forward(V, u[None].astype(np.float64), cfg_o, False)
        for name in got:
            one = eng.endpoint(name).cpu().numpy()[0]
            tol_one = 1e-6 if precision == "f32" else 2e-6      # f16x3: the operand scale follows the batch's largest value (incl. the padding)
            assert rel_err(got[name][i], one.astype(np.float64)) <= tol_one, (i, name, rel_err(got[name][i], one.astype(np.float64)))
            assert rel_err(got[name][i], ep[name][0]) <= 5e-5, (i, name, rel_err(got[name][i], ep[name][0]))
    with pytest.raises(Exception):
        eng.forward_lengths(x, np.asarray(lens[:-1], np.int32))        # one count per chunk
    eng.close()


def test_cm_decode_ragged_is_bit_identical_and_zero_pads():
    import torch
    from tf_kaldi_speaker_amd import engine as E
    from tf_kaldi_speaker_amd.dataset import kaldi_io
    import io
    rs = np.random.RandomState(5)
    lens = [300, 25, 129, 128, 1, 257]
    mats = [(rs.randn(n, 30) * rs.uniform(0.1, 20)).astype(np.float32) for n in lens]
    buf = io.BytesIO()
    for i, m in enumerate(mats):
        kaldi_io.write_compressed_mat(buf, m, key="u%d" % i)
    buf.seek(0)
    packed = [m for _, m in kaldi_io.read_mat_ark_packed(buf)]
    assert all(isinstance(m, kaldi_io.PackedMatrix) for m in packed)
    eng = E.Engine(E.make_config(30, 0, max_batch=1, max_frames=64), device="cuda:0")
    T = max(lens) + 3
    sizes = [len(m.payload) for m in packed]
    offsets = np.zeros(len(packed), np.int64)
    offsets[1:] = np.cumsum(sizes[:-1])
    payload = np.frombuffer(b"".join(m.payload for m in packed), np.uint8)
    x, rows = eng.decode_packed(payload, offsets, lens, T)
    x = x.cpu().numpy()
    for i, m in enumerate(packed):
        ref = m.decode()                                          # host codec (pinned bit-exact to the reference reader, tests/test_host_io.py)
        assert np.array_equal(x[i, :lens[i]], ref), i
        assert not x[i, lens[i]:].any(), i
    eng.close()


def _write_model(model, seed=0):
    from tf_kaldi_speaker_amd import engine as E
    nnet = os.path.join(model, "nnet")
    os.makedirs(nnet)
    cfg = {"network_type": "tdnn", "loss_func": "softmax", "pooling_type": "statistics_pooling", "embedding_node": "tdnn6_dense", "seed": 0,
           "last_layer_no_bn": False, "last_layer_linear": False, "weight_l2_regularizer": 1e-2, "batchnorm_momentum": 0.99,
           "optimizer": "sgd", "num_nodes_pooling_layer": 1500, "num_nodes_last_layer": 512, "feature_norm": False}
    json.dump(cfg, open(os.path.join(nnet, "config.json"), "w"))
    open(os.path.join(nnet, "feature_dim"), "w").write("30\n")
    eng = E.Engine(E.make_config(30, 10, max_batch=1, max_frames=100), device="cuda:0")
    eng.init_variables(seed=seed)
    V = eng.get_variables()
    np.savez(os.path.join(nnet, "model-1.npz"), **V)
    eng.close()
    open(os.path.join(nnet, "checkpoint"), "w").write('model_checkpoint_path: "model-1"\nall_model_checkpoint_paths: "model-1"\n')
    return V


def test_extract_driver_batched_equals_one_at_a_time(tmp_path):
    """nnet/lib/extract.py on a mixed archive ('CM ' and 'FM ' matrices; an utterance shorter than --min-chunk-size, one longer than
    --chunk-size, a run of similar lengths): same keys in archive order, embeddings equal to Trainer.predict one utterance at a
    time (1e-6) and to the oracle's chunk / length-weighted-average arithmetic (5e-5)."""
    from tf_kaldi_speaker_amd.dataset import kaldi_io
    model = str(tmp_path / "exp")
    V32 = _write_model(model)
    V = {k: v.astype(np.float64) for k, v in V32.items()}
    rs = np.random.RandomState(9)
    lens = [90, 20, 410, 33, 150, 149, 151, 25, 700]
    mats = {"utt%02d" % i: (rs.randn(n, 30) * 2).astype(np.float32) for i, n in enumerate(lens)}
    ark_in, ark_out = str(tmp_path / "in.ark"), str(tmp_path / "out.ark")
    with open(ark_in, "wb") as f:
        for i, (k, m) in enumerate(mats.items()):
            (kaldi_io.write_mat if i % 4 == 3 else kaldi_io.write_compressed_mat)(f, m, key=k)
    feats = dict(kaldi_io.read_mat_ark(ark_in))                   # what the network sees (the codec is lossy)
    env = dict(os.environ, TF_KALDI_ROOT=PKG, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, os.path.join(PKG, "nnet", "lib", "extract.py"), "--chunk-size", "300", "--min-chunk-size", "25", model,
                        "ark:" + ark_in, "ark:" + ark_out], env=env, cwd=PKG, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = list(kaldi_io.read_vec_flt_ark(ark_out))
    assert [k for k, _ in out] == [k for k, m in mats.items() if m.shape[0] >= 25]
    assert "Key utt01 length too short, 20 < 25, skip." in r.stderr and "Key utt02 length 410 > 300, split to 2 segments" in r.stderr
    out = dict(out)
    # one at a time through the same model
    sys.path.insert(0, PKG)
    try:
        from model.trainer import Trainer
        from misc.utils import Params, utterance_embedding
        tr = Trainer(Params(os.path.join(model, "nnet", "config.json")), model, single_cpu=True)
        tr.build("predict", dim=30)
        for k, e in out.items():
            one, _ = utterance_embedding(tr.predict, feats[k], 300, False)
            assert rel_err(e, one.astype(np.float64)) <= 1e-6, (k, rel_err(e, one.astype(np.float64)))
        tr.close()
    finally:
        sys.path.remove(PKG)
    from tf_kaldi_speaker_amd.misc.utils import split_into_chunks
    cfg_o = O.Config(feat_dim=30, num_speakers=0)
    for k in ("utt00", "utt02", "utt08"):
        f = feats[k].astype(np.float64)
        chunks = split_into_chunks(len(f), 300)              # (pinned against extract.py:69-79 in tests/test_host_io.py)
        embs = [O.tdnn_forward(V, f[None, s:s + n], cfg_o, False)[1]["tdnn6_dense"][0] for s, n in chunks]
        ref = (np.array(embs) * np.array([n for _, n in chunks])[:, None]).sum(0) / sum(n for _, n in chunks)
        assert rel_err(out[k], ref) <= 5e-5, (k, rel_err(out[k], ref))


def test_predict_batch_rejects_wrong_dimension_and_short_items(tmp_path):
    """the GPU 'CM ' decoder takes the column stride from the model's feature dimension, so a matrix of another width must
    be refused (it would decode garbage and read past its record), and an utterance shorter than the network's receptive field has no valid
    output frame (the pooling would clamp to a frame that sees padding): both raise before anything is enqueued."""
    from tf_kaldi_speaker_amd.dataset import kaldi_io
    import io
    model = str(tmp_path / "exp")
    _write_model(model)
    sys.path.insert(0, PKG)
    try:
        from model.trainer import Trainer
        from misc.utils import Params
        tr = Trainer(Params(os.path.join(model, "nnet", "config.json")), model, single_cpu=True)
        tr.build("predict", dim=30)
        rs = np.random.RandomState(3)
        good = rs.randn(40, 30).astype(np.float32)
        assert tr.predict_batch([good]).shape == (1, 512)
        assert tr.engine.min_frames == 15
        with pytest.raises(ValueError, match="expects \\[frames, 30\\]"):
            tr.predict_batch([good, rs.randn(40, 24).astype(np.float32)])
        buf = io.BytesIO()
        kaldi_io.write_compressed_mat(buf, rs.randn(50, 24).astype(np.float32), key="narrow")
        buf.seek(0)
        packed = [m for _, m in kaldi_io.read_mat_ark_packed(buf)]
        assert isinstance(packed[0], kaldi_io.PackedMatrix) and packed[0].shape == (50, 24)
        with pytest.raises(ValueError, match="expects \\[frames, 30\\]"):
            tr.predict_batch(packed)
        with pytest.raises(ValueError, match="receptive field"):
            tr.predict_batch([good, rs.randn(14, 30).astype(np.float32)])
        with pytest.raises(ValueError, match="frame counts must lie in"):
            tr.engine.forward_lengths(np.zeros((2, 40, 30), np.float32), np.array([40, 9], np.int32))
        tr.close()
    finally:
        sys.path.remove(PKG)
