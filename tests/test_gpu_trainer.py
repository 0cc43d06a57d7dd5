"""Drop-in boundary on a real MI355X: the reference's Trainer API surface (build / train / valid /
predict / save / load) and the train.py / extract.py / make_checkpoint.py drivers on a synthetic Kaldi
data directory, with embeddings checked against the oracle in inference mode."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import xvector_oracle as O
from tests.kaldi_fixture import make_data_dir

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f32", "f16x3"], autouse=True)
def xv_precision(request, monkeypatch):
    """Every test of this module runs twice: fp32-input MFMA and the split-precision (f16x3) path,
    against the same oracle and the same tolerances."""
    monkeypatch.setenv("XV_PRECISION", request.param)
    return request.param

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tf_kaldi_speaker_amd")

CONFIG = {
    "Note": "shape of egs/voxceleb/v1/nnet_conf/tdnn_amsoftmax_m0.20_linear_bn_1e-2.json, shrunk loop counts",
    "seed": 0, "network_type": "tdnn", "last_layer_no_bn": False, "last_layer_linear": True, "feature_norm": False,
    "loss_func": "additive_margin_softmax", "amsoftmax_m": 0.20, "amsoftmax_lambda_min": 0, "amsoftmax_lambda_base": 1000,
    "amsoftmax_lambda_gamma": 0.0001, "amsoftmax_lambda_power": 5,
    "batch_type": "softmax", "pooling_type": "statistics_pooling", "embedding_node": "tdnn6_dense",
    "learning_rate": 0.01, "use_nesterov": False, "clip_gradient": False, "clip_gradient_norm": 3,
    "weight_l2_regularizer": 1e-2, "batchnorm_momentum": 0.99,
    "num_epochs": 2, "num_steps_per_epoch": 6, "reduce_lr_epochs": 4, "show_training_progress": 2, "keep_checkpoint_max": 100,
    "save_summary_steps": 10000, "save_checkpoints_steps": 30000, "valid_max_iterations": 1000,
    "num_parallel_datasets": 2, "max_queue_size": 4, "num_speakers_per_batch": 4, "num_segments_per_speaker": 1,
    "min_segment_len": 30, "max_segment_len": 50, "early_stop_epochs": 10, "min_learning_rate": 1e-6,
}


def _oracle_cfg(num_speakers):
    return O.Config(feat_dim=30, num_speakers=num_speakers, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True)


def test_trainer_api_roundtrip(tmp_path):
    from tf_kaldi_speaker_amd.misc.utils import Params
    from tf_kaldi_speaker_amd.model.trainer import Trainer
    data, spklist, mats = make_data_dir(str(tmp_path / "train"), num_spk=6, utts_per_spk=3, min_frames=60, max_frames=110)
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps(CONFIG))
    params = Params(str(cfg_path))
    model = str(tmp_path / "exp")
    os.makedirs(os.path.join(model, "nnet"))
    tr = Trainer(params, model)
    with pytest.raises(NotImplementedError):
        tr.build("train", dim=30, loss_type="no_such_loss", num_speakers=6)
    tr.build("train", dim=30, loss_type=params.loss_func, num_speakers=6)
    tr.build("valid", dim=30, loss_type=params.loss_func, num_speakers=6)
    assert params.dict["num_nodes_pooling_layer"] == 1500 and params.dict["num_nodes_last_layer"] == 512   # defaults inserted
    before = tr.engine.get_variables()
    tr.train(data, spklist, 0.01)
    after = tr.engine.get_variables()
    assert np.abs(after["tdnn/tdnn3_conv/kernel"] - before["tdnn/tdnn3_conv/kernel"]).max() > 0
    assert not np.allclose(after["tdnn/tdnn2_bn/moving_mean"], 0)                # UPDATE_OPS ran
    assert os.path.isfile(os.path.join(model, "nnet", "checkpoint")) and os.path.isfile(os.path.join(model, "nnet", "model-6.npz"))
    assert 'model_checkpoint_path: "' in open(os.path.join(model, "nnet", "checkpoint")).read()
    loss, emb, labels = tr.valid(data, spklist, batch_type="softmax", output_embeddings=True)
    assert np.isfinite(loss) and emb.shape[1] == 512 and emb.shape[0] == labels.shape[0] and emb.shape[0] >= 16
    tr.train(data, spklist, 0.005)                                                   # second epoch resumes from step 6
    assert os.path.isfile(os.path.join(model, "nnet", "model-12.npz"))
    # predict in a fresh Trainer that only builds the predict graph (extract.py:54-58)
    p2 = Params(str(cfg_path))
    tr2 = Trainer(p2, model, single_cpu=True)
    tr2.build("predict", dim=30)
    feat = next(iter(mats.values()))
    e1 = tr2.predict(feat)
    assert e1.shape == (512,)
    e3 = tr2.predict(np.stack([feat[:40], feat[10:50]]))
    assert e3.shape == (2, 512)
    # oracle, inference mode, same variables
    V = {k: v.astype(np.float64) for k, v in tr.engine.get_variables().items()}
    _, ep, _ = O.tdnn_forward(V, feat[None].astype(np.float64), _oracle_cfg(6), False)
    ref = ep["tdnn6_dense"][0]
    assert np.abs(e1 - ref).max() / np.abs(ref).max() < 1e-4         # north_star: embeddings within 1e-4 relative
    p2.embedding_node = "output"
    assert tr2.predict(feat).shape == (512,)
    tr.close()
    tr2.close()


def test_trainer_trains_from_gpu_decoded_batches(tmp_path, monkeypatch):
    """XV_LOADER=gpu_decode: Trainer.train fed by the packed loader + xv_cm_decode (same features bit for bit, tests/test_gpu_loader.py):
    an epoch runs, moves the weights and writes its checkpoint."""
    from tf_kaldi_speaker_amd.misc.utils import Params
    from tf_kaldi_speaker_amd.model.trainer import Trainer
    monkeypatch.setenv("XV_LOADER", "gpu_decode")
    data, spklist, _ = make_data_dir(str(tmp_path / "train"), num_spk=6, utts_per_spk=3, min_frames=60, max_frames=110)
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps(CONFIG))
    model = str(tmp_path / "exp")
    os.makedirs(os.path.join(model, "nnet"))
    tr = Trainer(Params(str(cfg_path)), model)
    tr.build("train", dim=30, loss_type=CONFIG["loss_func"], num_speakers=6)
    before = tr.engine.get_variables()
    tr.train(data, spklist, 0.01)
    after = tr.engine.get_variables()
    assert all(np.isfinite(v).all() for v in after.values())
    assert np.abs(after["tdnn/tdnn1_conv/kernel"] - before["tdnn/tdnn1_conv/kernel"]).max() > 0
    assert os.path.isfile(os.path.join(model, "nnet", "model-6.npz"))
    tr.close()


ATT_KEYS = {   # the attention block of egs/voxceleb/v1/nnet_conf/tdnn_amsoftmax_m0.20_linear_bn_1e-2_tdnn4_att.json:17-28 (key widths shrunk)
    "pooling_type": "self_attention", "att_key_input": "tdnn4_relu", "att_key_num_nodes": [96, 64], "att_key_network_type": 3,
    "att_value_input": "tdnn5_relu", "att_value_num_nodes": [], "att_value_network_type": 0, "att_apply_nonlinear": False,
    "att_use_scale": True, "att_num_heads": 1, "att_split_key": False, "att_penalty_term": 0,
}


def test_trainer_with_self_attention_pooling(tmp_path):
    """The shipped attention configuration through the Trainer: train one epoch, checkpoint round trip, embeddings of a fresh
    predict-only Trainer against the oracle in inference mode; unsupported attention options are refused by name."""
    from tf_kaldi_speaker_amd.misc.utils import Params
    from tf_kaldi_speaker_amd.model.trainer import Trainer
    data, spklist, mats = make_data_dir(str(tmp_path / "train"), num_spk=6, utts_per_spk=3, min_frames=60, max_frames=110)
    cfg = dict(CONFIG, **ATT_KEYS)
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps(cfg))
    params = Params(str(cfg_path))
    model = str(tmp_path / "exp")
    os.makedirs(os.path.join(model, "nnet"))
    tr = Trainer(params, model)
    tr.build("train", dim=30, loss_type=params.loss_func, num_speakers=6)
    names = list(tr.engine.table)
    assert "tdnn/attention/att_key0/att_key0_dense/kernel" in names and "tdnn/attention/query" in names
    assert names.index("tdnn/attention/query") < names.index("tdnn/tdnn6_dense/kernel")      # graph order of the reference
    before = tr.engine.get_variables()
    tr.train(data, spklist, 0.01)
    after = tr.engine.get_variables()
    for k in ("tdnn/attention/att_key1/att_key1_dense/kernel", "tdnn/attention/query", "tdnn/attention/att_key0/att_key0_bn/moving_mean"):
        assert np.abs(after[k] - before[k]).max() > 0, k
    p2 = Params(str(cfg_path))
    tr2 = Trainer(p2, model, single_cpu=True)
    tr2.build("predict", dim=30)
    feat = next(iter(mats.values()))
    e1 = tr2.predict(feat)
    V = {k: v.astype(np.float64) for k, v in after.items()}
    cfg_o = O.Config(feat_dim=30, num_speakers=6, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True,
                     pooling_type="self_attention", att_key_num_nodes=(96, 64))
    _, ep, _ = O.tdnn_forward(V, feat[None].astype(np.float64), cfg_o, False)
    ref = ep["tdnn6_dense"][0]
    assert np.abs(e1 - ref).max() / np.abs(ref).max() < 1e-4
    tr.close()
    tr2.close()
    bad = dict(cfg, att_num_heads=4)
    (tmp_path / "bad.json").write_text(json.dumps(bad))
    with pytest.raises(NotImplementedError, match="att_num_heads"):
        Trainer(Params(str(tmp_path / "bad.json")), model).build("train", dim=30, loss_type="additive_margin_softmax", num_speakers=6)


def test_finetuning_helpers(tmp_path):
    """noupdate_var_list / set_trainable_variables / get_finetune_model / train_tune_lr / insight (trainer.py:379-403, 522-590, 728-920)."""
    from tf_kaldi_speaker_amd.misc.utils import Params
    from tf_kaldi_speaker_amd.model.trainer import Trainer
    data, spklist, mats = make_data_dir(str(tmp_path / "train"), num_spk=6, utts_per_spk=3, min_frames=60, max_frames=110)
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps(dict(CONFIG, optimizer="momentum", momentum=0.9)))
    model = str(tmp_path / "exp")
    os.makedirs(os.path.join(model, "nnet"))
    # pre-train one epoch
    tr = Trainer(Params(str(cfg_path)), model)
    tr.build("train", dim=30, loss_type="additive_margin_softmax", num_speakers=6)
    tr.train(data, spklist, 0.01)
    pre = tr.engine.get_variables()
    tr.close()
    # fine-tune: a new speaker set (softmax layer re-initialised), tdnn1-3 frozen incl. their BN statistics
    tr = Trainer(Params(str(cfg_path)), model)
    tr.build("train", dim=30, loss_type="additive_margin_softmax", num_speakers=6, noupdate_var_list=["tdnn1", "tdnn2", "tdnn3"])
    tr.build("valid", dim=30, loss_type="additive_margin_softmax", num_speakers=6)
    tr.get_finetune_model(["softmax"])
    start = tr.engine.get_variables()
    nnet = os.path.join(model, "nnet")
    assert os.path.isfile(os.path.join(nnet, "model-0.npz")) and any(f.endswith(".bak") for f in os.listdir(nnet))
    assert np.array_equal(start["tdnn/tdnn2_conv/kernel"], pre["tdnn/tdnn2_conv/kernel"])
    assert not np.array_equal(start["softmax/output/kernel"], pre["softmax/output/kernel"])
    tr.train(data, spklist, 0.01)
    after = tr.engine.get_variables()
    for k in after:
        frozen = any(s in k for s in ("tdnn1", "tdnn2", "tdnn3"))
        changed = not np.array_equal(after[k], start[k])
        if frozen:
            assert not changed, k                                   # values AND moving statistics untouched
        elif k.endswith(("kernel", "gamma", "beta", "moving_mean")):
            assert changed, k
    # set_trainable_variables: only the loss layer from now on (BN statistics keep moving)
    tr.set_trainable_variables(["softmax"])
    tr.train(data, spklist, 0.01)
    last = tr.engine.get_variables()
    assert np.array_equal(last["tdnn/tdnn5_dense/kernel"], after["tdnn/tdnn5_dense/kernel"])
    assert not np.array_equal(last["softmax/output/kernel"], after["softmax/output/kernel"])
    assert not np.array_equal(last["tdnn/tdnn5_bn/moving_mean"], after["tdnn/tdnn5_bn/moving_mean"])
    tr.set_trainable_variables(None)
    loss, emb, labels = tr.insight(data, spklist, output_embeddings=True)
    assert np.isfinite(loss) and emb.shape[0] == labels.shape[0]
    tr.train_tune_lr(data, spklist, tune_period=2, tune_times=3)
    lines = open(os.path.join(nnet, "learning_rate_tuning")).read().strip().split("\n")
    assert len(lines) == 3 and lines[0].split()[0] == "0" and abs(float(lines[1].split()[1]) - 1.15e-5) < 1e-6
    tr.close()


def test_drivers_as_run_sh_calls_them(tmp_path):
    """python nnet/lib/train.py ... ; make_checkpoint.py ; extract.py with PYTHONPATH=$TF_KALDI_ROOT (run_train_nnet.sh:30,64)."""
    data, spklist, mats = make_data_dir(str(tmp_path / "train"), num_spk=6, utts_per_spk=3, min_frames=60, max_frames=110)
    vdata, vspk, _ = make_data_dir(str(tmp_path / "valid"), num_spk=6, utts_per_spk=2, min_frames=60, max_frames=110, seed=5)
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps(CONFIG))
    model = str(tmp_path / "exp")
    os.makedirs(model)
    env = dict(os.environ, TF_KALDI_ROOT=PKG, PYTHONPATH=PKG)
    lib = os.path.join(PKG, "nnet", "lib")
    r = subprocess.run([sys.executable, os.path.join(lib, "train.py"), "--config", str(cfg_path), data, spklist, vdata, vspk, model],
                       env=env, cwd=PKG, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    nnet = os.path.join(model, "nnet")
    assert open(os.path.join(nnet, "feature_dim")).read().strip() == "30"
    lr_lines = open(os.path.join(nnet, "learning_rate")).read().strip().split("\n")
    vl_lines = open(os.path.join(nnet, "valid_loss")).read().strip().split("\n")
    assert len(lr_lines) == 3 and lr_lines[0].startswith("0 0.0100") and len(vl_lines) == 2 and len(vl_lines[0].split(" ")) == 3
    assert os.path.isdir(os.path.join(model, "codes", "model")) and os.path.isfile(os.path.join(nnet, "config.json"))
    r = subprocess.run([sys.executable, os.path.join(lib, "make_checkpoint.py"), "--checkpoint", "last", model], env=env, cwd=PKG,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "model-12" in r.stdout, r.stderr[-2000:]
    # extraction: ark in, float-vector ark out; a 130-frame utterance with --chunk-size 60 exercises the split/average path
    from tf_kaldi_speaker_amd.dataset import kaldi_io
    ark_in, ark_out = str(tmp_path / "in.ark"), str(tmp_path / "xvector.ark")
    long_utt = np.concatenate(list(mats.values())[:2])[:130]
    with open(ark_in, "wb") as f:
        kaldi_io.write_mat(f, long_utt, key="long")
        kaldi_io.write_mat(f, list(mats.values())[2], key="short")
        kaldi_io.write_mat(f, long_utt[:20], key="tiny")           # < --min-chunk-size: skipped
    r = subprocess.run([sys.executable, os.path.join(lib, "extract.py"), "--node", "tdnn6_dense", "--chunk-size", "60", "--min-chunk-size", "25",
                        model, "ark:" + ark_in, "ark:" + ark_out], env=env, cwd=PKG, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = dict(kaldi_io.read_vec_flt_ark(ark_out))
    assert sorted(out) == ["long", "short"] and out["long"].shape == (512,) and out["long"].dtype == np.float32
    # oracle re-derivation of the chunked embedding from the saved variables
    ck = np.load(os.path.join(nnet, "model-12.npz"))
    V = {k: ck[k].astype(np.float64) for k in ck.files if not k.startswith("__")}
    cfg_o = _oracle_cfg(6)
    embs, lens = [], []
    for s, n in [(0, 60), (30, 60), (60, 60), (90, 40)]:
        _, ep, _ = O.tdnn_forward(V, long_utt[None, s:s + n].astype(np.float64), cfg_o, False)
        embs.append(ep["tdnn6_dense"][0]); lens.append(n)
    ref = (np.array(embs) * np.array(lens)[:, None]).sum(0) / sum(lens)
    assert np.abs(out["long"] - ref).max() / np.abs(ref).max() < 1e-4
    # fine-tuning drivers (finetune.py, train_lr_learning.py in its fine-tuning form, train_insight.py): a new loss layer on the frozen
    # tdnn1-3 of the model trained above
    ft_cfg = dict(CONFIG, num_epochs=1, noupdate_var_list=["tdnn1", "tdnn2", "tdnn3"], noload_var_list=["softmax"])
    ft_path = tmp_path / "finetune.json"
    ft_path.write_text(json.dumps(ft_cfg))
    ft_model = str(tmp_path / "exp_ft")
    os.makedirs(ft_model)
    r = subprocess.run([sys.executable, os.path.join(lib, "finetune.py"), "--checkpoint", "last", "--config", str(ft_path), data, spklist, vdata,
                        vspk, model, ft_model], env=env, cwd=lib, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "In the beginning: Valid EER" in r.stderr
    ft = np.load(os.path.join(ft_model, "nnet", "model-6.npz"))
    assert np.array_equal(ft["tdnn/tdnn2_conv/kernel"], ck["tdnn/tdnn2_conv/kernel"])              # frozen
    assert np.array_equal(ft["tdnn/tdnn1_bn/moving_mean"], ck["tdnn/tdnn1_bn/moving_mean"])          # its BN update op too
    assert not np.array_equal(ft["tdnn/tdnn5_dense/kernel"], ck["tdnn/tdnn5_dense/kernel"])
    assert not np.array_equal(ft["softmax/output/kernel"], ck["softmax/output/kernel"])
    tune_model = str(tmp_path / "exp_tune")
    os.makedirs(tune_model)
    r = subprocess.run([sys.executable, os.path.join(lib, "finetune_lr_learning.py"), "--tune_period", "1", "--checkpoint", "last", "--config",
                        str(ft_path), data, spklist, vdata, vspk, model, tune_model], env=dict(env, XV_TUNE_TIMES="3"), cwd=lib,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(open(os.path.join(tune_model, "nnet", "learning_rate_tuning")).read().strip().split("\n")) == 3
    r = subprocess.run([sys.executable, os.path.join(lib, "train_insight.py"), vdata, vspk, ft_model], env=env, cwd=lib, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "EER:" in r.stderr, r.stderr[-3000:]


def test_train_driver_with_two_ranks_sharing_the_gpu(tmp_path, xv_precision):
    """nnet/lib/train.py under torch.distributed.run with WORLD_SIZE=2 (both ranks on the one GPU of the box, gloo transport:
    XV_SHARE_GPU=1): rank 0 snapshots the config, every rank draws its own batches, gradients are averaged, the LR / stop
    decision is broadcast, BN moving statistics are averaged into the checkpoint rank 0 writes, and the next epoch resumes
    from it on both ranks."""
    if xv_precision == "f32":
        pytest.skip("one precision is enough for the process wiring")
    data, spklist, _ = make_data_dir(str(tmp_path / "train"), num_spk=6, utts_per_spk=3, min_frames=60, max_frames=110)
    vdata, vspk, _ = make_data_dir(str(tmp_path / "valid"), num_spk=6, utts_per_spk=2, min_frames=60, max_frames=110, seed=5)
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps(CONFIG))
    model = str(tmp_path / "exp")
    os.makedirs(model)
    env = dict(os.environ, TF_KALDI_ROOT=PKG, PYTHONPATH=PKG, XV_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29300 + os.getpid() % 150), os.path.join(PKG, "nnet", "lib", "train.py"), "--config", str(cfg_path),
           data, spklist, vdata, vspk, model]
    r = subprocess.run(cmd, env=env, cwd=PKG, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    nnet = os.path.join(model, "nnet")
    lr_lines = open(os.path.join(nnet, "learning_rate")).read().strip().split("\n")
    vl_lines = open(os.path.join(nnet, "valid_loss")).read().strip().split("\n")
    assert len(lr_lines) == 3 and lr_lines[0].startswith("0 0.0100") and len(vl_lines) == 2
    ck = np.load(os.path.join(nnet, "model-12.npz"))         # 2 epochs x 6 steps per rank: the step count is per rank
    assert all(np.isfinite(ck[k]).all() for k in ck.files)
    assert float(np.abs(ck["tdnn/tdnn1_bn/moving_mean"]).max()) > 0


def test_finetune_driver_with_two_ranks_sharing_the_gpu(tmp_path, xv_precision):
    """nnet/lib/finetune.py under torch.distributed.run with WORLD_SIZE=2 (the step-0 checkpoint of the pre-trained model is
    written by rank 0 alone while the other rank waits in a barrier, so that save must not issue a collective): the run finishes, the
    frozen layers stay bit-identical to the pre-trained model and the new loss layer is trained."""
    if xv_precision == "f32":
        pytest.skip("one precision is enough for the process wiring")
    data, spklist, _ = make_data_dir(str(tmp_path / "train"), num_spk=6, utts_per_spk=3, min_frames=60, max_frames=110)
    vdata, vspk, _ = make_data_dir(str(tmp_path / "valid"), num_spk=6, utts_per_spk=2, min_frames=60, max_frames=110, seed=5)
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps(dict(CONFIG, num_epochs=1)))
    model = str(tmp_path / "exp")
    os.makedirs(model)
    env = dict(os.environ, TF_KALDI_ROOT=PKG, PYTHONPATH=PKG, XV_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    lib = os.path.join(PKG, "nnet", "lib")
    r = subprocess.run([sys.executable, os.path.join(lib, "train.py"), "--config", str(cfg_path), data, spklist, vdata, vspk, model],
                       env=env, cwd=PKG, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    ck = np.load(os.path.join(model, "nnet", "model-6.npz"))
    ft_path = tmp_path / "finetune.json"
    ft_path.write_text(json.dumps(dict(CONFIG, num_epochs=1, noupdate_var_list=["tdnn1", "tdnn2", "tdnn3"], noload_var_list=["softmax"])))
    ft_model = str(tmp_path / "exp_ft")
    os.makedirs(ft_model)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + os.getpid() % 150), os.path.join(lib, "finetune.py"), "--checkpoint", "last", "--config", str(ft_path),
           data, spklist, vdata, vspk, model, ft_model]
    r = subprocess.run(cmd, env=env, cwd=lib, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    nnet = os.path.join(ft_model, "nnet")
    assert os.path.isfile(os.path.join(nnet, "model-0.npz"))                                           # the pre-trained model as step 0
    ft = np.load(os.path.join(nnet, "model-6.npz"))
    assert all(np.isfinite(ft[k]).all() for k in ft.files)
    assert np.array_equal(ft["tdnn/tdnn2_conv/kernel"], ck["tdnn/tdnn2_conv/kernel"])                   # frozen on both ranks
    assert not np.array_equal(ft["tdnn/tdnn5_dense/kernel"], ck["tdnn/tdnn5_dense/kernel"])
    assert not np.array_equal(ft["softmax/output/kernel"], ck["softmax/output/kernel"])


@pytest.mark.parametrize("optimizer", ["momentum", "adam"])
def test_resume_from_checkpoint_is_bit_identical(tmp_path, optimizer):
    """Four optimiser steps in one go == two steps, save, a fresh Trainer that loads the checkpoint, two more steps: variables,
    BN moving statistics, optimiser slots and (for Adam) the update count all travel through the checkpoint."""
    from tf_kaldi_speaker_amd.misc.utils import Params
    from tf_kaldi_speaker_amd.model.trainer import Trainer
    cfg = dict(CONFIG, optimizer=optimizer, momentum=0.9, use_nesterov=(optimizer == "momentum"))
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps(cfg))
    rs = np.random.RandomState(4)
    batches = [(rs.randn(6, 40 + 3 * i, 30).astype(np.float32), rs.randint(0, 6, 6).astype(np.int32)) for i in range(4)]

    def trainer(model):
        os.makedirs(os.path.join(model, "nnet"), exist_ok=True)
        tr = Trainer(Params(str(cfg_path)), model)
        tr.build("train", dim=30, loss_type=cfg["loss_func"], num_speakers=6)
        return tr

    a = trainer(str(tmp_path / "a"))
    assert a.train_batches(iter(batches), 0.02, 0, num_steps=4) == 4
    want = a.engine.variables.cpu().numpy().copy()
    want_opt = a.engine.opt_state.cpu().numpy().copy()
    v_init = None
    a.close()

    b = trainer(str(tmp_path / "b"))
    assert b.train_batches(iter(batches[:2]), 0.02, 0, num_steps=2) == 2
    b.save(2)
    b.close()
    c = trainer(str(tmp_path / "b"))
    assert c.load() == 2
    assert c.train_batches(iter(batches[2:]), 0.02, 2, num_steps=2) == 4
    got = c.engine.variables.cpu().numpy()
    got_opt = c.engine.opt_state.cpu().numpy()
    assert np.array_equal(got, want) and np.array_equal(got_opt, want_opt) and np.abs(want_opt).max() > 0
    c.close()


def test_trainer_reads_and_writes_tensorflow_checkpoints(tmp_path, xv_precision):
    """`save_tf_checkpoint: true` writes the reference's payload format (tf.train.Saver V2, reference trainer.py:318,444) beside the
    .npz, and Trainer.load() restores from such a checkpoint when no .npz is there - the path a pretrained upstream model
    (reference README.md:86-104) takes.  Format restated in misc/tf_checkpoint.py (pinned by tests/test_tf_checkpoint.py)."""
    if xv_precision != "f32":
        pytest.skip("checkpoint I/O does not depend on the GEMM precision")
    from tf_kaldi_speaker_amd.misc.utils import Params
    from tf_kaldi_speaker_amd.misc import tf_checkpoint
    from tf_kaldi_speaker_amd.model.trainer import Trainer
    data, spklist, mats = make_data_dir(str(tmp_path / "train"), num_spk=5, utts_per_spk=3, min_frames=60, max_frames=90)
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps(dict(CONFIG, save_tf_checkpoint=True, num_steps_per_epoch=3)))
    model = str(tmp_path / "exp")
    os.makedirs(os.path.join(model, "nnet"))
    tr = Trainer(Params(str(cfg_path)), model)
    tr.build("train", dim=30, loss_type="additive_margin_softmax", num_speakers=5)
    tr.train(data, spklist, 0.01)
    want = tr.engine.get_variables()
    tr.close()
    prefix = os.path.join(model, "nnet", "model-3")
    assert os.path.isfile(prefix + ".index") and os.path.isfile(prefix + ".data-00000-of-00001")
    stored = tf_checkpoint.read_checkpoint(prefix, verify=True)
    assert set(stored) == set(want) and stored["tdnn/tdnn1_conv/kernel"].shape == (1, 5, 30, 512)
    os.remove(prefix + ".npz")                                   # only the TensorFlow payload is left, as in an upstream model directory
    tr2 = Trainer(Params(str(cfg_path)), model)
    tr2.build("predict", dim=30)
    assert tr2.load() == 3
    got = tr2.engine.get_variables()              # (the predict graph has no loss head: a subset of the training graph's variables)
    assert len(got) >= 40 and set(got) <= set(want)
    for k, v in got.items():
        assert np.array_equal(want[k], v), k
    tr2.close()

