"""Throughput of the real training entry point: Trainer.train (one epoch of the reference's API: load checkpoint, native loader,
hot loop with the reference's logging cadence, save) on a synthetic Kaldi directory, shipped batch shape (64 speakers x 2
segments, T ~ U[200,400], 30-dim, 7351 output classes via a padded spklist)."""
import json, os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.kaldi_fixture import make_data_dir
from tf_kaldi_speaker_amd.misc.utils import Params
from tf_kaldi_speaker_amd.model.trainer import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
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
model = os.path.join(root, "exp"); os.makedirs(os.path.join(model, "nnet"))
tr = Trainer(Params(cfg_path), model)
tr.build("train", dim=30, loss_type=cfg["loss_func"], num_speakers=7351)
tr.train(data, spklist, 0.01)          # epoch 1: includes engine warm-up
t0 = time.perf_counter()
tr.train(data, spklist, 0.01)          # epoch 2: load checkpoint, `steps` steps, save
dt = time.perf_counter() - t0
print("Trainer.train: %d steps x 128 chunks in %.2f s = %.0f chunks/s (checkpoint load + save of 39 MB variables included)" % (steps, dt, steps * 128 / dt))
tr.close()
