#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/train_lr_learning.py and finetune_lr_learning.py (same CLIs): the learning-rate
range test of Trainer.train_tune_lr; writes <model>/nnet/learning_rate_tuning ("step lr loss" per period).

    python nnet/lib/train_lr_learning.py [--tune_period N] --config CFG train_dir train_spklist valid_dir valid_spklist model
    python nnet/lib/train_lr_learning.py [--tune_period N] [--checkpoint C] --config CFG train_dir train_spklist valid_dir \
        valid_spklist pretrain_model finetune_model          (seven positionals: the fine-tuning form)
"""
import argparse
import logging
import os
import random
import sys

import numpy as np

from misc.utils import save_codes_and_config, get_pretrain_model
from model.trainer import Trainer
from dataset.data_loader import KaldiDataRandomQueue
from dataset.kaldi_io import FeatureReader

parser = argparse.ArgumentParser()
parser.add_argument("--tune_period", type=int, default=100, help="How many steps per learning rate.")
parser.add_argument("--checkpoint", type=str, default="-1", help="The checkpoint in the pre-trained model (fine-tuning form).")
parser.add_argument("--config", type=str, help="The configuration file.")
parser.add_argument("train_dir", type=str, help="The data directory of the training set.")
parser.add_argument("train_spklist", type=str, help="The spklist file maps the TRAINING speakers to the indices.")
parser.add_argument("valid_dir", type=str, help="The data directory of the validation set.")
parser.add_argument("valid_spklist", type=str, help="The spklist maps the VALID speakers to the indices.")
parser.add_argument("model", type=str, nargs="+", help="model   |   pretrain_model finetune_model")

if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO, format="%(levelname)s:%(name)s:%(message)s")
    log = logging.getLogger("tf_kaldi_speaker_amd")
    args = parser.parse_args()
    if len(args.model) not in (1, 2):
        sys.exit("expected `model` or `pretrain_model finetune_model`")
    model = args.model[-1]
    params = save_codes_and_config(False, model, args.config)
    model_dir = os.path.join(model, "nnet")
    if len(args.model) == 2:
        get_pretrain_model(os.path.join(args.model[0], "nnet"), model_dir, args.checkpoint)
    random.seed(params.seed)
    np.random.seed(params.seed)
    dim = FeatureReader(args.train_dir).get_dim()
    with open(os.path.join(model_dir, "feature_dim"), "w") as f:
        f.write("%d\n" % dim)
    num_total_train_speakers = KaldiDataRandomQueue(args.train_dir, args.train_spklist).num_total_speakers
    log.info("There are %d speakers in the training set and the dim is %d" % (num_total_train_speakers, dim))
    trainer = Trainer(params, model)
    trainer.build("train", dim=dim, loss_type=params.loss_func, num_speakers=num_total_train_speakers)
    trainer.build("valid", dim=dim, loss_type=params.loss_func, num_speakers=num_total_train_speakers)
    if len(args.model) == 2:
        trainer.get_finetune_model(params.noload_var_list)
    trainer.train_tune_lr(args.train_dir, args.train_spklist, args.tune_period)
    trainer.close()
    log.info("Finish tuning.")
