#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/train_lr_learning.py and finetune_lr_learning.py (same CLIs): the learning-rate
range test of Trainer.train_tune_lr; writes <model>/nnet/learning_rate_tuning ("step lr loss" per period).

    python nnet/lib/train_lr_learning.py [--tune_period N] --config CFG train_dir train_spklist valid_dir valid_spklist model
    python nnet/lib/train_lr_learning.py [--tune_period N] [--checkpoint C] --config CFG train_dir train_spklist valid_dir \
        valid_spklist pretrain_model finetune_model          (seven positionals: the fine-tuning form)
"""
import os
import sys

import _cli
from misc.utils import save_codes_and_config, get_pretrain_model
from model.trainer import Trainer
from dataset.data_loader import KaldiDataRandomQueue


def main():
    log = _cli.logger()
    args = _cli.parser_for("tune_period", "checkpoint", "config", "train_dir", "train_spklist", "valid_dir", "valid_spklist", "models").parse_args()
    if len(args.model) not in (1, 2):
        sys.exit("expected `model` or `pretrain_model finetune_model`")
    finetune, model = len(args.model) == 2, args.model[-1]
    params = save_codes_and_config(False, model, args.config)
    nnet = os.path.join(model, "nnet")
    if finetune:
        get_pretrain_model(os.path.join(args.model[0], "nnet"), nnet, args.checkpoint)
    _cli.seed_from(params)
    dim = _cli.feature_dim(args.train_dir)
    with open(os.path.join(nnet, "feature_dim"), "w") as f:
        f.write("%d\n" % dim)
    speakers = KaldiDataRandomQueue(args.train_dir, args.train_spklist).num_total_speakers
    log.info("There are %d speakers in the training set and the dim is %d" % (speakers, dim))
    trainer = Trainer(params, model)
    for mode in ("train", "valid"):
        trainer.build(mode, dim=dim, loss_type=params.loss_func, num_speakers=speakers)
    if finetune:
        trainer.get_finetune_model(params.noload_var_list)
    trainer.train_tune_lr(args.train_dir, args.train_spklist, args.tune_period)
    trainer.close()
    log.info("Finish tuning.")


if __name__ == "__main__":
    main()
