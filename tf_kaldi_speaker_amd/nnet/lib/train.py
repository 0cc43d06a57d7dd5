#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/train.py (same CLI, run_train_nnet.sh:64-65):

    python nnet/lib/train.py [-c] --config C train_dir train_spklist valid_dir valid_spklist model

Epoch loop, learning-rate halving, early stop and the side-car files nnet/{feature_dim,
learning_rate,valid_loss} follow the reference line by line in behaviour (train.py:26-143); the
graph runs on the MI355X engine.  Under torch.distributed.run (WORLD_SIZE > 1) every rank trains on
its own batches and gradients are all-reduced over RCCL; rank 0 owns the model directory.
"""
import argparse
import logging
import os
import random
import re
import sys

import numpy as np

from misc.utils import (ValidLoss, load_lr, load_valid_loss, save_codes_and_config, compute_cos_pairwise_eer, Params,
                        read_checkpoint_state, tune_learning_rate, should_stop)
from model.trainer import Trainer
from dataset.data_loader import KaldiDataRandomQueue
from dataset.kaldi_io import FeatureReader

parser = argparse.ArgumentParser()
parser.add_argument("-c", "--cont", action="store_true", help="Continue training from an existing model.")
parser.add_argument("--config", type=str, help="The configuration file.")
parser.add_argument("train_dir", type=str, help="The data directory of the training set.")
parser.add_argument("train_spklist", type=str, help="The spklist file maps the TRAINING speakers to the indices.")
parser.add_argument("valid_dir", type=str, help="The data directory of the validation set.")
parser.add_argument("valid_spklist", type=str, help="The spklist maps the VALID speakers to the indices.")
parser.add_argument("model", type=str, help="The output model directory.")


def main():
    run(parser.parse_args())


def run(args, finetune=False):
    """The epoch loop shared by train.py and finetune.py (reference finetune.py:36-175 is train.py plus: the pre-trained
    checkpoint copied in as step 0, params.noupdate_var_list frozen, params.noload_var_list re-initialised, an evaluation
    before the first epoch).  args.model is the (fine-tuned) model directory."""
    logging.basicConfig(level=logging.INFO, format="%(levelname)s:%(name)s:%(message)s")
    log = logging.getLogger("tf_kaldi_speaker_amd")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        import _lib                                        # $TF_KALDI_ROOT/_lib.py (PYTHONPATH=$TF_KALDI_ROOT)
        torch.cuda.set_device(_lib.local_device_index())
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(_lib.dist_backend())      # "nccl" (= RCCL); gloo only in the XV_SHARE_GPU test mode
    if rank == 0:
        params = save_codes_and_config(args.cont, args.model, args.config)
    if dist is not None:
        dist.barrier()
        if rank != 0:
            params = Params(os.path.join(args.model, "nnet/config.json"))
    model_dir = os.path.join(args.model, "nnet")
    random.seed(params.seed + rank)
    np.random.seed(params.seed + rank)

    if args.cont:
        current, _ = read_checkpoint_state(model_dir)
        if not current:
            sys.exit("Cannot load checkpoint from %s" % model_dir)
        step = int(next(re.finditer(r"(\d+)(?!.*\d)", os.path.basename(current))).group(0))
        start_epoch = int(step / params.num_steps_per_epoch)
    else:
        if finetune and rank == 0:
            # the pre-trained model becomes step 0 of the new directory: "just like an initialized model" (finetune.py:60-66)
            from misc.utils import get_pretrain_model
            get_pretrain_model(os.path.join(args.pretrain_model, "nnet"), model_dir, args.checkpoint)
        if finetune and dist is not None:
            dist.barrier()
        start_epoch = 0

    learning_rate = params.learning_rate
    learning_rate_array = []
    if os.path.isfile(str(learning_rate)):
        with open(str(learning_rate), "r") as f:
            learning_rate_array = [float(line.strip()) for line in f if line.strip()]
        assert len(learning_rate_array) > params.num_epochs, "The learning rate file is shorter than the num of epochs."
        log.info("Using specified learning rate decay strategy.")
    elif os.path.isfile(os.path.join(model_dir, "learning_rate")):
        learning_rate_array = load_lr(os.path.join(model_dir, "learning_rate"))
        assert len(learning_rate_array) == start_epoch + 1, "Not enough learning rates in the learning_rate file."
    else:
        learning_rate_array = [float(learning_rate)] * (start_epoch + 1)

    dim = FeatureReader(args.train_dir).get_dim()
    if rank == 0:
        with open(os.path.join(model_dir, "feature_dim"), "w") as f:
            f.write("%d\n" % dim)
    num_total_train_speakers = KaldiDataRandomQueue(args.train_dir, args.train_spklist).num_total_speakers
    log.info("There are %d speakers in the training set and the dim is %d" % (num_total_train_speakers, dim))

    min_valid_loss = ValidLoss()
    if os.path.isfile(os.path.join(model_dir, "valid_loss")):
        min_valid_loss = load_valid_loss(os.path.join(model_dir, "valid_loss"))

    trainer = Trainer(params, args.model)
    trainer.build("train", dim=dim, loss_type=params.loss_func, num_speakers=num_total_train_speakers,
                  noupdate_var_list=params.noupdate_var_list if finetune else None)
    trainer.build("valid", dim=dim, loss_type=params.loss_func, num_speakers=num_total_train_speakers)

    if "early_stop_epochs" not in params.dict:
        params.dict["early_stop_epochs"] = 5 if finetune else 10      # finetune.py:117 / train.py:101
    if "min_learning_rate" not in params.dict:
        params.dict["min_learning_rate"] = 1e-5

    if finetune and start_epoch == 0:
        if rank == 0:
            trainer.get_finetune_model(params.noload_var_list)
            valid_loss, valid_embeddings, valid_labels = trainer.valid(args.valid_dir, args.valid_spklist,
                                                                       batch_type=params.batch_type, output_embeddings=True)
            log.info("In the beginning: Valid EER: %f" % compute_cos_pairwise_eer(valid_embeddings, valid_labels))
        if dist is not None:
            dist.barrier()

    for epoch in range(start_epoch, params.num_epochs):
        trainer.train(args.train_dir, args.train_spklist, learning_rate_array[epoch])
        stop = False
        if rank == 0:
            valid_loss, valid_embeddings, valid_labels = trainer.valid(args.valid_dir, args.valid_spklist,
                                                                       batch_type=params.batch_type, output_embeddings=True)
            eer = compute_cos_pairwise_eer(valid_embeddings, valid_labels)
            log.info("[INFO] Valid EER: %f" % eer)
            if not os.path.isfile(str(learning_rate)):
                learning_rate_array.append(tune_learning_rate(epoch, learning_rate_array[epoch], valid_loss, min_valid_loss,
                                                              params.reduce_lr_epochs))
            if epoch == 0:
                with open(os.path.join(model_dir, "learning_rate"), "a") as f:
                    f.write("0 %.8f\n" % learning_rate_array[0])
            with open(os.path.join(model_dir, "learning_rate"), "a") as f:
                f.write("%d %.8f\n" % (epoch + 1, learning_rate_array[epoch + 1]))
            with open(os.path.join(model_dir, "valid_loss"), "a") as f:
                f.write("%d %f %f\n" % (epoch, valid_loss, eer))
            if not os.path.isfile(str(learning_rate)):
                stop = should_stop(epoch, learning_rate_array[epoch + 1], min_valid_loss, params.min_learning_rate,
                                   params.early_stop_epochs)
        if dist is not None:      # every rank must take the same LR / stop decision
            import torch
            msg = torch.tensor([learning_rate_array[epoch + 1] if rank == 0 else 0.0, 1.0 if stop else 0.0],
                               dtype=torch.float64, device="cuda")
            dist.broadcast(msg, 0)
            if rank != 0:
                learning_rate_array.append(float(msg[0].item()))
            stop = bool(msg[1].item() > 0.5)
        if stop:
            break
    trainer.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
