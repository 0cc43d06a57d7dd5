#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/train.py (same CLI, run_train_nnet.sh:64-65):

    python nnet/lib/train.py [-c] --config C train_dir train_spklist valid_dir valid_spklist model

Epoch loop, learning-rate halving, early stop and the side-car files nnet/{feature_dim,
learning_rate,valid_loss} behave as the reference's (train.py:26-143; the bookkeeping lives in
misc.utils.EpochLedger); the graph runs on the MI355X engine.  Under torch.distributed.run (WORLD_SIZE > 1) every rank trains on
its own batches and gradients are all-reduced over RCCL; rank 0 owns the model directory.
"""
import os
import sys

import _cli
from misc.utils import save_codes_and_config, compute_cos_pairwise_eer, Params, EpochLedger, checkpoint_step
from model.trainer import Trainer
from dataset.data_loader import KaldiDataRandomQueue


def main():
    run(_cli.parser_for("cont", "config", "train_dir", "train_spklist", "valid_dir", "valid_spklist", "model").parse_args())


class _Ranks(object):
    """The process group of a data-parallel run (one process per GPU under torch.distributed.run), or a single process."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.dist = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            import _lib                                        # $TF_KALDI_ROOT/_lib.py (PYTHONPATH=$TF_KALDI_ROOT)
            torch.cuda.set_device(_lib.local_device_index())
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(_lib.dist_backend())      # "nccl" (= RCCL); gloo only in the XV_SHARE_GPU test mode
            self.dist = dist

    @property
    def first(self):
        return self.rank == 0

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def share(self, values):
        """Rank 0's list of floats on every rank."""
        if self.dist is None:
            return values
        import torch
        msg = torch.tensor([float(v) for v in values] if self.first else [0.0] * len(values), dtype=torch.float64, device="cuda")
        self.dist.broadcast(msg, 0)
        return [float(v) for v in msg.tolist()]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def run(args, finetune=False):
    """train.py and finetune.py (reference finetune.py:36-175 = train.py plus: the pre-trained checkpoint copied in as step 0,
    params.noupdate_var_list frozen, params.noload_var_list re-initialised, one evaluation before the first epoch).
    args.model is the (fine-tuned) model directory.  Rank 0 owns the directory, evaluates and decides; its decision is
    shared, every rank trains."""
    log = _cli.logger()
    ranks = _Ranks()
    if ranks.first:
        params = save_codes_and_config(args.cont, args.model, args.config)
    ranks.barrier()
    if not ranks.first:
        params = Params(os.path.join(args.model, "nnet/config.json"))
    model_dir = os.path.join(args.model, "nnet")
    _cli.seed_from(params, offset=ranks.rank)      # every rank draws its own batches

    first_epoch = 0
    if args.cont:
        step = checkpoint_step(model_dir)
        if step is None:
            sys.exit("Cannot load checkpoint from %s" % model_dir)
        first_epoch = int(step / params.num_steps_per_epoch)
    elif finetune:
        if ranks.first:      # the pre-trained model becomes step 0 of the new directory (finetune.py:60-66)
            from misc.utils import get_pretrain_model
            get_pretrain_model(os.path.join(args.pretrain_model, "nnet"), model_dir, args.checkpoint)
        ranks.barrier()

    ledger = EpochLedger(model_dir, params, first_epoch, default_early_stop=5 if finetune else 10)      # finetune.py:117 / train.py:101
    dim = _cli.feature_dim(args.train_dir)
    if ranks.first:
        ledger.write_feature_dim(dim)
    num_total_train_speakers = KaldiDataRandomQueue(args.train_dir, args.train_spklist).num_total_speakers
    log.info("There are %d speakers in the training set and the dim is %d" % (num_total_train_speakers, dim))

    trainer = Trainer(params, args.model)
    trainer.build("train", dim=dim, loss_type=params.loss_func, num_speakers=num_total_train_speakers,
                  noupdate_var_list=params.noupdate_var_list if finetune else None)
    trainer.build("valid", dim=dim, loss_type=params.loss_func, num_speakers=num_total_train_speakers)

    def evaluate():
        loss, embeddings, labels = trainer.valid(args.valid_dir, args.valid_spklist, batch_type=params.batch_type, output_embeddings=True)
        return loss, compute_cos_pairwise_eer(embeddings, labels)

    if finetune and first_epoch == 0:
        if ranks.first:      # no collective inside: get_finetune_model saves with sync_bn=False
            trainer.get_finetune_model(params.noload_var_list)
            log.info("In the beginning: Valid EER: %f" % evaluate()[1])
        ranks.barrier()

    for epoch in range(first_epoch, params.num_epochs):
        trainer.train(args.train_dir, args.train_spklist, ledger.rate(epoch))
        stop = False
        if ranks.first:
            valid_loss, eer = evaluate()
            log.info("[INFO] Valid EER: %f" % eer)
            stop = ledger.close_epoch(epoch, valid_loss, eer)
        next_rate, stop_flag = ranks.share([ledger.rate(epoch + 1) if ranks.first else 0.0, 1.0 if stop else 0.0])
        ledger.adopt(epoch, next_rate)
        if stop_flag > 0.5:
            break
    trainer.close()
    ranks.close()


if __name__ == "__main__":
    main()
