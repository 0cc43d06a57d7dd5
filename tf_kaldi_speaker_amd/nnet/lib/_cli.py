"""The command lines of the nnet/lib drivers, declared once.

The recipes call these scripts with the reference's flags (run_train_nnet.sh:64-65, wrap/extract_wrapper.sh:39-40, run.sh), so the flag
names, positions, types and defaults are the contract; every driver builds its parser from this table by naming the arguments it takes,
in order.  Also the few lines every driver repeats (logging set-up, seeding, the feature dimension / speaker count of a data directory).
"""
import argparse
import logging
import random

import numpy as np

_BEST = "The default is to load the BEST checkpoint (according to valid_loss)."
ARGUMENTS = {
    # options
    "cont": (("-c", "--cont"), dict(action="store_true", help="Continue training from an existing model.")),
    "config": (("--config",), dict(type=str, help="The configuration file.")),
    "checkpoint": (("--checkpoint",), dict(type=str, default="-1", help="The checkpoint in the pre-trained model. " + _BEST)),
    "set_checkpoint": (("-c", "--checkpoint"), dict(type=str, default="-1", help="The checkpoint to load. " + _BEST)),
    "tune_period": (("--tune_period",), dict(type=int, default=100, help="How many steps per learning rate.")),
    "gpu": (("-g", "--gpu"), dict(type=int, default=-1, help="The GPU id (-1: device 0).")),
    "min_chunk_size": (("-m", "--min-chunk-size"), dict(type=int, default=25, help="Segments shorter than this are skipped.")),
    "chunk_size": (("-s", "--chunk-size"), dict(type=int, default=10000, help="Longer segments are split and averaged.")),
    "normalize": (("-n", "--normalize"), dict(action="store_true", help="Normalize the embedding before averaging and output.")),
    "node": (("--node",), dict(type=str, default="", help="The node to output the embeddings.")),
    # positionals
    "train_dir": (("train_dir",), dict(type=str, help="The data directory of the training set.")),
    "train_spklist": (("train_spklist",), dict(type=str, help="The spklist file maps the TRAINING speakers to the indices.")),
    "valid_dir": (("valid_dir",), dict(type=str, help="The data directory of the validation set.")),
    "valid_spklist": (("valid_spklist",), dict(type=str, help="The spklist maps the VALID speakers to the indices.")),
    "data_dir": (("data_dir",), dict(type=str, help="The data directory of the dataset.")),
    "data_spklist": (("data_spklist",), dict(type=str, help="The spklist maps the speakers to the indices.")),
    "model": (("model",), dict(type=str, help="The output model directory.")),
    "models": (("model",), dict(type=str, nargs="+", help="model   |   pretrain_model finetune_model")),
    "model_dir": (("model_dir",), dict(type=str, help="The model directory.")),
    "pretrain_model": (("pretrain_model",), dict(type=str, help="The pre-trained model directory.")),
    "finetune_model": (("finetune_model",), dict(type=str, help="The fine-tuned model directory")),
    "rspecifier": (("rspecifier",), dict(type=str, help="Kaldi feature rspecifier (or ark file).")),
    "wspecifier": (("wspecifier",), dict(type=str, help="Kaldi output wspecifier (or ark file).")),
}


def parser_for(*names, **kw):
    p = argparse.ArgumentParser(**kw)
    for n in names:
        flags, spec = ARGUMENTS[n]
        p.add_argument(*flags, **spec)
    return p


class _BatchedStderr(logging.Handler):
    """The records basicConfig's handler would write to stderr one write() each, collected and written in one piece per flush()
    (extract.py logs a line per utterance as the reference does, several thousand a second: the system calls were a tenth of its
    per-utterance host time).  Flushed by the driver after every window and by logging.shutdown() at exit."""

    def __init__(self):
        logging.Handler.__init__(self)
        self.lines = []

    def emit(self, record):
        self.lines.append(self.format(record))
        if record.levelno >= logging.WARNING or len(self.lines) >= 4096:
            self.flush()

    def flush(self):
        if self.lines:
            import sys
            lines, self.lines = self.lines, []
            sys.stderr.write("\n".join(lines) + "\n")
            sys.stderr.flush()


def logger(batched=False):
    if batched and not logging.getLogger().handlers:
        h = _BatchedStderr()
        h.setFormatter(logging.Formatter("%(levelname)s:%(name)s:%(message)s"))
        logging.getLogger().addHandler(h)
        logging.getLogger().setLevel(logging.INFO)
    else:
        logging.basicConfig(level=logging.INFO, format="%(levelname)s:%(name)s:%(message)s")
    return logging.getLogger("tf_kaldi_speaker_amd")


def seed_from(params, offset=0):
    random.seed(params.seed + offset)
    np.random.seed(params.seed + offset)


def feature_dim(data_dir):
    from dataset.kaldi_io import FeatureReader
    return FeatureReader(data_dir).get_dim()


def count_lines(path):
    with open(path, "r") as f:
        return sum(1 for _ in f)
