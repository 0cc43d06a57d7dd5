#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/train_insight.py (same CLI): evaluate a trained model on a data directory -
validation loss, embeddings, EER - through Trainer.insight.

    python nnet/lib/train_insight.py data_dir data_spklist model
"""
import argparse
import logging
import random

import numpy as np

from misc.utils import save_codes_and_config, compute_cos_pairwise_eer
from model.trainer import Trainer
from dataset.kaldi_io import FeatureReader

parser = argparse.ArgumentParser()
parser.add_argument("data_dir", type=str, help="The data directory of the dataset.")
parser.add_argument("data_spklist", type=str, help="The spklist maps the speakers to the indices.")
parser.add_argument("model", type=str, help="The output model directory.")

if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO, format="%(levelname)s:%(name)s:%(message)s")
    log = logging.getLogger("tf_kaldi_speaker_amd")
    args = parser.parse_args()
    params = save_codes_and_config(True, args.model, None)
    random.seed(params.seed)
    np.random.seed(params.seed)
    dim = FeatureReader(args.data_dir).get_dim()
    with open(args.data_spklist, "r") as f:
        num_total_train_speakers = len(f.readlines())
    trainer = Trainer(params, args.model)
    trainer.build("valid", dim=dim, loss_type=params.loss_func, num_speakers=num_total_train_speakers)
    valid_loss, valid_embeddings, valid_labels = trainer.insight(args.data_dir, args.data_spklist, batch_type=params.batch_type,
                                                                  output_embeddings=True)
    log.info("EER: %f" % compute_cos_pairwise_eer(valid_embeddings, valid_labels))
    trainer.close()
