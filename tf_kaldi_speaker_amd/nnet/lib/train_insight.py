#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/train_insight.py (same CLI): evaluate a trained model on a data directory -
validation loss, embeddings, EER - through Trainer.insight.

    python nnet/lib/train_insight.py data_dir data_spklist model
"""
import _cli
from misc.utils import save_codes_and_config, compute_cos_pairwise_eer
from model.trainer import Trainer

if __name__ == "__main__":
    log = _cli.logger()
    args = _cli.parser_for("data_dir", "data_spklist", "model").parse_args()
    params = save_codes_and_config(True, args.model, None)
    _cli.seed_from(params)
    trainer = Trainer(params, args.model)
    trainer.build("valid", dim=_cli.feature_dim(args.data_dir), loss_type=params.loss_func, num_speakers=_cli.count_lines(args.data_spklist))
    _, embeddings, labels = trainer.insight(args.data_dir, args.data_spklist, batch_type=params.batch_type, output_embeddings=True)
    log.info("EER: %f" % compute_cos_pairwise_eer(embeddings, labels))
    trainer.close()
