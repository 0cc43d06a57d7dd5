#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/finetune.py (same CLI):

    python nnet/lib/finetune.py [-c] [--checkpoint C] --config CFG train_dir train_spklist valid_dir valid_spklist \
        pretrain_model finetune_model

Fine-tunes a pre-trained model: the chosen checkpoint (default: best by valid_loss) is copied into the new model
directory as step 0, variables matching params.noload_var_list keep their fresh initialisation, variables (and BN update
ops) matching params.noupdate_var_list are frozen; then the train.py epoch loop (same side-car files).  To resume, use -c.
"""
import _cli
import train as _train      # the shared epoch loop (nnet/lib/train.py, same directory)

if __name__ == "__main__":
    args = _cli.parser_for("cont", "checkpoint", "config", "train_dir", "train_spklist", "valid_dir", "valid_spklist", "pretrain_model",
                           "finetune_model").parse_args()
    args.model = args.finetune_model
    _train.run(args, finetune=True)
