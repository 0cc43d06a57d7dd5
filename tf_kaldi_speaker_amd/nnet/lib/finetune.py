#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/finetune.py (same CLI):

    python nnet/lib/finetune.py [-c] [--checkpoint C] --config CFG train_dir train_spklist valid_dir valid_spklist \
        pretrain_model finetune_model

Fine-tunes a pre-trained model: the chosen checkpoint (default: best by valid_loss) is copied into the new model
directory as step 0, variables matching params.noload_var_list keep their fresh initialisation, variables (and BN update
ops) matching params.noupdate_var_list are frozen; then the train.py epoch loop (same side-car files).  To resume, use -c.
"""
import argparse

import train as _train      # the shared epoch loop (nnet/lib/train.py, same directory)

parser = argparse.ArgumentParser()
parser.add_argument("-c", "--cont", action="store_true", help="Continue training from an existing model.")
parser.add_argument("--checkpoint", type=str, default="-1",
                    help="The checkpoint in the pre-trained model. The default is to load the BEST checkpoint (according to valid_loss)")
parser.add_argument("--config", type=str, help="The configuration file.")
parser.add_argument("train_dir", type=str, help="The data directory of the training set.")
parser.add_argument("train_spklist", type=str, help="The spklist file maps the TRAINING speakers to the indices.")
parser.add_argument("valid_dir", type=str, help="The data directory of the validation set.")
parser.add_argument("valid_spklist", type=str, help="The spklist maps the VALID speakers to the indices.")
parser.add_argument("pretrain_model", type=str, help="The pre-trained model directory.")
parser.add_argument("finetune_model", type=str, help="The fine-tuned model directory")

if __name__ == "__main__":
    args = parser.parse_args()
    args.model = args.finetune_model
    _train.run(args, finetune=True)
