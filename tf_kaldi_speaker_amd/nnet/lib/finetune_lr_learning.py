#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/finetune_lr_learning.py: the fine-tuning form of the learning-rate range test
(seven positionals: ... pretrain_model finetune_model) - one implementation, nnet/lib/train_lr_learning.py."""
from train_lr_learning import main

if __name__ == "__main__":
    main()
