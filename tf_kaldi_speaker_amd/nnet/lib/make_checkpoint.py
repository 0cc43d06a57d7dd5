#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/make_checkpoint.py: point nnet/checkpoint at the chosen step
(-1 = best epoch by valid_loss, "last", or an explicit step).

    python nnet/lib/make_checkpoint.py [-c CHECKPOINT] model_dir
"""
import os

import _cli
from misc.utils import get_checkpoint

if __name__ == "__main__":
    _cli.logger()
    args = _cli.parser_for("set_checkpoint", "model_dir").parse_args()
    print("Set the checkpoint to %s" % get_checkpoint(os.path.join(args.model_dir, "nnet"), args.checkpoint))
