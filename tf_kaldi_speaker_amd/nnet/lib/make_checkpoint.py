#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/make_checkpoint.py: point nnet/checkpoint at the chosen step
(-1 = best epoch by valid_loss, "last", or an explicit step)."""
import argparse
import logging
import os

from misc.utils import get_checkpoint

if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--checkpoint", type=str, default="-1",
                        help="The checkpoint to load. The default is to load the BEST checkpoint (according to valid_loss).")
    parser.add_argument("model_dir", type=str, help="The model directory.")
    args = parser.parse_args()
    checkpoint = get_checkpoint(os.path.join(args.model_dir, "nnet"), args.checkpoint)
    print("Set the checkpoint to %s" % checkpoint)
