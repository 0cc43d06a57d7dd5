#!/usr/bin/env python3
"""Convert between the reference's TensorFlow checkpoints and this package's `.npz` payloads - without TensorFlow.

  tools/tf_ckpt_to_npz.py <model_dir>/nnet/model-1200000            -> <model_dir>/nnet/model-1200000.npz
  tools/tf_ckpt_to_npz.py --to-tf <model_dir>/nnet/model-3000       -> model-3000.index + model-3000.data-00000-of-00001
  tools/tf_ckpt_to_npz.py --list <prefix>                           names, dtypes and shapes in a TF checkpoint

The reference writes its models with tf.train.Saver (model/trainer.py:318,444; the pretrained VoxCeleb / SRE models of its
README.md:86-104 are such files: model-<step>.{index,data-00000-of-00001,meta} + the text file `checkpoint`).  Variable names and
shapes are the same on both sides (tdnn/tdnn1_conv/kernel [1,5,30,512], ..., softmax/output/kernel), so the conversion is a
name -> array map; optimiser slots (`.../Momentum`, `.../Adam`, beta*_power) are dropped on the way in.  `Trainer.load()` also reads a
TF checkpoint directly when no `.npz` of that name exists, so converting is optional.

The on-disk format is restated in tf_kaldi_speaker_amd/misc/tf_checkpoint.py (LevelDB table + tensor_bundle.proto); it could not
be checked against a TensorFlow-written file in the build environment - tests/golden/make_tf_golden.py produces one on a box that has
TF 1.x, and tests/test_tf_checkpoint.py then checks this reader against it.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_kaldi_speaker_amd.misc import tf_checkpoint as T      # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("prefix", help="checkpoint path without extension, e.g. exp/xvector_nnet/nnet/model-1200000")
    ap.add_argument("--to-tf", action="store_true", help="read <prefix>.npz, write a TF V2 checkpoint")
    ap.add_argument("--list", action="store_true", help="only list the variables of the TF checkpoint")
    ap.add_argument("--verify", action="store_true", help="check the CRC32C of every tensor while reading")
    ap.add_argument("--keep-slots", action="store_true", help="keep optimiser slot variables")
    args = ap.parse_args()
    if args.list:
        entries, shards = T.list_variables(args.prefix)
        for name, (dtype, shape, shard, offset, size, _) in sorted(entries.items()):
            print("%-60s %-8s %-22s shard %d @ %d (%d bytes)" % (name, dtype, list(shape), shard, offset, size))
        print("%d variables in %d shard(s)" % (len(entries), shards))
        return
    if args.to_tf:
        data = np.load(args.prefix + ".npz")
        variables = {k: data[k] for k in data.files if not k.startswith("__")}
        T.write_checkpoint(args.prefix, variables)
        print("wrote %s.index and %s (%d variables)" % (args.prefix, os.path.basename(T._shard_name(args.prefix, 0, 1)), len(variables)))
        return
    variables = T.read_checkpoint(args.prefix, verify=args.verify)
    kept = {k: v for k, v in variables.items() if args.keep_slots or T.is_model_variable(k)}
    np.savez(args.prefix + ".npz", **kept)
    print("wrote %s.npz (%d variables, %d optimiser slots dropped)" % (args.prefix, len(kept), len(variables) - len(kept)))


if __name__ == "__main__":
    main()
