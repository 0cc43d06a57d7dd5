#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/extract.py (same CLI, wrap/extract_wrapper.sh:39-40):

    python nnet/lib/extract.py [-g GPU] [-m MIN] [-s CHUNK] [-n] [--node NAME] model_dir rspecifier wspecifier

Reads an ark / input pipe of feature matrices, writes a Kaldi float-vector ark of embeddings
(is_training=False graph, moving BN statistics).  Utterances longer than --chunk-size are cut into
half-overlapping chunks whose embeddings are length-weighted averaged (reference extract.py:65-94).
-g selects the HIP device (the reference's CPU mode `-g -1` maps to device 0: there is no CPU path).
"""
import os
import sys

import _cli
from model.trainer import Trainer
from misc.utils import Params, utterance_embedding, prefetch_iter
from dataset.kaldi_io import open_or_fd, read_mat_ark, write_vec_flt


def main():
    log = _cli.logger()
    args = _cli.parser_for("gpu", "min_chunk_size", "chunk_size", "normalize", "node", "model_dir", "rspecifier", "wspecifier").parse_args()
    # -g is an enable flag in the reference ("an arbitrary number except -1"; run_extract_embeddings.sh passes the JOB
    # number with --gpuid), device choice being left to CUDA_VISIBLE_DEVICES.  Here: among the devices HIP_VISIBLE_DEVICES
    # leaves visible, job N takes device N modulo their count, so `nj` parallel jobs spread over the GPUs of the node and
    # a node with one device serves every job; -1 (the default) = the first visible device.
    if args.gpu >= 0:
        import torch
        os.environ["LOCAL_RANK"] = str(args.gpu % max(torch.cuda.device_count(), 1))
    nnet_dir = os.path.join(args.model_dir, "nnet")
    config_json = os.path.join(args.model_dir, "nnet/config.json")
    if not os.path.isfile(config_json):
        sys.exit("Cannot find params.json in %s" % config_json)
    params = Params(config_json)
    if len(args.node) != 0:
        params.embedding_node = args.node
    log.info("Extract embedding from %s" % params.embedding_node)
    trainer = Trainer(params, args.model_dir, single_cpu=True)
    with open(os.path.join(nnet_dir, "feature_dim"), "r") as f:
        dim = int(f.readline().strip())
    trainer.build("predict", dim=dim)
    if "." in args.rspecifier and args.rspecifier.rsplit(".", 1)[1] == "scp":
        sys.exit("The rspecifier must be ark or input pipe")
    fp_out = open_or_fd(args.wspecifier, "wb")
    for key, feature in prefetch_iter(read_mat_ark(args.rspecifier)):       # the reader runs ahead of the GPU in its own thread
        frames = feature.shape[0]
        if frames < args.min_chunk_size:
            log.info("[INFO] Key %s length too short, %d < %d, skip." % (key, frames, args.min_chunk_size))
            continue
        embedding, pieces = utterance_embedding(trainer.predict, feature, args.chunk_size, args.normalize)
        log.info("[INFO] Key %s length %d%s." % (key, frames, "" if pieces == 1 else " > %d, split to %d segments" % (args.chunk_size, pieces)))
        write_vec_flt(fp_out, embedding, key=key)
    fp_out.close()
    trainer.close()


if __name__ == "__main__":
    main()
