#!/usr/bin/env python3
"""Drop-in for egs/voxceleb/v1/nnet/lib/extract.py (same CLI, wrap/extract_wrapper.sh:39-40):

    python nnet/lib/extract.py [-g GPU] [-m MIN] [-s CHUNK] [-n] [--node NAME] model_dir rspecifier wspecifier

Reads an ark / input pipe of feature matrices, writes a Kaldi float-vector ark of embeddings
(is_training=False graph, moving BN statistics).  Utterances longer than --chunk-size are cut into
half-overlapping chunks whose embeddings are length-weighted averaged (reference extract.py:65-94).
-g selects the HIP device (the reference's CPU mode `-g -1` maps to device 0: there is no CPU path).
"""
import logging
import os
import sys
import time

import _cli
from model.trainer import Trainer
from misc.utils import Params, EmbeddingWindow, prefetch_iter
from dataset.kaldi_io import open_or_fd, read_mat_ark_packed, write_vec_flt

# utterances gathered before they go to the GPU together (sorted by length into padded batches, Trainer.predict_batch); results are
# written in archive order.  One utterance per forward, as the reference runs (extract.py:64-93), left the chip at 9-20 % of its
# training-pass rate at real utterance lengths (profiles/r03_extract_bench.txt).
WINDOW_UTTERANCES = 512
WINDOW_FRAMES = 400000


def main():
    log = _cli.logger(batched=True)
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
    window, window_frames = [], 0
    pending = []           # the window whose forward passes are on the GPU while the next one is read, planned and enqueued
    stats = {"utts": 0, "frames": 0, "t0": time.time(), "warm": None}      # warm: (time, utterances, frames) once the first window is done

    def submit():
        entries = list(window)
        del window[:]
        keep = [feature for _, feature in entries if feature.shape[0] >= args.min_chunk_size]
        job = EmbeddingWindow(lambda pieces: trainer.predict_batch(pieces, return_device=True), keep, args.chunk_size, args.normalize)
        finish()           # the previous window: read back, log, write - while this one runs
        pending.append((entries, job))

    def finish():
        if not pending:
            return
        entries, job = pending.pop()
        results = iter(job.results())
        for key, feature in entries:         # log lines and output vectors in archive order, as the one-at-a-time loop gives them
            frames = feature.shape[0]
            if frames < args.min_chunk_size:
                log.info("[INFO] Key %s length too short, %d < %d, skip." % (key, frames, args.min_chunk_size))
                continue
            embedding, pieces = next(results)
            log.info("[INFO] Key %s length %d%s." % (key, frames, "" if pieces == 1 else " > %d, split to %d segments" % (args.chunk_size, pieces)))
            write_vec_flt(fp_out, embedding, key=key)
            stats["utts"] += 1
            stats["frames"] += frames
        if stats["warm"] is None:
            stats["warm"] = (time.time(), stats["utts"], stats["frames"])
        for h in logging.getLogger().handlers:
            h.flush()

    # the reader runs ahead of the GPU in its own thread; 'CM ' matrices arrive undecoded (kaldi_io.PackedMatrix) and are decoded on the GPU
    for key, feature in prefetch_iter(read_mat_ark_packed(args.rspecifier), depth=2 * WINDOW_UTTERANCES):
        window.append((key, feature))
        window_frames += feature.shape[0]
        if len(window) >= WINDOW_UTTERANCES or window_frames >= WINDOW_FRAMES:
            submit()
            window_frames = 0
    submit()
    finish()
    now = time.time()
    t1, u1, f1 = stats["warm"] or (now, 0, 0)
    rate = ", %.0f utterances/s = %.2f M frames/s after the first window" % ((stats["utts"] - u1) / (now - t1), (stats["frames"] - f1) / (now - t1) / 1e6) \
        if stats["utts"] > u1 and now > t1 else ""
    log.info("[INFO] Extracted %d utterances (%d frames) in %.2f s%s." % (stats["utts"], stats["frames"], now - stats["t0"], rate))
    fp_out.close()
    trainer.close()


if __name__ == "__main__":
    main()
