"""Trainer with the reference's public surface (model/trainer.py:17-730):

    Trainer(params, model_dir, single_cpu=False)
    .build(mode in {"train","valid","predict"}, dim, loss_type=None, num_speakers=None, noupdate_var_list=None)
    .train(data, spklist, learning_rate, aux_data=None)       one epoch, resumes from the latest checkpoint
    .valid(data, spklist, batch_type="softmax", output_embeddings=False, aux_data=None) -> (loss, emb, labels)
    .predict(features [T,D] | [B,T,D]) -> [E] | [B,E]
    .save(step) / .load() -> step / .reset() / .close()
    attributes callers touch: .sess, .embeddings, .endpoints, .params, .model

The TF1 session/graph is replaced by the native MI355X engine (csrc/xv_engine.hip) reached through
the C-ABI; this file is host logic only (loader, loop, logging, checkpoints, data parallelism).
Checkpoints: <model>/nnet/model-<step>.npz (variables by TF name + optimiser state) plus the
TF-format text index <model>/nnet/checkpoint that run_extract_embeddings.sh tests for.
"""
import logging
import os
import re
import sys
import time
from collections import OrderedDict

import numpy as np
import torch

try:
    from .. import engine as E
    from .. import _lib
    from ..parallel import GradAllReduce, average_bn_statistics, broadcast_variables
    from ..dataset.data_loader import KaldiDataRandomQueue, KaldiDataSeqQueue, PlannedRandomQueue, DataOutOfRange
    from ..misc.utils import substring_in_list, read_checkpoint_state, write_checkpoint_state, plan_length_batches
    from ..dataset.kaldi_io import PackedMatrix
    from ..misc import tf_checkpoint
    from .tdnn import tdnn, extended_tdnn, engine_config, collect_endpoints, check_params
    from . import loss as _loss
except (ImportError, ValueError):      # drop-in layout: PYTHONPATH=$TF_KALDI_ROOT
    import engine as E
    import _lib
    from parallel import GradAllReduce, average_bn_statistics, broadcast_variables
    from dataset.data_loader import KaldiDataRandomQueue, KaldiDataSeqQueue, PlannedRandomQueue, DataOutOfRange
    from misc.utils import substring_in_list, read_checkpoint_state, write_checkpoint_state, plan_length_batches
    from dataset.kaldi_io import PackedMatrix
    from misc import tf_checkpoint
    from model.tdnn import tdnn, extended_tdnn, engine_config, collect_endpoints, check_params
    from model import loss as _loss

log = logging.getLogger("tf_kaldi_speaker_amd")


class _PendingEmbeddings(object):
    """Embeddings of Trainer.predict_batch(..., return_device=True): still on the GPU, in batch order; cpu().numpy() reads them back and
    puts them into the caller's order."""

    def __init__(self, dev, inv):
        # the read-back is enqueued NOW, behind this window's forward passes, into pinned memory, and an event marks its end: numpy()
        # then waits for that event only.  (A dev.cpu() at read time queues behind whatever the caller has enqueued since - the next
        # window's decode and forward passes in extract.py - so host post-processing and GPU work took turns)
        self.inv = inv
        self.host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
        self.host.copy_(dev, non_blocking=True)
        self.done = torch.cuda.Event()
        self.done.record(torch.cuda.current_stream(dev.device))
        self.dev = dev          # (kept alive until the copy has run)

    def cpu(self):
        return self

    def numpy(self):
        self.done.synchronize()
        self.dev = None
        return self.host.numpy()[self.inv]

SOFTMAX_FAMILY = ("softmax", "asoftmax", "additive_margin_softmax", "additive_angular_margin_softmax")
OTHER_LOSSES = ("semihard_triplet_loss", "angular_triplet_loss", "generalized_angular_triplet_loss")


class _Session(object):
    """Stand-in for the tf.Session attribute some callers close explicitly."""

    def __init__(self, owner):
        self._owner = owner

    def close(self):
        self._owner._close_engine()


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist
    except Exception:
        pass
    return None


class Trainer(object):
    def __init__(self, params, model_dir, single_cpu=False):
        self.network_type = params.network_type
        if params.network_type == "tdnn":
            self.network = tdnn
        elif params.network_type == "extended_tdnn":      # tdnn's layer recipe on params.tdnn_layers (no reference counterpart)
            self.network = extended_tdnn
        else:
            raise NotImplementedError("Not implement %s network" % params.network_type)
        self.loss_type = None
        self.loss_network = None
        self.params = params
        self.single_cpu = single_cpu       # accepted for signature parity; the engine always runs on the GPU
        self.model = os.path.join(model_dir, "nnet")
        self.sess = _Session(self)
        self.engine = None
        self.dim = None
        self.num_speakers = None
        self.global_step = None
        self.learning_rate = None
        self.embeddings = None
        self.endpoints = None
        self.optimizer = None
        self.total_loss = None
        self.train_op = None
        self.train_ops = {}
        self.valid_ops = {}
        self.saver = None
        self.is_built = False
        self.is_loaded = False
        self.modes = set()
        self.device = "cuda:%d" % _lib.local_device_index()

    # ------------------------------------------------------------------ lifecycle
    def _close_engine(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None

    def reset(self):
        """Drop the graph so a new one can be built (reference trainer.py:114-133)."""
        self._close_engine()
        self.is_built = False
        self.is_loaded = False
        self.modes = set()
        self.saver = None

    def close(self):
        self._close_engine()

    def _capacity(self):
        p = self.params.dict
        b = int(p.get("num_speakers_per_batch", 64)) * int(p.get("num_segments_per_speaker", 1))
        t = int(p.get("max_segment_len", 400))
        return max(b, 1), max(t, 15)

    def _make_engine(self, num_speakers, loss_type, max_batch, max_frames, keep=None, max_rows=0):
        cfg = engine_config(self.params, self.dim, num_speakers or 0, loss_type or "softmax", max_batch, max_frames, max_rows)
        eng = E.Engine(cfg, device=self.device)
        seed = int(self.params.dict.get("seed", 0))
        eng.init_variables(seed=seed)
        if keep:
            eng.set_variables({k: v for k, v in keep.items() if k in eng.table})
        return eng

    def _ensure_capacity(self, b, t, rows=0):
        """Grow the engine to hold b chunks of t frames.  rows > 0 (batched extraction): a capacity in rows = chunks x frames instead
        of the full b x t rectangle - many short utterances or a few long ones per batch."""
        eng = self.engine
        cfg = eng.config
        have_rows = cfg.max_rows if cfg.max_rows > 0 else cfg.max_batch * cfg.max_frames
        need_rows = rows if rows > 0 else b * t      # (rows given: b and t are the largest batch and the longest utterance of DIFFERENT batches)
        if b <= cfg.max_batch and t <= cfg.max_frames and need_rows <= have_rows:
            return
        values = eng.get_variables()
        opt, cnt = eng.opt_state.clone(), eng.update_count
        nb, nt = max(b, cfg.max_batch), max(t, cfg.max_frames)
        if t > cfg.max_frames:
            nt = max(t, min(2 * cfg.max_frames, 20000))
        nrows = 0
        if rows > 0 or cfg.max_rows > 0:
            # row-bounded capacity: chunks and frames per chunk cost nothing beyond the rows, so take them generously once instead of
            # rebuilding the engine whenever a later window holds a slightly longer utterance or a slightly larger batch
            nrows = max(need_rows, cfg.max_rows)
            nb, nt = max(nb, self.PREDICT_CHUNKS), max(nt, 20000)
        self._close_engine()
        self.engine = self._make_engine(self.num_speakers, self.loss_type, nb, nt, keep=values, max_rows=nrows)
        if self.engine.opt_state.numel() == opt.numel():
            self.engine.opt_state.copy_(opt)
        self._apply_update_filter()
        self.engine.update_count = cnt

    # ------------------------------------------------------------------ build
    def build(self, mode, dim, loss_type=None, num_speakers=None, noupdate_var_list=None):
        assert (mode == "train" or mode == "valid" or mode == "predict")
        check_params(self.params)
        self.dim = int(dim)
        if mode == "predict":
            if "embedding_node" not in self.params.dict:
                raise KeyError("embedding_node")
            log.info("Extract embedding from node %s" % self.params.embedding_node)
            if self.engine is None:
                self.engine = self._make_engine(0, "softmax", 1, 400)
            self.embeddings = self.params.embedding_node
            self.modes.add(mode)
            self.is_built = True
            return

        self.params.dict["global_step"] = 0
        self.loss_type = loss_type
        if loss_type in SOFTMAX_FAMILY:
            self.loss_network = getattr(_loss, loss_type)
        elif loss_type in OTHER_LOSSES:
            raise NotImplementedError("Not implement %s loss on the MI355X engine (outside the hot path)" % loss_type)
        else:
            raise NotImplementedError("Not implement %s loss" % self.loss_type)
        if mode == "train":
            if "optimizer" not in self.params.dict:
                self.params.dict["optimizer"] = "sgd"
            if self.params.optimizer == "sgd":
                if "momentum" in self.params.dict:
                    sys.exit("Using sgd as the optimizer and you should not specify the momentum.")
            elif self.params.optimizer not in ("momentum", "adam"):
                sys.exit("Optimizer %s is not supported." % self.params.optimizer)
            self.optimizer = self.params.optimizer
        self.num_speakers = int(num_speakers)
        b, t = self._capacity()
        if self.engine is None:
            self.engine = self._make_engine(self.num_speakers, loss_type, b, t)
        elif self.engine.config.num_speakers != self.num_speakers:
            keep = self.engine.get_variables()
            self._close_engine()
            self.engine = self._make_engine(self.num_speakers, loss_type, b, t, keep=keep)
        self.embeddings = "output"          # valid embeddings = endpoints["output"], trainer.py:308
        self.modes.add(mode)
        self.is_built = True
        if mode == "train" and noupdate_var_list is not None:
            # fine-tuning (trainer.py:379-403): variables - and BN UPDATE_OPS - whose name contains one of the strings are left alone
            self._frozen = [n for n in self.engine.table if substring_in_list(n, noupdate_var_list)]
            for n in self.engine.table:
                log.info("[Info] Var %s will not be updated" % n if n in self._frozen else "[Info] Train %s" % n)
        self._apply_update_filter()

    def _apply_update_filter(self):
        if self.engine is not None:
            self.engine.set_update_filter(getattr(self, "_frozen", None) or ())

    # ------------------------------------------------------------------ checkpoints
    def save(self, step, sync_bn=True):
        """sync_bn: average the BN moving statistics over the ranks first - a COLLECTIVE, so every rank must make this call
        (the epoch loop does); sync_bn=False is for a caller that runs on one rank only (get_finetune_model on rank 0)."""
        dist = _dist()
        if dist is not None:
            if sync_bn:
                average_bn_statistics(dist, self.engine.variables, self.engine.n_train, dist.get_world_size())
            if dist.get_rank() != 0:
                return
        os.makedirs(self.model, exist_ok=True)
        path = os.path.join(self.model, "model-%d" % step)
        arrays = dict(self.engine.get_variables())
        arrays["__opt_state__"] = self.engine.opt_state.cpu().numpy()
        arrays["__update_count__"] = np.array([self.engine.update_count], np.int64)
        tmp = path + ".tmp.npz"
        np.savez(tmp, **arrays)
        os.replace(tmp, path + ".npz")
        if self.params.dict.get("save_tf_checkpoint", False):
            # the same variables as a TensorFlow V2 checkpoint (model-<step>.index / .data-00000-of-00001) the reference's
            # tf.train.Saver restores (trainer.py:142-158): an upstream user can pick the model up without this package
            tf_checkpoint.write_checkpoint(path, self.engine.get_variables())
        # the index keeps BASENAMES (TF's save_relative_paths form): the model directory can be moved or copied and pruning
        # / loading still find the payloads; entries written with a directory part by older runs are read by basename
        name = os.path.basename(path)
        _, listed = read_checkpoint_state(self.model)
        names = [n for n in (os.path.basename(q) for q in listed) if n != name] + [name]
        keep = int(self.params.dict.get("keep_checkpoint_max", 5))
        while keep > 0 and len(names) > keep:
            old = os.path.join(self.model, names.pop(0))
            for ext in (".npz", ".index", ".data-00000-of-00001"):
                if os.path.exists(old + ext):
                    os.remove(old + ext)
        write_checkpoint_state(self.model, name, names)

    def load(self):
        log.info("Reading checkpoints...")
        current, _ = read_checkpoint_state(self.model)
        if not current:
            sys.exit("Failed to find a checkpoint in {}".format(self.model))
        ckpt_name = os.path.basename(current)
        step = int(next(re.finditer(r"(\d+)(?!.*\d)", ckpt_name)).group(0))
        path = os.path.join(self.model, ckpt_name + ".npz")
        if not os.path.isfile(path):
            prefix = os.path.join(self.model, ckpt_name)
            if os.path.isfile(prefix + ".index"):
                # a checkpoint written by the reference's tf.train.Saver (trainer.py:318,444; the pretrained models of README.md:86-104):
                # same variable names and shapes, read without TensorFlow (misc/tf_checkpoint.py).  Optimiser slots are not taken over.
                tf_vars = tf_checkpoint.read_checkpoint(prefix)
                values = {k: v for k, v in tf_vars.items() if k in self.engine.table}
                missing = [k for k in self.engine.table if k not in values]
                if missing:
                    sys.exit("Checkpoint %s.index lacks variables: %s" % (prefix, ", ".join(missing[:5])))
                self.engine.set_variables(values)
                log.info("Succeed to load TensorFlow checkpoint {}".format(ckpt_name))
                self.is_loaded = True
                return step
            sys.exit("Failed to find a checkpoint in {}".format(self.model))
        data = np.load(path)
        values = {k: data[k] for k in data.files if not k.startswith("__") and k in self.engine.table}
        missing = [k for k in self.engine.table if k not in values]
        if missing:
            sys.exit("Checkpoint %s lacks variables: %s" % (path, ", ".join(missing[:5])))
        self.engine.set_variables(values)
        if "__opt_state__" in data.files and data["__opt_state__"].size == self.engine.opt_state.numel():
            self.engine.opt_state.copy_(torch.from_numpy(data["__opt_state__"]))
        if "__update_count__" in data.files:
            self.engine.update_count = int(data["__update_count__"][0])
        log.info("Succeed to load checkpoint {}".format(ckpt_name))
        self.is_loaded = True
        return step

    # ------------------------------------------------------------------ training
    def train(self, data, spklist, learning_rate, aux_data=None):
        """One epoch over `num_steps_per_epoch` random batches (reference trainer.py:451-520)."""
        assert "train" in self.modes, "call build('train', ...) first"
        p = self.params
        curr_step = 0
        if os.path.isfile(os.path.join(self.model, "checkpoint")):
            curr_step = self.load()
        dist = _dist()
        if dist is not None:
            broadcast_variables(dist, self.engine.variables, 0)
            self.engine.lib.xv_engine_invalidate_weights(self.engine.h)
        # KaldiDataRandomQueue = the C++ loader (libxvector_io.so: planning + decoding in native threads); XV_LOADER=python
        # plans the batches in Python and only decodes natively (dataset/data_loader.py) - same rules, same batch contract
        # XV_LOADER=gpu_decode: the native threads only gather the rows' 'CM ' bytes, the batch is decoded on the GPU (xv_cm_decode) -
        # a quarter of the host memory traffic and PCIe bytes, bit-identical features ('CM ' archives only)
        which = os.environ.get("XV_LOADER", "native")
        queue_cls = PlannedRandomQueue if which == "python" else KaldiDataRandomQueue
        extra = {"packed": True} if which == "gpu_decode" else {}
        loader = queue_cls(data, spklist, num_parallel=p.num_parallel_datasets, max_qsize=p.max_queue_size,
                           num_speakers=p.num_speakers_per_batch, num_segments=p.num_segments_per_speaker,
                           min_len=p.min_segment_len, max_len=p.max_segment_len, shuffle=True, **extra)
        loader.start()
        try:
            if hasattr(loader, "device_batches"):        # pinned staging + asynchronous H2D on a copy stream
                batches = loader.device_batches(self.device)
            else:
                batches = iter(loader.fetch, None)
            curr_step = self.train_batches(batches, learning_rate, curr_step)
        finally:
            loader.stop()
        self.save(curr_step)
        return

    def train_batches(self, batches, learning_rate, curr_step=0, num_steps=None):
        """The hot loop on an iterator of (features [B,T,D], labels [B]); returns the new global step.
        curr_step (the count of optimiser steps incl. resumed ones) feeds the lambda schedule."""
        p = self.params
        dist = _dist()
        allreduce = GradAllReduce(dist, dist.get_world_size()) if dist is not None else None
        rank0 = dist is None or dist.get_rank() == 0
        steps_per_epoch = int(p.num_steps_per_epoch) if num_steps is None else int(num_steps)
        epoch = int(curr_step / p.num_steps_per_epoch) if num_steps is None else 0
        first = curr_step % steps_per_epoch if num_steps is None else 0
        show = int(p.dict.get("show_training_progress", 100))
        save_every = int(p.dict.get("save_checkpoints_steps", 0) or 0)
        # data parallelism: BN moving statistics follow each rank's own batches (local batch statistics);
        # they are re-averaged every `bn_sync_steps` optimiser steps (one all-reduce of a few thousand floats) and before every
        # checkpoint, so replicas never drift further apart than that many moving-average updates
        bn_sync = int(p.dict.get("bn_sync_steps", 100) or 0) if dist is not None else 0
        for step in range(first, steps_per_epoch):
            try:
                start_time = time.time()
                features, labels = next(batches)
            except (DataOutOfRange, StopIteration):
                log.info("Finished reading features.")
                break
            self._ensure_capacity(features.shape[0], features.shape[1])
            verbose = show > 0 and step % show == 0
            losses = self.engine.train_step(features, labels, learning_rate, curr_step, allreduce=allreduce,
                                            fetch_losses=verbose)
            if verbose and rank0:
                raw, reg = losses
                self.train_ops = {"raw_loss": raw, "loss": raw + reg}
                log.info("Epoch: [%2d] step: [%2d/%2d] time: %.4f s/step, raw loss: %f, total loss: %f"
                         % (epoch, step, steps_per_epoch, time.time() - start_time, raw, raw + reg))
            if save_every > 0 and step % save_every == 0 and curr_step != 0:
                self.save(curr_step)
            elif bn_sync > 0 and (curr_step + 1) % bn_sync == 0:
                average_bn_statistics(dist, self.engine.variables, self.engine.n_train, dist.get_world_size())
            curr_step += 1
        return curr_step

    # ------------------------------------------------------------------ validation / inference
    def valid(self, data, spklist, batch_type="softmax", output_embeddings=False, aux_data=None):
        """Mean validation loss with the margin switched off (trainer.py:261-271) and BN in inference
        mode; optionally all "output" embeddings + labels in file order (trainer.py:592-706)."""
        assert "valid" in self.modes, "call build('valid', ...) first"
        assert batch_type == "softmax" or batch_type == "end2end", "The batch_type can only be softmax or end2end"
        if batch_type == "end2end":
            raise NotImplementedError("end2end validation batches belong to the triplet/GE2E losses (outside the hot path)")
        p = self.params
        curr_step = 0
        if os.path.isfile(os.path.join(self.model, "checkpoint")):
            curr_step = self.load()
        else:
            log.info("[Warning] Cannot find model in %s. Random initialization is used in validation." % self.model)
        bsz = p.num_speakers_per_batch * p.num_segments_per_speaker
        embeddings_val, labels_val = None, None
        if output_embeddings:
            loader = KaldiDataSeqQueue(data, spklist, num_parallel=2, max_qsize=10, batch_size=bsz,
                                       min_len=p.min_segment_len, max_len=p.max_segment_len, shuffle=False)
            loader.start()
            embs, labs = [], []
            try:
                while True:
                    try:
                        features, labels = loader.fetch()
                    except DataOutOfRange:
                        break
                    self._ensure_capacity(features.shape[0], features.shape[1])
                    self.engine.forward(features, False)
                    embs.append(self.engine.endpoint("output").cpu().numpy())
                    labs.append(np.asarray(labels))
            finally:
                loader.stop()
            if embs:
                embeddings_val, labels_val = np.concatenate(embs, 0), np.concatenate(labs, 0)
        loader = KaldiDataSeqQueue(data, spklist, num_parallel=2, max_qsize=10, batch_size=bsz,
                                   min_len=p.min_segment_len, max_len=p.max_segment_len, shuffle=True)
        loader.start()
        total, num_batches = 0.0, 0
        try:
            for _ in range(int(p.valid_max_iterations)):
                try:
                    features, labels = loader.fetch()
                except DataOutOfRange:
                    break
                self._ensure_capacity(features.shape[0], features.shape[1])
                self.engine.forward(features, False)
                self.engine.loss(labels, curr_step, with_margin=False)
                self._last_valid_labels = np.asarray(labels)
                total += self.engine.raw_loss()
                num_batches += 1
        finally:
            loader.stop()
        loss = total / max(num_batches, 1)
        log.info("[Validation %d batches] valid loss: %f" % (num_batches, loss))
        return loss, embeddings_val, labels_val

    def predict(self, features):
        """Embedding(s) of `embedding_node` with BN in inference mode (trainer.py:708-726)."""
        if not self.is_loaded:
            if os.path.isfile(os.path.join(self.model, "checkpoint")):
                self.load()
            else:
                sys.exit("Cannot find model in %s" % self.model)
        features = np.asarray(features, np.float32)
        rank = len(features.shape)
        assert (rank == 2 or rank == 3)
        if rank == 2:
            features = np.expand_dims(features, axis=0)
        self._ensure_capacity(features.shape[0], features.shape[1])
        self.engine.forward(features, False)
        node = self.params.embedding_node
        emb = self.engine.endpoint(node)
        b = features.shape[0]
        if emb.shape[0] != b:
            emb = emb.view(b, emb.shape[0] // b, emb.shape[1])
        emb = emb.cpu().numpy()
        self.endpoints = OrderedDict([(node, emb)])
        if rank == 2:
            emb = np.squeeze(emb, axis=0)
        return emb

    # rows (chunks x frames) of one batched-extraction forward: about two S1 training batches - large enough for the GEMMs to run at
    # their training-pass rate, small enough that the activation arena stays at a few GB
    PREDICT_ROWS = 49152
    PREDICT_CHUNKS = 128

    def predict_batch(self, items, return_device=False):
        """Embeddings of MANY utterances of different lengths, [n, E] in the order given - the batched form of the loop
        extract.py:64-93 runs one predict() at a time.  items: float32 matrices [T_i, D] and / or kaldi_io.PackedMatrix ('CM '
        matrices as read from the archive, decoded on the GPU).  Utterances are sorted by length, padded to the longest of their
        batch (misc.utils.plan_length_batches) and run through xv_engine_forward_lengths, whose pooling sees only each utterance's
        own frames; the embedding node must be a segment-level one (tdnn6_dense, tdnn7_*, output)."""
        if not self.is_loaded:
            if os.path.isfile(os.path.join(self.model, "checkpoint")):
                self.load()
            else:
                sys.exit("Cannot find model in %s" % self.model)
        n = len(items)
        if n == 0:
            return np.zeros((0, 0), np.float32)
        lengths = [int(it.shape[0]) for it in items]
        dim = self.dim
        # every matrix must have the model's feature dimension: the GPU decoder takes the column stride from `dim`, not from the archive
        # header, so a mismatch would decode garbage (and read past the record) instead of failing; predict() rejects it through
        # Engine.forward, this is the same check for the batched path
        for i, it in enumerate(items):
            if len(it.shape) != 2 or int(it.shape[1]) != dim:
                raise ValueError("predict_batch: item %d has shape %s, the model expects [frames, %d]" % (i, tuple(it.shape), dim))
        # an utterance shorter than the network's receptive field has no valid output frame (predict() refuses it in Engine.forward; here the
        # pooling would clamp to one frame that already sees zero padding); longer than max_frames cannot happen: the plan pads to the longest
        min_frames = self.engine.min_frames
        for i, t in enumerate(lengths):
            if t < min_frames:
                raise ValueError("predict_batch: item %d has %d frames, fewer than the network's receptive field (%d)" % (i, t, min_frames))
        # the on-GPU 'CM ' decoder handles at most 128 feature dimensions (one lane per column header); wider archives go through the host codec
        if dim > 128:
            items = [it.decode() if isinstance(it, PackedMatrix) else it for it in items]
        node = self.params.embedding_node
        plan = plan_length_batches(lengths, self.PREDICT_ROWS, self.PREDICT_CHUNKS)
        self._ensure_capacity(max(len(idx) for idx, _ in plan), max(t for _, t in plan), rows=max(self.PREDICT_ROWS, max(len(idx) * t for idx, t in plan)))
        eng = self.engine
        outs = []
        # 'CM ' matrices stay where the reader cut them out: the archive blocks they are views of go to the GPU whole, once per call (a
        # quarter of the fp32 bytes, no per-batch gather on the host); a batch is then a list of byte offsets for the on-GPU decode
        base, staged, total = {}, [], 0
        for it in items:
            if isinstance(it, PackedMatrix) and id(it.block) not in base:
                base[id(it.block)] = total
                staged.append(it.block)
                total += (len(it.block) + 15) // 16 * 16
        packed_dev = None
        if total:
            host = torch.empty(total, dtype=torch.uint8, pin_memory=True)
            hv = host.numpy()
            for blk in staged:
                o = base[id(blk)]
                hv[o:o + len(blk)] = np.frombuffer(blk, np.uint8)
            packed_dev = host.to(eng.device, non_blocking=True)
        # byte offsets and frame counts of every batch in ONE pinned upload as well: a pageable host -> device copy per batch makes the
        # host wait for the stream (the previous batch's forward), i.e. host and GPU work would take turns instead of overlapping
        flat = [i for idx, _ in plan for i in idx]
        meta = torch.empty((2, len(flat)), dtype=torch.int64, pin_memory=True)
        mv = meta.numpy()
        mv[0] = [base[id(items[i].block)] + items[i].start if isinstance(items[i], PackedMatrix) else 0 for i in flat]
        mv[1] = [lengths[i] for i in flat]
        meta_dev = meta.to(eng.device, non_blocking=True)
        rows_dev = meta_dev[1].to(torch.int32)
        cursor = 0
        for idx, t in plan:
            b = len(idx)
            sl = slice(cursor, cursor + b)
            cursor += b
            if all(isinstance(items[i], PackedMatrix) for i in idx):
                x, rows = eng.decode_packed(packed_dev, meta_dev[0, sl], rows_dev[sl], t)
            else:
                host = np.zeros((b, t, dim), np.float32)
                for j, i in enumerate(idx):
                    m = items[i].decode() if isinstance(items[i], PackedMatrix) else np.asarray(items[i], np.float32)
                    host[j, :lengths[i]] = m
                x, rows = host, rows_dev[sl]
            eng.forward_lengths(x, rows)
            emb = eng.endpoint(node)
            if emb.shape[0] != b:
                raise ValueError("predict_batch: embedding node %s is a frame-level endpoint (%d rows for %d utterances)" % (node, emb.shape[0], b))
            outs.append((idx, emb))
        dev = torch.cat([e for _, e in outs], dim=0)
        order = np.concatenate([np.asarray(idx) for idx, _ in outs])
        inv = np.empty(n, np.int64)
        inv[order] = np.arange(n)
        if return_device:
            # nothing here waits for the GPU (the permutation back to the caller's order is applied on the host after the read-back: an
            # index tensor uploaded from pageable memory would have made this call wait for the whole window's forward passes)
            return _PendingEmbeddings(dev, inv)
        emb = dev.cpu().numpy()[inv]
        self.endpoints = OrderedDict([(node, emb)])
        return emb

    # ------------------------------------------------------------------ fine-tuning / diagnostics (trainer.py:522-590, 728-920)
    def set_trainable_variables(self, variable_list=None):
        """Only the trainable variables whose name contains one of the strings are optimised (None: all).  BN moving statistics
        keep updating (trainer.py:728-773)."""
        assert "train" in self.modes, "call build('train', ...) first"
        if variable_list is None:
            log.info("[Info] Add all trainable variables to the optimizer.")
            self._frozen = []
        else:
            self._frozen = []
            for n, (_, _, trainable) in self.engine.table.items():
                if not trainable:
                    continue
                if substring_in_list(n, variable_list):
                    log.info("[Info] Add %s to trainable list" % n)
                else:
                    self._frozen.append(n)
        self._apply_update_filter()

    def get_finetune_model(self, excluded_list):
        """Start from the pre-trained checkpoint in the model directory; variables whose name contains a string of `excluded_list`
        keep their default initialisation.  The pre-trained files are backed up (`.bak`) and the result is saved as step 0
        (trainer.py:775-819)."""
        assert self.engine is not None, "call build(...) first"
        self.engine.init_variables(seed=int(self.params.dict.get("seed", 0)))
        fresh = self.engine.get_variables()
        current, _ = read_checkpoint_state(self.model)
        if not current:
            sys.exit("Failed to find a checkpoint in {}".format(self.model))
        path = os.path.join(self.model, os.path.basename(current) + ".npz")
        data = np.load(path)
        values = {}
        for name in self.engine.table:
            if substring_in_list(name, excluded_list):
                log.info("[Info] Ignore %s when loading the checkpoint" % name)
                values[name] = fresh[name]
            elif name in data.files and tuple(data[name].shape) == tuple(fresh[name].shape):
                values[name] = data[name]
            else:
                sys.exit("Checkpoint %s lacks variable %s (exclude it to re-initialise it)" % (path, name))
        self.engine.set_variables(values)
        self.engine.opt_state.zero_()
        self.engine.update_count = 0
        import glob
        import shutil
        for filename in glob.glob(os.path.join(self.model, os.path.basename(current)) + "*"):
            if not filename.endswith(".bak"):
                shutil.copyfile(filename, filename + ".bak")
        self.save(0, sync_bn=False)      # no collective: nnet/lib/train.py calls this on rank 0 only (replicas are identical here)
        self.is_loaded = True

    def train_tune_lr(self, data, spklist, tune_period=100, aux_data=None, tune_times=None):
        """Learning-rate range test (trainer.py:522-590): lr = 1e-5 * 1.15^(step // tune_period), global_step fed as 0, the
        (step, lr, total loss) of the first step of every period written to <model>/learning_rate_tuning."""
        assert "train" in self.modes, "call build('train', ...) first"
        p = self.params
        if tune_times is None:      # the reference hard-codes 100 learning rates (trainer.py:556); XV_TUNE_TIMES shortens a smoke run
            tune_times = int(os.environ.get("XV_TUNE_TIMES", "100"))
        self.engine.init_variables(seed=int(p.dict.get("seed", 0)))
        if os.path.isfile(os.path.join(self.model, "checkpoint")):
            self.load()
        queue_cls = PlannedRandomQueue if os.environ.get("XV_LOADER", "native") == "python" else KaldiDataRandomQueue
        loader = queue_cls(data, spklist, num_parallel=p.num_parallel_datasets, max_qsize=p.max_queue_size,
                           num_speakers=p.num_speakers_per_batch, num_segments=p.num_segments_per_speaker,
                           min_len=p.min_segment_len, max_len=p.max_segment_len, shuffle=True)
        loader.start()
        init_learning_rate, factor = 1e-5, 1.15
        os.makedirs(self.model, exist_ok=True)
        try:
            with open(os.path.join(self.model, "learning_rate_tuning"), "w") as fp_lr:
                for step in range(int(tune_period) * int(tune_times)):
                    lr = init_learning_rate * (factor ** (step // tune_period))
                    start_time = time.time()
                    features, labels = loader.fetch()
                    self._ensure_capacity(features.shape[0], features.shape[1])
                    first = step % tune_period == 0
                    losses = self.engine.train_step(features, labels, lr, 0, fetch_losses=first)
                    if first:
                        raw, reg = losses
                        log.info("Epoch: step: %2d, time: %.4f s/step, lr: %f, raw loss: %f, total loss: %f"
                                 % (step, time.time() - start_time, lr, raw, raw + reg))
                        fp_lr.write("%d %f %f\n" % (step, lr, raw + reg))
        finally:
            loader.stop()

    def insight(self, data, spklist, batch_type="softmax", output_embeddings=False, aux_data=None):
        """The reference's debugging pass (trainer.py:821-920) without its pdb breakpoint: validation loss, optional embeddings /
        labels in file order, and the accuracy of the last validation batch in the log."""
        loss, emb, labels = self.valid(data, spklist, batch_type=batch_type, output_embeddings=output_embeddings, aux_data=aux_data)
        try:
            logits = self.engine.endpoint("logits").cpu().numpy()
            lab = self._last_valid_labels
            log.info("Acc: %f" % (float(np.sum(np.argmax(logits[:, :self.num_speakers], axis=1) == lab)) / float(lab.shape[0])))
        except Exception:
            pass
        return loss, emb, labels
