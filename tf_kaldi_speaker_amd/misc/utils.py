"""Config / checkpoint-selection / EER helpers with the reference's misc/utils.py API
(Params :13, save_codes_and_config :64, load_lr :193, load_valid_loss :203, get_checkpoint :217,
compute_cos_pairwise_eer :273).  No TensorFlow: the checkpoint *index* file keeps TF's text format
(`model_checkpoint_path: "..."`), the payload is this repo's own .npz (model/trainer.py)."""
import json
import logging
import os
import re
import shutil
import sys

import numpy as np

log = logging.getLogger("tf_kaldi_speaker_amd")


class Params(object):
    """Hyper-parameters loaded from a JSON file; attribute and `.dict[...]` access.
    Unknown / comment keys ("Note": ...) are legal (tdnn_softmax_1e-2.json:2,13)."""

    def __init__(self, json_path):
        self.update(json_path)

    def save(self, json_path):
        with open(json_path, "w") as f:
            json.dump({k: v for k, v in self.__dict__.items() if _jsonable(v)}, f, indent=4)

    def update(self, json_path):
        with open(json_path) as f:
            self.__dict__.update(json.load(f))

    @property
    def dict(self):
        return self.__dict__


class ParamsPlain(object):
    def __init__(self):
        pass

    @property
    def dict(self):
        return self.__dict__


def _jsonable(v):
    try:
        json.dumps(v)
        return True
    except TypeError:
        return False


def save_codes_and_config(cont, model, config):
    """Snapshot code + config into <model>/{codes,lib,nnet} exactly like the reference (utils.py:64-123);
    with `cont` reload <model>/nnet/config.json instead."""
    if cont:
        if not os.path.isdir(os.path.join(model, "nnet")) or not os.path.isdir(os.path.join(model, "codes")):
            sys.exit("To continue training the model, nnet and codes must be existed in %s." % model)
        log.info("Continue training from %s." % model)
        return Params(os.path.join(model, "nnet/config.json"))
    if os.path.isdir(os.path.join(model, "nnet")):
        backup = os.path.join(model, ".backup")
        log.info("Save backup to %s" % backup)
        if os.path.isdir(backup):
            shutil.rmtree(backup)
        os.makedirs(backup)
        for d in ("codes", "nnet", "lib"):
            if os.path.exists(os.path.join(model, d)):
                shutil.move(os.path.join(model, d), backup + "/")
    for d in ("codes", "lib"):
        if os.path.isdir(os.path.join(model, d)):
            shutil.rmtree(os.path.join(model, d))
    os.makedirs(os.path.join(model, "codes"))
    root = os.environ.get("TF_KALDI_ROOT")
    if not root:
        log.error("TF_KALDI_ROOT should be set before training. Refer to path.sh to set the value manually. ")
        sys.exit(1)
    for d in ("dataset", "model", "misc"):
        shutil.copytree(os.path.join(root, d), os.path.join(model, "codes", d),
                        ignore=shutil.ignore_patterns("__pycache__", "*.pyc"))
    lib_src = os.path.join(os.getcwd(), "nnet/lib")
    if os.path.isdir(lib_src):
        shutil.copytree(lib_src, os.path.join(model, "lib"), ignore=shutil.ignore_patterns("__pycache__", "*.pyc"))
    if not os.path.isdir(os.path.join(model, "nnet")):
        os.makedirs(os.path.join(model, "nnet"))
    shutil.copyfile(config, os.path.join(model, "nnet", "config.json"))
    log.info("Train the model from scratch.")
    return Params(config)


class ValidLoss(object):
    def __init__(self):
        self.min_loss = 1e16
        self.min_loss_epoch = -1


def load_lr(filename):
    out = []
    with open(filename, "r") as f:
        for line in f:
            if line.strip():
                out.append(float(line.strip().split(" ")[1]))
    return out


def load_valid_loss(filename):
    best = ValidLoss()
    with open(filename, "r") as f:
        for line in f:
            if not line.strip():
                continue
            epoch, loss = line.strip().split(" ")[:2]
            if float(loss) < best.min_loss:
                best.min_loss = float(loss)
                best.min_loss_epoch = int(epoch)
    return best


def read_checkpoint_state(model):
    """Parse the TF-style `checkpoint` index file -> (model_checkpoint_path, [all paths])."""
    path = os.path.join(model, "checkpoint")
    if not os.path.isfile(path):
        return None, []
    current, all_paths = None, []
    with open(path) as f:
        for line in f:
            m = re.match(r'\s*(model_checkpoint_path|all_model_checkpoint_paths):\s*"(.*)"', line)
            if not m:
                continue
            if m.group(1) == "model_checkpoint_path":
                current = m.group(2)
            else:
                all_paths.append(m.group(2))
    return current, all_paths


def write_checkpoint_state(model, current, all_paths):
    with open(os.path.join(model, "checkpoint"), "w") as f:
        f.write("model_checkpoint_path: \"%s\"\n" % current)
        for p in all_paths:
            f.write("all_model_checkpoint_paths: \"%s\"\n" % p)


def get_checkpoint(model, checkpoint="-1"):
    """Select a checkpoint and rewrite the index: "last", an explicit step, or -1 = best by
    valid_loss -> (min_epoch + 1) * num_steps_per_epoch (reference utils.py:217-270)."""
    if not os.path.isfile(os.path.join(model, "checkpoint")):
        sys.exit("[ERROR] Cannot find checkpoint in %s." % model)
    current, all_paths = read_checkpoint_state(model)
    if not current:
        sys.exit("[ERROR] Cannot read checkpoint %s." % os.path.join(model, "checkpoint"))
    steps = sorted(int(c.rsplit("-", 1)[1]) for c in all_paths)
    if checkpoint == "last":
        checkpoint = steps[-1]
    else:
        checkpoint = int(checkpoint)
        if checkpoint == -1:
            min_epoch, min_loss = -1, 1e10
            with open(os.path.join(model, "valid_loss")) as f:
                for line in f:
                    if not line.strip():
                        continue
                    epoch, loss = line.split(" ")[:2]
                    if float(loss) < min_loss:
                        min_loss, min_epoch = float(loss), int(epoch)
            params = Params(os.path.join(model, "config.json"))
            checkpoint = (min_epoch + 1) * params.num_steps_per_epoch
    log.info("The checkpoint is %d" % checkpoint)
    assert checkpoint in steps, "The checkpoint %d not in the model directory" % checkpoint
    path = os.path.join(model, os.path.basename(current.rsplit("-", 1)[0] + "-" + str(checkpoint)))
    write_checkpoint_state(model, os.path.basename(path), [os.path.basename(p) for p in all_paths])      # basenames: relocatable
    return path


def get_pretrain_model(pretrain_model, target_model, checkpoint="-1"):
    """Copy one checkpoint of a pre-trained model into `target_model` as step 0 and point the index at it (reference
    utils.py:126-182).  checkpoint: "last", an explicit step, or -1 = best by valid_loss."""
    if not os.path.isfile(os.path.join(pretrain_model, "checkpoint")):
        sys.exit("[ERROR] Cannot find checkpoint in %s." % pretrain_model)
    current, all_paths = read_checkpoint_state(pretrain_model)
    if not current:
        sys.exit("[ERROR] Cannot read checkpoint %s." % os.path.join(pretrain_model, "checkpoint"))
    steps = sorted(int(c.rsplit("-", 1)[1]) for c in all_paths)
    if checkpoint == "last":
        log.info("Load the last saved model.")
        checkpoint = steps[-1]
    else:
        checkpoint = int(checkpoint)
        if checkpoint == -1:
            log.info("Load the best model according to valid_loss")
            min_epoch, min_loss = -1, 1e10
            with open(os.path.join(pretrain_model, "valid_loss")) as f:
                for line in f:
                    if not line.strip():
                        continue
                    epoch, loss = line.split(" ")[:2]
                    if float(loss) < min_loss:
                        min_loss, min_epoch = float(loss), int(epoch)
            params = Params(os.path.join(pretrain_model, "config.json"))
            checkpoint = (min_epoch + 1) * params.num_steps_per_epoch
    assert checkpoint in steps, "The checkpoint %d not in the model directory" % checkpoint
    stem = os.path.basename(current).rsplit("-", 1)[0]
    src = os.path.join(pretrain_model, "%s-%d" % (stem, checkpoint))
    log.info("Copy the pre-trained model %s as the fine-tuned initialization" % src)
    os.makedirs(target_model, exist_ok=True)
    import glob
    import shutil
    for filename in glob.glob(src + ".*"):
        ext = filename[len(src):]
        if ext.endswith(".bak"):
            continue
        shutil.copyfile(filename, os.path.join(target_model, stem + "-0" + ext))
    path = os.path.join(target_model, stem + "-0")
    write_checkpoint_state(target_model, os.path.basename(path), [os.path.basename(path)])


def compute_cos_pairwise_eer(embeddings, labels, max_num_embeddings=1000):
    """Pairwise cosine EER (reference utils.py:273-312): L2-normalise, subsample to <= max_num_embeddings
    with an integer stride, score all pairs i<j, EER = root of 1 - x - tpr(x) on the ROC curve."""
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn import metrics
    embeddings = embeddings / np.sqrt(np.sum(embeddings ** 2, axis=1, keepdims=True) + 1e-12)
    labels = np.asarray(labels)
    n = embeddings.shape[0]
    if n > max_num_embeddings:
        step = n // max_num_embeddings          # py2 integer division in the reference
        idx = np.arange(0, n, step)
        embeddings, labels = embeddings[idx], labels[idx]
        n = embeddings.shape[0]
    score_mat = embeddings @ embeddings.T
    iu = np.triu_indices(n, k=1)                # same (i, j>i) order as the reference's double loop
    scores = score_mat[iu]
    keys = (labels[iu[0]] == labels[iu[1]]).astype(np.float64)
    fpr, tpr, _ = metrics.roc_curve(keys, scores, pos_label=1)
    return float(brentq(lambda x: 1.0 - x - interp1d(fpr, tpr)(x), 0.0, 1.0))


def substring_in_list(s, varlist):
    if varlist is None:
        return False
    return any(v in s for v in varlist)


def activation_summaries(endpoints):
    """TensorBoard histograms in the reference (utils.py:333-346); this engine logs scalars only."""
    return None


def remove_params_prefix(params, prefix):
    """A NEW ParamsPlain with `<prefix>_` cut off the keys that carry it and every other key as it was (reference utils.py:349-358; the
    argument is left untouched, a prefixed key replaces an unprefixed one of the same name when it comes later in the dict)."""
    out = ParamsPlain()
    cut = prefix + "_"
    for k, v in params.dict.items():
        out.dict[k[len(cut):] if k[:len(cut)] == cut else k] = v
    return out


def add_dict_prefix(d, prefix):
    return {prefix + "_" + k: v for k, v in d.items()}


# ------------------------------------------------------------------------------------------
# Pure host logic of the reference drivers, factored out so it can be tested without a GPU
# ------------------------------------------------------------------------------------------
def tune_learning_rate(epoch, lr_now, valid_loss, min_valid_loss, reduce_lr_epochs):
    """LR-halving state machine of egs/voxceleb/v1/nnet/lib/train.py:108-120.
    Mutates `min_valid_loss` (ValidLoss) exactly as the reference does - including the
    `min_loss_epoch += 2` bump after a halving - and returns the learning rate of epoch+1."""
    new_lr = lr_now
    if valid_loss < min_valid_loss.min_loss:
        min_valid_loss.min_loss = valid_loss
        min_valid_loss.min_loss_epoch = epoch
    elif epoch - min_valid_loss.min_loss_epoch >= reduce_lr_epochs:
        new_lr = lr_now / 2
        log.info("After epoch %d, no improvement. Reduce the learning rate to %.8f" % (min_valid_loss.min_loss_epoch, new_lr))
        min_valid_loss.min_loss_epoch += 2
    return new_lr


def should_stop(epoch, next_lr, min_valid_loss, min_learning_rate, early_stop_epochs):
    """train.py:134-139."""
    return next_lr < (min_learning_rate - 1e-12) or epoch - min_valid_loss.min_loss_epoch >= early_stop_epochs


class EpochLedger(object):
    """The epoch-level bookkeeping of a training run and its side-car files under <model>/nnet (what the reference keeps in
    loose variables across egs/voxceleb/v1/nnet/lib/train.py:44-139):

        feature_dim     one line, the feature dimension                       (read by extract.py)
        learning_rate   "<epoch> <lr>" per line, epoch e = the rate epoch e ran / will run with
        valid_loss      "<epoch> <loss> <eer>" per line                      (get_checkpoint picks the best epoch from it)

    The learning rate of an epoch comes from one of three sources: a schedule file named by params.learning_rate (one rate
    per line, fixed in advance), the learning_rate file of a run being continued, or params.learning_rate itself; in the
    last two cases the next rate is decided after each epoch by tune_learning_rate / should_stop."""

    def __init__(self, model_dir, params, first_epoch, default_early_stop=10):
        self.dir = model_dir
        self.params = params
        self.first_epoch = int(first_epoch)
        p = params.dict
        p.setdefault("early_stop_epochs", default_early_stop)
        p.setdefault("min_learning_rate", 1e-5)
        spec = params.learning_rate
        self.fixed_schedule = os.path.isfile(str(spec))
        own = os.path.join(model_dir, "learning_rate")
        if self.fixed_schedule:
            with open(str(spec)) as f:
                self.rates = [float(line) for line in f if line.strip()]
            assert len(self.rates) > params.num_epochs, "The learning rate file is shorter than the num of epochs."
            log.info("Using specified learning rate decay strategy.")
        elif os.path.isfile(own):
            self.rates = load_lr(own)
            assert len(self.rates) == self.first_epoch + 1, "Not enough learning rates in the learning_rate file."
        else:
            self.rates = [float(spec)] * (self.first_epoch + 1)
        best = os.path.join(model_dir, "valid_loss")
        self.best = load_valid_loss(best) if os.path.isfile(best) else ValidLoss()

    def write_feature_dim(self, dim):
        with open(os.path.join(self.dir, "feature_dim"), "w") as f:
            f.write("%d\n" % dim)

    def rate(self, epoch):
        return self.rates[epoch]

    def close_epoch(self, epoch, valid_loss, eer):
        """Record the epoch's validation result, derive the next learning rate, append the side-car lines; returns True when
        training should stop (rate below min_learning_rate, or no improvement for early_stop_epochs)."""
        p = self.params
        if not self.fixed_schedule:
            self.rates.append(tune_learning_rate(epoch, self.rates[epoch], valid_loss, self.best, p.reduce_lr_epochs))
        with open(os.path.join(self.dir, "learning_rate"), "a") as f:
            if epoch == 0:
                f.write("0 %.8f\n" % self.rates[0])
            f.write("%d %.8f\n" % (epoch + 1, self.rates[epoch + 1]))
        with open(os.path.join(self.dir, "valid_loss"), "a") as f:
            f.write("%d %f %f\n" % (epoch, valid_loss, eer))
        if self.fixed_schedule:
            return False
        return should_stop(epoch, self.rates[epoch + 1], self.best, p.min_learning_rate, p.early_stop_epochs)

    def adopt(self, epoch, next_rate):
        """A rank that did not evaluate takes over rank 0's decision for the next epoch."""
        if len(self.rates) <= epoch + 1:
            self.rates.append(float(next_rate))


def checkpoint_step(model_dir):
    """Global step of the checkpoint the index file points at (the digits that end its name), or None."""
    current, _ = read_checkpoint_state(model_dir)
    if not current:
        return None
    return int(re.search(r"(\d+)(?!.*\d)", os.path.basename(current)).group(1))


def split_into_chunks(num_frames, chunk_size):
    """[(start, length)] of the half-overlapping chunks extract.py:69-79 cuts a long utterance into.
    (`chunk_size / 2` is Python-2 integer division in the reference.)"""
    if num_frames <= chunk_size:
        return [(0, num_frames)]
    half = chunk_size // 2
    num_chunks = int(np.ceil(float(num_frames - chunk_size) / half)) + 1
    out = []
    for i in range(num_chunks):
        start = i * half
        out.append((start, chunk_size if num_frames - start > chunk_size else num_frames - start))
    return out


def average_chunk_embeddings(embeddings, lengths, normalize):
    """Length-weighted mean of per-chunk embeddings (extract.py:81-89)."""
    embeddings = np.array(embeddings, dtype=np.float32)
    lengths = np.expand_dims(np.array(lengths), axis=1)
    if normalize:
        embeddings = embeddings / np.sqrt(np.sum(np.square(embeddings), axis=1, keepdims=True))
    return np.sum(embeddings * lengths, axis=0) / np.sum(lengths)


def utterance_embedding(predict, feature, chunk_size, normalize):
    """One embedding per utterance the way extract.py:65-94 produces it: utterances of at most chunk_size frames in one piece, longer
    ones as the length-weighted mean over half-overlapping chunks (the full-length chunks go through `predict` as one batch, the
    shorter last one on its own); optionally L2-normalised per chunk and at the end.  Returns (embedding, number of chunks)."""
    chunks = split_into_chunks(feature.shape[0], chunk_size)
    if len(chunks) == 1:
        embedding = predict(feature)
    else:
        body = predict(np.array([feature[s:s + n] for s, n in chunks[:-1]], dtype=np.float32))
        s, n = chunks[-1]
        tail = predict(feature[s:s + n])
        embedding = average_chunk_embeddings(np.concatenate([body, tail[None]], axis=0), [n for _, n in chunks], normalize)
    if normalize:
        embedding = embedding / np.sqrt(np.sum(np.square(embedding)))
    return np.asarray(embedding, np.float32), len(chunks)


def plan_length_batches(lengths, max_rows, max_chunks, min_fill=0.9, min_rows=8192):
    """Utterances of different lengths -> padded batches for one forward each: [(indices, t)] with t = the longest utterance of the
    batch.  Sorted by length (longest first), a batch takes utterances while it stays within max_chunks utterances and max_rows =
    chunks x t padded rows, and - once it holds min_rows rows - only members that fill at least min_fill of their padded length: a
    forward of fewer than ~min_rows rows leaves most of the chip idle, so padding is cheaper than a small batch below that size and
    dearer above it.  With a window of a few hundred utterances neighbours in the sorted order differ by a few frames."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    batches, i = [], 0
    while i < len(order):
        t = int(lengths[order[i]])
        cap = max(1, min(int(max_chunks), int(max_rows) // max(t, 1)))
        j = i + 1
        while j < len(order) and j - i < cap and ((j - i) * t < min_rows or lengths[order[j]] >= min_fill * t):
            j += 1
        batches.append((order[i:j], t))
        i = j
    return batches


class EmbeddingWindow(object):
    """utterance_embedding() for a window of utterances at once, in two steps so that the caller can put the next window on the GPU
    before it post-processes this one: the constructor sends every utterance's pieces (itself, or its half-overlapping chunks when
    longer than chunk_size, extract.py:69-79) through ONE predict_batch call - which may return a device tensor, nothing is read back
    yet; results() fetches the embeddings and averages long utterances by piece length (extract.py:81-93).
    features: matrices or kaldi_io.PackedMatrix.  results() -> [(embedding, number of pieces)] in input order."""

    def __init__(self, predict_batch, features, chunk_size, normalize):
        self.normalize = normalize
        self.n = len(features)
        pieces, self.owner = [], []
        for u, f in enumerate(features):
            chunks = split_into_chunks(f.shape[0], chunk_size)
            for s, n in chunks:
                if len(chunks) == 1:
                    pieces.append(f)
                else:
                    pieces.append(f.row_range(s, n) if hasattr(f, "row_range") else f[s:s + n])
                self.owner.append((u, n))
        self.emb = predict_batch(pieces) if pieces else np.zeros((0, 0), np.float32)

    def results(self):
        emb = self.emb if isinstance(self.emb, np.ndarray) else self.emb.cpu().numpy()      # (a torch tensor: the read-back waits for the GPU here)
        owner, normalize = self.owner, self.normalize
        out, k = [], 0
        for u in range(self.n):
            m = k
            while m < len(owner) and owner[m][0] == u:
                m += 1
            if m - k == 1:
                e = emb[k]
            else:
                e = average_chunk_embeddings(emb[k:m], [owner[j][1] for j in range(k, m)], normalize)
            if normalize:
                e = e / np.sqrt(np.sum(np.square(e)))
            out.append((np.asarray(e, np.float32), m - k))
            k = m
        return out


def batched_utterance_embeddings(predict_batch, features, chunk_size, normalize):
    return EmbeddingWindow(predict_batch, features, chunk_size, normalize).results()


def prefetch_iter(iterable, depth=8):
    """The items of `iterable`, produced by a background thread up to `depth` ahead of the consumer (extract.py: reading and decoding
    utterance i+1.. from the ark while the GPU runs utterance i - the host side was half of the 0.63 ms per utterance).  Exceptions of
    the producer are re-raised in the consumer; the thread is a daemon and stops with the iterator."""
    import queue
    import threading
    q = queue.Queue(maxsize=max(1, int(depth)))
    end, stop = object(), threading.Event()

    def produce():
        try:
            for item in iterable:
                while not stop.is_set():
                    try:
                        q.put((item, None), timeout=0.1)
                        break
                    except queue.Full:
                        continue
                if stop.is_set():
                    return
            q.put((end, None))
        except BaseException as exc:       # noqa: B902 - handed to the consumer
            q.put((end, exc))

    threading.Thread(target=produce, name="prefetch", daemon=True).start()
    try:
        while True:
            item, exc = q.get()
            if item is end:
                if exc is not None:
                    raise exc
                return
            yield item
    finally:
        stop.set()
