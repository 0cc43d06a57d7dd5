"""Minibatch queues over a Kaldi data directory with the reference's names (dataset/data_loader.py:
get_speaker_info :14, KaldiDataRandomQueue :310, KaldiDataSeqQueue :467, DataOutOfRange) and batch contract
(features [B,T,D] float32 + labels [B] int32, one length T per batch) - built differently:

  * the directory is parsed ONCE into a flat utterance table (`KaldiIndex`: NumPy columns for ark, byte offset, frame
    count and speaker id, plus one index array per speaker) instead of dictionaries of "utt path:offset" strings;
  * a batch is first *planned* - which utterance, which start frame, which T (`plan_random_batch`,
    `plan_sequence_batches`: the reference's sampling rules, data_loader.py:271-298 and :417-464, on the index
    arrays) - and then *filled* by the native codec (`libxvector_io.so::xvio_read_rows`, which decodes only the planned
    rows of a 'CM ' matrix) on a pool of threads that write straight into the batch array; ctypes releases the GIL, so
    there are no worker processes, no pickling and no multiprocessing.Queue;
  * the training stream `KaldiDataRandomQueue` IS the C++ loader (`NativeRandomQueue`: planning and decoding both in
    native threads, pinned staging, asynchronous H2D); `PlannedRandomQueue` is the same stream planned in Python, kept
    for tests of the sampling rules and as `XV_LOADER=python`.
"""
import concurrent.futures
import os
import queue as _queue
import threading

import numpy as np

from . import native_loader
from .native_loader import NativeRandomQueue


class DataOutOfRange(Exception):
    pass


class KaldiIndex(object):
    """Flat table of the utterances of a Kaldi data directory whose speaker is listed in `spklist`.

    Columns (one entry per feats.scp line, in file order): key, ark (index into .arks), offset, frames, speaker.
    .by_speaker[s] = utterance rows of speaker id s; .speakers = ids that own at least one utterance."""

    def __init__(self, data, spklist):
        assert os.path.isdir(data) and os.path.isfile(spklist)
        self.data = data
        self.spk2index = {}
        with open(spklist) as f:
            for line in f:
                name, idx = line.split()
                self.spk2index[name] = int(idx)
        owner = {}
        with open(os.path.join(data, "spk2utt")) as f:
            for line in f:
                fields = line.split()
                sid = self.spk2index[fields[0]]          # a speaker missing from spklist is an error, as in the reference
                for utt in fields[1:]:
                    owner[utt] = sid
        nframes = {}
        path = os.path.join(data, "utt2num_frames")
        assert os.path.exists(path), "[Error] Expect utt2num_frames exists in %s " % data
        with open(path) as f:
            for line in f:
                utt, n = line.split()
                nframes[utt] = int(n)
        keys, ark, offset, frames, speaker = [], [], [], [], []
        self.arks, ark_id = [], {}
        with open(os.path.join(data, "feats.scp")) as f:
            for line in f:
                key, rx = line.strip().split(" ", 1)
                fname, off = rx.rsplit(":", 1)
                if fname not in ark_id:
                    ark_id[fname] = len(self.arks)
                    self.arks.append(fname)
                keys.append(key)
                ark.append(ark_id[fname])
                offset.append(int(off))
                frames.append(nframes[key])
                speaker.append(owner[key])
        self.keys = keys
        self.ark = np.asarray(ark, np.int32)
        self.offset = np.asarray(offset, np.int64)
        self.frames = np.asarray(frames, np.int32)
        self.speaker = np.asarray(speaker, np.int32)
        order = np.argsort(self.speaker, kind="stable")
        ids, first = np.unique(self.speaker[order], return_index=True)
        bounds = list(first) + [len(order)]
        self.by_speaker = {int(s): order[bounds[i]:bounds[i + 1]] for i, s in enumerate(ids)}
        self.speakers = np.asarray(sorted(self.by_speaker), np.int32)
        self.num_total_speakers = len(self.spk2index)
        self._dim = None

    def __len__(self):
        return len(self.keys)

    @property
    def dim(self):
        if self._dim is None:
            self._dim = native_loader.read_rows(self.arks[self.ark[0]], int(self.offset[0]), 0, 1, max_elems=1 << 16).shape[1]
        return self._dim

    def feature_name(self, u):
        """The reference's "utt path:offset" string of utterance row u."""
        return "%s %s:%d" % (self.keys[u], self.arks[self.ark[u]], self.offset[u])


def get_speaker_info(data, spklist):
    """(spk2features, features2spk, spk2index) in the reference's shape (data_loader.py:14-60): speaker id -> list of
    "utt path:offset" strings and back.  A view of KaldiIndex for callers of the reference API; the queues below work on
    the index arrays."""
    index = KaldiIndex(data, spklist)
    spk2features, features2spk = {}, {}
    for u in range(len(index)):
        name, s = index.feature_name(u), int(index.speaker[u])
        spk2features.setdefault(s, []).append(name)
        features2spk[name] = s
    return spk2features, features2spk, index.spk2index


class BatchPlan(object):
    __slots__ = ("utts", "starts", "length", "labels")

    def __init__(self, utts, starts, length, labels):
        self.utts, self.starts, self.length, self.labels = utts, starts, int(length), labels


def _starts(index, rng, utts, length, shuffle):
    if not shuffle:
        return np.zeros(len(utts), np.int64)
    room = index.frames[utts].astype(np.int64) - length          # a start frame uniform in [0, frames - T]
    return (rng.random(len(utts)) * (room + 1)).astype(np.int64).clip(0, np.maximum(room, 0))


def plan_random_batch(index, rng, num_speakers, num_segments, min_len, max_len, shuffle=True):
    """The random queue's rules (reference data_loader.py:271-298): `num_speakers` distinct speakers; ONE length T in
    [min_len, max_len] for the whole batch; per speaker only utterances with MORE than T frames qualify - a speaker with
    none is replaced by one drawn from outside the batch's original pick; `num_segments` different utterances of the
    speaker (repeating the list first when it is shorter); a random start frame each.  rng: numpy Generator."""
    pool = index.speakers
    if len(pool) < num_speakers:       # fewer speakers than asked for: some are duplicated (reference :263-266)
        pool = np.tile(pool, num_speakers // len(pool) + 1)
    picked = rng.choice(pool, size=num_speakers, replace=False)
    length = int(rng.integers(min_len, max_len + 1))
    spare = np.setdiff1d(pool, picked)
    utts = np.empty(num_speakers * num_segments, np.int64)
    labels = np.empty(num_speakers * num_segments, np.int32)
    for i in range(num_speakers):
        s = int(picked[i])
        while True:
            mine = index.by_speaker[s]
            ok = mine[index.frames[mine] > length]
            if len(ok):
                break
            if len(spare) == 0:
                # NOT DataOutOfRange: Trainer.train_batches takes that as the regular end of a sequence queue's data, so a data
                # directory whose utterances are all too short would end every epoch silently (and hang the other ranks of a
                # data-parallel run in their all-reduce).  The reference loops forever here (data_loader.py:281-290).
                raise ValueError("no utterance longer than %d frames outside the sampled speakers: max_segment_len is larger than the "
                                 "data allows" % length)
            s = int(rng.choice(spare))
            spare = spare[spare != s]      # in the batch now (or found too short): not a candidate for a later replacement
        if len(ok) < num_segments:
            ok = np.tile(ok, num_segments // len(ok) + 1)
        utts[i * num_segments:(i + 1) * num_segments] = rng.choice(ok, size=num_segments, replace=False)
        labels[i * num_segments:(i + 1) * num_segments] = s
    return BatchPlan(utts, _starts(index, rng, utts, length, shuffle), length, labels)


def plan_sequence_batches(index, utt_order, batch_size, rng, min_len, max_len, shuffle=True):
    """Every utterance of `utt_order` once, in batches of `batch_size` (an incomplete last batch is dropped, reference
    data_loader.py:443); T = a random length in [min_len, max_len], shortened to the shortest utterance of the batch."""
    for b in range(len(utt_order) // batch_size):
        utts = np.asarray(utt_order[b * batch_size:(b + 1) * batch_size], np.int64)
        length = min(int(rng.integers(min_len, max_len + 1)), int(index.frames[utts].min()))
        yield BatchPlan(utts, _starts(index, rng, utts, length, shuffle), length, index.speaker[utts].astype(np.int32))


class PlanReader(object):
    """Fills planned batches with the native codec on a thread pool (one task per utterance segment, written in place)."""

    def __init__(self, index, threads=4):
        self.index = index
        self.lib = native_loader.load()
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, int(threads)))
        self._arks = [a.encode() for a in index.arks]

    def _segment(self, out_row, u, start, length):
        import ctypes as C
        rows, cols = C.c_int32(), C.c_int32()
        rc = self.lib.xvio_read_rows(self._arks[self.index.ark[u]], int(self.index.offset[u]), int(start), int(length),
                                     out_row.ctypes.data, out_row.size, C.byref(rows), C.byref(cols))
        if rc != 0 or rows.value != length:
            raise IOError("Cannot read features from %s (%s)" % (self.index.feature_name(u),
                                                                 self.lib.xvio_last_error().decode(errors="replace")))

    def read(self, plan):
        feats = np.empty((len(plan.utts), plan.length, self.index.dim), np.float32)
        jobs = [self.pool.submit(self._segment, feats[j], int(plan.utts[j]), int(plan.starts[j]), plan.length)
                for j in range(len(plan.utts))]
        for j in jobs:
            j.result()
        return feats, plan.labels

    def close(self):
        self.pool.shutdown(wait=True)


class _Prefetcher(threading.Thread):
    """Runs a (plan iterator -> batches) pipeline ahead of the consumer through a bounded queue."""

    def __init__(self, plans, reader, depth):
        threading.Thread.__init__(self, daemon=True)
        self.plans, self.reader = plans, reader
        self.out = _queue.Queue(max(1, int(depth)))
        self.halt = threading.Event()
        self.failure = None          # the exception that ended the pipeline: raised again by every later get()
        self.ended = False

    def run(self):
        try:
            for plan in self.plans:
                if self.halt.is_set():
                    return
                item = self.reader.read(plan)
                while not self.halt.is_set():
                    try:
                        self.out.put(item, timeout=0.2)
                        break
                    except _queue.Full:
                        continue
            self.out.put(None)
        except Exception as exc:      # surfaces in fetch()
            self.out.put(exc)

    def get(self):
        if self.failure is not None:       # the thread is gone: a second get() would wait on an empty queue for ever
            raise self.failure
        if self.ended:
            return None
        item = self.out.get()
        if isinstance(item, Exception):
            self.failure = item
            raise item
        if item is None:
            self.ended = True
        return item


class _QueueBase(object):
    def _setup(self, data_dir, spklist, num_parallel, max_qsize, min_len, max_len, shuffle):
        self.data = data_dir
        self.index = KaldiIndex(data_dir, spklist)
        self.num_total_speakers = self.index.num_total_speakers
        self.num_parallel_datasets = num_parallel
        self.max_qsize = max_qsize
        self.min_len, self.max_len, self.shuffle = min_len, max_len, shuffle
        self._reader = self._thread = None

    def set_length(self, min_len, max_len):
        self.min_len, self.max_len = min_len, max_len

    def _launch(self, plans):
        self._reader = PlanReader(self.index, threads=max(2, 2 * int(self.num_parallel_datasets)))
        self._thread = _Prefetcher(plans, self._reader, self.max_qsize)
        self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._thread.halt.set()
            while self._thread.is_alive():          # unblock a producer waiting on a full queue
                try:
                    self._thread.out.get(timeout=0.05)
                except _queue.Empty:
                    pass
            self._thread = None
        if self._reader is not None:
            self._reader.close()
            self._reader = None


class PlannedRandomQueue(_QueueBase):
    """The random training stream planned in Python (plan_random_batch) and decoded by the native codec: the readable
    twin of NativeRandomQueue, selected with XV_LOADER=python."""

    def __init__(self, data_dir, spklist, num_parallel=1, max_qsize=10, num_speakers=None, num_segments=None,
                 min_len=None, max_len=None, shuffle=True, seed=None):
        self._setup(data_dir, spklist, num_parallel, max_qsize, min_len, max_len, shuffle)
        self.num_speakers, self.num_segments = num_speakers, num_segments
        self.seed = seed

    def set_batch(self, num_speakers, num_segments):
        self.num_speakers, self.num_segments = num_speakers, num_segments

    def start(self):
        rng = np.random.default_rng(self.seed)

        def plans():
            while True:
                yield plan_random_batch(self.index, rng, self.num_speakers, self.num_segments, self.min_len, self.max_len,
                                        self.shuffle)
        self._launch(plans())

    def fetch(self):
        return self._thread.get()


class KaldiDataRandomQueue(NativeRandomQueue):
    """Endless stream of random (features [B,T,D] f32, labels [B] i32) batches: the C++ loader under the reference's
    class name.  `num_total_speakers` is available before start() (train.py:74 reads it from a bare instance)."""

    def __init__(self, data_dir, spklist, num_parallel=1, max_qsize=10, num_speakers=None, num_segments=None,
                 min_len=None, max_len=None, shuffle=True, seed=None, packed=False):
        NativeRandomQueue.__init__(self, data_dir, spklist, num_parallel=num_parallel, max_qsize=max_qsize,
                                   num_speakers=num_speakers, num_segments=num_segments, min_len=min_len, max_len=max_len,
                                   shuffle=shuffle, seed=seed, packed=packed)
        with open(spklist) as f:
            self.num_total_speakers = sum(1 for line in f if line.strip())


class KaldiDataSeqQueue(_QueueBase):
    """Every utterance once per pass (validation); fetch() raises DataOutOfRange when the pass is over.  The utterance
    list is cut into `num_parallel` runs, each batched on its own (so up to batch_size - 1 utterances per run are dropped,
    as the reference's per-worker batching does, data_loader.py:443,530-537)."""

    def __init__(self, data_dir, spklist, num_parallel=1, max_qsize=10, batch_size=128, min_len=None, max_len=None,
                 shuffle=True, seed=None):
        self._setup(data_dir, spklist, num_parallel, max_qsize, min_len, max_len, shuffle)
        self.batch_size = batch_size
        self._rng = np.random.default_rng(seed)
        order = np.concatenate([self.index.by_speaker[int(s)] for s in self.index.speakers]) if len(self.index) else np.empty(0, np.int64)
        if shuffle:
            order = self._rng.permutation(order)
        n_sub = len(order) // num_parallel
        self.runs = [order[i * n_sub:(i + 1) * n_sub] if i < num_parallel - 1 else order[i * n_sub:] for i in range(num_parallel)]

    def set_batch(self, batch_size):
        self.batch_size = batch_size

    def start(self):
        def plans():
            for run in self.runs:
                for plan in plan_sequence_batches(self.index, run, self.batch_size, self._rng, self.min_len, self.max_len,
                                                  self.shuffle):
                    yield plan
        self._launch(plans())

    def fetch(self):
        item = self._thread.get() if self._thread is not None else None
        if item is None:
            self.stop()
            raise DataOutOfRange
        return item
