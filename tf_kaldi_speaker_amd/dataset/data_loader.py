"""Minibatch loaders over a Kaldi data directory - the host side that feeds Trainer.train / .valid.

Mirrors the reference's dataset/data_loader.py API (KaldiDataRandomQueue :310, KaldiDataSeqQueue
:467, DataOutOfRange, get_speaker_info :14) and its sampling rules (batch_random :229-307,
batch_sequence :417-464):
  random queue : sample `num_speakers` speakers -> ONE random length T in [min_len, max_len] for the
                 whole batch -> per speaker keep the utterances with num_frames > T (resample the
                 speaker if none) -> `num_segments` utterances -> random start frame.
  seq queue    : every utterance once, batches of `batch_size`, T = min(random T, shortest utt).
Workers are daemon processes started with the *spawn* method (the parent owns a HIP context;
forking it is not safe) feeding a bounded multiprocessing.Queue.  NumPy only - no torch import here.
"""
import multiprocessing as mp
import os
import random
import time

import numpy as np

from .kaldi_io import FeatureReader


class DataOutOfRange(Exception):
    pass


def get_speaker_info(data, spklist):
    """(spk2features, features2spk, spk2index); a feature is the string "utt path:offset"."""
    assert os.path.isdir(data) and os.path.isfile(spklist)
    spk2index = {}
    with open(spklist, "r") as f:
        for line in f:
            spk, index = line.strip().split(" ")
            spk2index[spk] = int(index)
    utt2spk = {}
    with open(os.path.join(data, "spk2utt"), "r") as f:
        for line in f:
            spk, utts = line.strip().split(" ", 1)
            for utt in utts.split(" "):
                utt2spk[utt] = spk2index[spk]
    spk2features, features2spk = {}, {}
    with open(os.path.join(data, "feats.scp"), "r") as f:
        for line in f:
            key, rxfile = line.strip().split(" ", 1)
            spk = utt2spk[key]
            feat = key + " " + rxfile
            spk2features.setdefault(spk, []).append(feat)
            features2spk[feat] = spk
    return spk2features, features2spk, spk2index


def sample_random_batch(rd, reader, spk2features, speakers, num_speakers, num_segments, min_len, max_len, shuffle):
    """One batch of the random queue (reference batch_random loop body, data_loader.py:271-298)."""
    batch_speakers = rd.sample(speakers, num_speakers)
    batch_length = rd.randint(min_len, max_len)
    features = np.zeros((num_speakers * num_segments, batch_length, reader.dim), dtype=np.float32)
    labels = np.zeros((num_speakers * num_segments), dtype=np.int32)
    for i, speaker in enumerate(batch_speakers):
        spk = speaker
        feature_list = []
        while len(feature_list) == 0:
            feature_list = [f for f in spk2features[spk] if reader.utt2num_frames[f.split(" ")[0]] > batch_length]
            if len(feature_list) == 0:
                spk = rd.choice(list(set(speakers) - set(batch_speakers)))
                batch_speakers[i] = spk
        labels[i * num_segments:(i + 1) * num_segments] = spk
        if len(feature_list) < num_segments:
            feature_list = feature_list * (int(num_segments / len(feature_list)) + 1)
        for j, feat in enumerate(rd.sample(feature_list, num_segments)):
            features[i * num_segments + j], _ = reader.read_segment(feat, batch_length, shuffle=shuffle)
    return features, labels


def batch_random(stop_event, queue, data, spk2features, num_total_speakers, num_speakers=10, num_segments=10,
                 min_len=200, max_len=400, shuffle=True, seed=0):
    rd = random.Random(int.from_bytes(os.urandom(4), "little") + 7919 * seed)
    random.seed(int.from_bytes(os.urandom(4), "little") + 104729 * seed)   # FeatureReader draws start frames here
    reader = FeatureReader(data)
    speakers = list(spk2features.keys())
    if num_total_speakers < num_speakers:
        print("[Warning] The number of available speakers are less than the required speaker. Some speakers will be duplicated.")
        speakers = speakers * (int(num_speakers / num_total_speakers) + 1)
    while not stop_event.is_set():
        batch = sample_random_batch(rd, reader, spk2features, speakers, num_speakers, num_segments, min_len, max_len, shuffle)
        while not stop_event.is_set():
            try:
                queue.put(batch, timeout=0.5)
                break
            except Exception:
                continue
    reader.close()


def batch_sequence(stop_event, queue, data, feature_list, features2spk, batch_size=128, min_len=200, max_len=400,
                   shuffle=True, seed=0):
    rd = random.Random(int.from_bytes(os.urandom(4), "little") + 7919 * seed)
    random.seed(int.from_bytes(os.urandom(4), "little") + 104729 * seed)
    reader = FeatureReader(data)
    num_batches = int(len(feature_list) / batch_size)
    for i in range(num_batches):
        chunk = feature_list[i * batch_size:(i + 1) * batch_size]
        batch_length = rd.randint(min_len, max_len)
        for feat in chunk:
            n = reader.utt2num_frames[feat.split(" ")[0]]
            if n < batch_length:
                batch_length = n
        features = np.zeros((batch_size, batch_length, reader.dim), dtype=np.float32)
        labels = np.zeros((batch_size), dtype=np.int32)
        for j, feat in enumerate(chunk):
            features[j], _ = reader.read_segment(feat, batch_length, shuffle=shuffle)
            labels[j] = features2spk[feat]
        queue.put((features, labels))
    stop_event.set()
    reader.close()


def _ctx():
    return mp.get_context("spawn")


class KaldiDataRandomQueue(object):
    """Endless stream of random (features [B,T,D] f32, labels [B] i32) batches."""

    def __init__(self, data_dir, spklist, num_parallel=1, max_qsize=10, num_speakers=None, num_segments=None,
                 min_len=None, max_len=None, shuffle=True):
        self.data = data_dir
        self.num_speakers = num_speakers
        self.num_segments = num_segments
        self.min_len = min_len
        self.max_len = max_len
        self.num_parallel_datasets = num_parallel
        self.shuffle = shuffle
        self.spk2features, self.features2spk, spk2index = get_speaker_info(data_dir, spklist)
        self.num_total_speakers = len(list(spk2index.keys()))
        self._mp = _ctx()
        self.queue = self._mp.Queue(max_qsize)
        self.stop_event = self._mp.Event()
        self.processes = []

    def set_batch(self, num_speakers, num_segments):
        self.num_speakers = num_speakers
        self.num_segments = num_segments

    def set_length(self, min_len, max_len):
        self.min_len = min_len
        self.max_len = max_len

    def start(self):
        self.processes = [self._mp.Process(target=batch_random,
                                           args=(self.stop_event, self.queue, self.data, self.spk2features,
                                                 self.num_total_speakers, self.num_speakers, self.num_segments,
                                                 self.min_len, self.max_len, self.shuffle, i))
                          for i in range(self.num_parallel_datasets)]
        for p in self.processes:
            p.daemon = True
            p.start()

    def fetch(self):
        return self.queue.get()

    def stop(self):
        self.stop_event.set()
        deadline = time.time() + 5.0
        while time.time() < deadline and any(p.is_alive() for p in self.processes):
            try:
                self.queue.get(timeout=0.1)
            except Exception:
                pass
        for p in self.processes:
            if p.is_alive():
                p.terminate()
            p.join(1.0)
        self.processes = []


class KaldiDataSeqQueue(object):
    """Every utterance once per pass (validation); raises DataOutOfRange when exhausted."""

    def __init__(self, data_dir, spklist, num_parallel=1, max_qsize=10, batch_size=128, min_len=None, max_len=None,
                 shuffle=True):
        self.data = data_dir
        self.batch_size = batch_size
        self.min_len = min_len
        self.max_len = max_len
        self.num_parallel_datasets = num_parallel
        self.shuffle = shuffle
        self.spk2features, self.features2spk, spk2index = get_speaker_info(data_dir, spklist)
        self.num_total_speakers = len(list(spk2index.keys()))
        self.feature_list = []
        for spk in self.spk2features:
            self.feature_list += self.spk2features[spk]
        if shuffle:
            random.shuffle(self.feature_list)
        n_sub = len(self.feature_list) // num_parallel
        self.sub_feature_list = []
        for i in range(num_parallel):
            if i == num_parallel - 1:
                self.sub_feature_list.append(self.feature_list[i * n_sub:])
            else:
                self.sub_feature_list.append(self.feature_list[i * n_sub:(i + 1) * n_sub])
        self._mp = _ctx()
        self.queue = self._mp.Queue(max_qsize)
        self.stop_event = [self._mp.Event() for _ in range(num_parallel)]
        self.processes = []

    def set_batch(self, batch_size):
        self.batch_size = batch_size

    def set_length(self, min_len, max_len):
        self.min_len = min_len
        self.max_len = max_len

    def start(self):
        self.processes = [self._mp.Process(target=batch_sequence,
                                           args=(self.stop_event[i], self.queue, self.data, self.sub_feature_list[i],
                                                 self.features2spk, self.batch_size, self.min_len, self.max_len,
                                                 self.shuffle, i))
                          for i in range(self.num_parallel_datasets)]
        for p in self.processes:
            p.daemon = True
            p.start()

    def fetch(self):
        while True:
            try:
                return self.queue.get(timeout=0.2)
            except Exception:
                if all(e.is_set() for e in self.stop_event) and self.queue.empty():
                    self.stop()
                    raise DataOutOfRange

    def stop(self):
        for p in self.processes:
            if p.is_alive():
                p.terminate()
            p.join(1.0)
        self.processes = []
