"""CPU tests of the host side of the boundary: Kaldi codecs against golden vectors produced by the
reference's own kaldi_io (tests/golden/make_kaldi_golden.py), the loaders' sampling rules, config /
checkpoint-index helpers, EER, and the pure logic of the train / extract drivers."""
import io
import json
import os

import numpy as np
import pytest

from tf_kaldi_speaker_amd.dataset import kaldi_io
from tf_kaldi_speaker_amd.dataset.data_loader import (KaldiDataRandomQueue, KaldiDataSeqQueue, PlannedRandomQueue, DataOutOfRange,
                                                      get_speaker_info, KaldiIndex, PlanReader, plan_random_batch)
from tf_kaldi_speaker_amd.misc import utils as U
from tests.kaldi_fixture import make_data_dir

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "kaldi_golden.npz"))


def test_read_reference_written_fm_and_fv_arks():
    got = list(kaldi_io.read_mat_ark(io.BytesIO(G["fm_bytes"].tobytes())))
    assert [k for k, _ in got] == ["utt-a", "utt-b"]
    assert np.array_equal(got[0][1], G["fm_m1"]) and np.array_equal(got[1][1], G["fm_m2"])
    got = list(kaldi_io.read_vec_flt_ark(io.BytesIO(G["fv_bytes"].tobytes())))
    assert [k for k, _ in got] == ["spk1-utt1", "spk1-utt2"]
    assert np.array_equal(got[0][1], G["fv_v1"]) and np.array_equal(got[1][1], G["fv_v2"])


def test_writers_are_byte_identical_to_reference():
    class B(io.BytesIO):
        def close(self):
            pass
    b = B()
    kaldi_io.write_mat(b, G["fm_m1"], key="utt-a")
    kaldi_io.write_mat(b, G["fm_m2"], key="utt-b")
    assert b.getvalue() == G["fm_bytes"].tobytes()
    b = B()
    kaldi_io.write_vec_flt(b, G["fv_v1"], key="spk1-utt1")
    kaldi_io.write_vec_flt(b, G["fv_v2"], key="spk1-utt2")
    assert b.getvalue() == G["fv_bytes"].tobytes()       # "<key> \0BFV \x04<int32 dim><data>", kaldi_io.py:640-653


def test_compressed_matrix_decode_matches_reference_decoder_bit_exact():
    raw = G["cm_bytes"].tobytes()
    (key, mat), = list(kaldi_io.read_mat_ark(io.BytesIO(raw)))
    assert key == "cm-utt"
    assert np.array_equal(mat, G["cm_full"])              # same bytes -> same float32 values as the reference
    for name in G.files:
        if not name.startswith("cm_sub_"):
            continue
        start, length = (int(v) for v in name.split("_")[2:])
        fd = io.BytesIO(raw)
        kaldi_io.read_key(fd)
        assert fd.read(2) == b"\0B"
        sub = kaldi_io._read_submat_binary(fd, start, length)
        assert np.array_equal(sub, G[name]), name
        assert np.array_equal(sub, G["cm_full"][start:start + length])
    # the encoder is lossy by design (8 bit): within 1/64 of the inter-quartile span per column
    err = np.abs(G["cm_full"] - G["cm_source"]).max(axis=0)
    span = np.maximum(G["cm_source"].max(axis=0) - G["cm_source"].min(axis=0), 1e-2)
    assert (err / span).max() < 0.03


def test_uncompressed_training_matrices_are_rejected_like_the_reference():
    fd = io.BytesIO(G["fm_bytes"].tobytes())
    kaldi_io.read_key(fd)
    fd.read(2)
    with pytest.raises(ValueError):
        kaldi_io._read_submat_binary(fd, 0, 5)            # kaldi_io.py:743-749


def test_feature_reader_and_speaker_info(tmp_path):
    data, spklist, mats = make_data_dir(str(tmp_path / "train"))
    rd = kaldi_io.FeatureReader(data)
    assert rd.dim == 30
    spk2features, features2spk, spk2index = get_speaker_info(data, spklist)
    assert len(spk2index) == 6 and sorted(spk2features) == list(range(6))
    feat = spk2features[2][1]
    utt = feat.split(" ")[0]
    full, _ = rd.read(feat)
    assert full.shape == mats[utt].shape and np.abs(full - mats[utt]).max() < 0.2
    seg, start = rd.read_segment(feat, 20, shuffle=False, start=7)
    assert start == 7 and np.array_equal(seg, full[7:27])
    seg, start = rd.read_segment(feat, 20, shuffle=True)
    assert 0 <= start <= full.shape[0] - 20 and np.array_equal(seg, full[start:start + 20])
    whole, _ = rd.read_segment(feat, 10 ** 6)             # longer than the utterance -> clipped
    assert whole.shape[0] == full.shape[0]
    rd.close()


def test_random_batch_sampling_rules(tmp_path):
    """plan_random_batch on the utterance index: N distinct speakers x M segments, one T per batch, only utterances with
    more than T frames, start frames inside the utterance; the planned rows come back from the native codec bit-exact
    with the Python reader's."""
    data, spklist, mats = make_data_dir(str(tmp_path / "train"), num_spk=8, utts_per_spk=2, min_frames=50, max_frames=90)
    index = KaldiIndex(data, spklist)
    assert len(index) == 16 and index.dim == 30 and index.num_total_speakers == 8
    rng = np.random.default_rng(3)
    reader = PlanReader(index, threads=3)
    rd = kaldi_io.FeatureReader(data)
    for _ in range(8):
        plan = plan_random_batch(index, rng, 4, 2, 30, 49, True)      # every utterance has >= 50 frames: no speaker needs replacing
        assert 30 <= plan.length <= 49 and plan.labels.dtype == np.int32
        assert len(set(plan.labels[::2])) == 4 and np.array_equal(plan.labels[::2], plan.labels[1::2])   # N speakers x M segments
        assert np.all(index.frames[plan.utts] > plan.length)
        assert np.all(plan.starts >= 0) and np.all(plan.starts + plan.length <= index.frames[plan.utts])
        assert np.array_equal(index.speaker[plan.utts], plan.labels)
        feats, labels = reader.read(plan)
        assert feats.shape == (8, plan.length, 30) and feats.dtype == np.float32
        for j in (0, 5):
            ref, _ = rd.read_segment(index.feature_name(int(plan.utts[j])), plan.length, shuffle=False, start=int(plan.starts[j]))
            assert np.array_equal(feats[j], ref)
    # every speaker too short for T: the plan replaces speakers until none is left, then gives up loudly - and NOT with
    # DataOutOfRange, which Trainer.train takes for the regular end of the data
    with pytest.raises(ValueError) as ei:
        plan_random_batch(index, rng, 8, 1, 95, 95, True)
    assert not isinstance(ei.value, DataOutOfRange)
    # a T only some speakers can serve: the others are replaced from outside the batch's first pick
    long_spk = {int(index.speaker[u]) for u in range(len(index)) if index.frames[u] > 80}
    if 0 < len(long_spk) < 8:
        plan = plan_random_batch(index, rng, min(2, len(long_spk)), 1, 80, 80, True)
        assert set(plan.labels.tolist()) <= long_spk
    reader.close()
    rd.close()


def test_random_queue_failure_is_sticky_and_not_an_end_of_data(tmp_path):
    """A random queue whose data cannot serve the requested length fails in fetch() with the planner's error, and every later
    fetch() raises it again instead of waiting for ever on the dead prefetch thread."""
    data, spklist, _ = make_data_dir(str(tmp_path / "short"), num_spk=4, utts_per_spk=2, min_frames=30, max_frames=40)
    q = PlannedRandomQueue(data, spklist, num_parallel=1, max_qsize=2, num_speakers=3, num_segments=1, min_len=90, max_len=95)
    q.start()
    for _ in range(3):
        with pytest.raises(ValueError):
            q.fetch()
    q.stop()


def test_queues_end_to_end(tmp_path):
    data, spklist, mats = make_data_dir(str(tmp_path / "train"), num_spk=5, utts_per_spk=4, min_frames=45, max_frames=70)
    for cls in (KaldiDataRandomQueue, PlannedRandomQueue):
        q = cls(data, spklist, num_parallel=2, max_qsize=4, num_speakers=3, num_segments=1, min_len=20, max_len=40)
        assert q.num_total_speakers == 5
        q.start()
        for _ in range(3):
            f, l = q.fetch()
            assert f.shape[0] == 3 and 20 <= f.shape[1] <= 40 and len(set(l)) == 3 and f.dtype == np.float32
        q.stop()
    s = KaldiDataSeqQueue(data, spklist, num_parallel=2, max_qsize=4, batch_size=4, min_len=20, max_len=40, shuffle=False)
    s.start()
    seen, keys = 0, []
    with pytest.raises(DataOutOfRange):
        while True:
            f, l = s.fetch()
            assert f.shape[0] == 4 and 20 <= f.shape[1] <= 40 and l.dtype == np.int32
            seen += f.shape[0]
    # 20 utterances in 2 runs of 10 -> int(10 / 4) = 2 full batches per run; the remainder is
    # dropped exactly as in the reference (data_loader.py:443)
    assert seen == 16
    # shuffle=False: frame 0 onwards, labels = the utterances' speakers, file order within a speaker
    s = KaldiDataSeqQueue(data, spklist, num_parallel=1, max_qsize=2, batch_size=5, min_len=30, max_len=30, shuffle=False)
    s.start()
    f, l = s.fetch()
    index = KaldiIndex(data, spklist)
    first = np.concatenate([index.by_speaker[int(sp)] for sp in index.speakers])[:5]
    assert np.array_equal(l, index.speaker[first])
    rd = kaldi_io.FeatureReader(data)
    ref, _ = rd.read_segment(index.feature_name(int(first[2])), 30, shuffle=False, start=0)
    assert np.array_equal(f[2], ref)
    rd.close()
    s.stop()


def test_params_accepts_comment_keys_and_roundtrips(tmp_path):
    cfg = {"Note": "comment keys are legal", "seed": 0, "network_type": "tdnn", "loss_func": "additive_margin_softmax",
           "amsoftmax_m": 0.2, "learning_rate": 0.01, "num_steps_per_epoch": 30000}
    p = tmp_path / "c.json"
    p.write_text(json.dumps(cfg))
    params = U.Params(str(p))
    assert params.amsoftmax_m == 0.2 and params.dict["Note"].startswith("comment")
    params.dict["num_nodes_pooling_layer"] = 1500
    params.save(str(tmp_path / "d.json"))
    assert U.Params(str(tmp_path / "d.json")).num_nodes_pooling_layer == 1500


def test_get_checkpoint_best_last_and_explicit(tmp_path):
    model = tmp_path / "nnet"
    model.mkdir()
    (model / "config.json").write_text(json.dumps({"num_steps_per_epoch": 100}))
    paths = [str(model / ("model-%d" % s)) for s in (100, 200, 300)]
    U.write_checkpoint_state(str(model), paths[-1], paths)
    (model / "valid_loss").write_text("0 2.5 0.10\n1 1.5 0.08\n2 1.9 0.09\n")
    assert U.get_checkpoint(str(model), "-1").endswith("model-200")      # best epoch 1 -> (1+1)*100
    cur, allp = U.read_checkpoint_state(str(model))
    assert cur.endswith("model-200") and len(allp) == 3
    assert U.get_checkpoint(str(model), "last").endswith("model-300")
    assert U.get_checkpoint(str(model), "100").endswith("model-100")
    with pytest.raises(AssertionError):
        U.get_checkpoint(str(model), "150")
    assert U.load_valid_loss(str(model / "valid_loss")).min_loss_epoch == 1


def test_learning_rate_schedule_table():
    """(valid-loss sequence -> LR sequence) derived by reading train.py:108-139: halve after
    `reduce_lr_epochs` epochs without improvement, then push the reference epoch forward by 2."""
    losses = [3.0, 2.5, 2.6, 2.7, 2.8, 2.9, 2.4, 2.5, 2.6, 2.7, 2.8]
    best = U.ValidLoss()
    lrs = [0.01]
    for epoch, loss in enumerate(losses):
        lrs.append(U.tune_learning_rate(epoch, lrs[epoch], loss, best, reduce_lr_epochs=2))
    #       e0    e1    e2    e3(halve, ref 1->3)  e4    e5(halve, 3->5)  e6 best  e7   e8(halve 6->8) e9   e10(halve)
    assert np.allclose(lrs[1:], [0.01, 0.01, 0.01, 0.005, 0.005, 0.0025, 0.0025, 0.0025, 0.00125, 0.00125, 0.000625])
    assert U.should_stop(10, 1e-7, best, 1e-6, 10)        # below min_learning_rate
    assert not U.should_stop(10, 1e-3, best, 1e-6, 10)
    best.min_loss_epoch = 0
    assert U.should_stop(10, 1e-3, best, 1e-6, 10)        # early stop


def test_epoch_ledger_side_car_files_resume_and_fixed_schedule(tmp_path):
    """EpochLedger = the bookkeeping of nnet/lib/train.py: same LR / stop decisions as the table above, the side-car files in
    the reference's formats (train.py:121-131), a continued run picks up rates and the best epoch from them, and a schedule
    file fixes the rates in advance (train.py:52-58)."""
    model = tmp_path / "nnet"
    model.mkdir()
    cfg = {"learning_rate": 0.01, "num_epochs": 12, "reduce_lr_epochs": 2, "num_steps_per_epoch": 100}
    (tmp_path / "c.json").write_text(json.dumps(cfg))
    params = U.Params(str(tmp_path / "c.json"))
    led = U.EpochLedger(str(model), params, 0)
    assert params.early_stop_epochs == 10 and params.min_learning_rate == 1e-5       # defaults inserted like train.py:101-104
    led.write_feature_dim(30)
    assert (model / "feature_dim").read_text() == "30\n"
    losses = [3.0, 2.5, 2.6, 2.7, 2.8]
    for epoch, loss in enumerate(losses):
        assert led.rate(epoch) == [0.01, 0.01, 0.01, 0.01, 0.005][epoch]
        assert not led.close_epoch(epoch, loss, 0.1)
    assert (model / "learning_rate").read_text().splitlines() == ["0 0.01000000", "1 0.01000000", "2 0.01000000", "3 0.01000000",
                                                                  "4 0.00500000", "5 0.00500000"]
    assert (model / "valid_loss").read_text().splitlines()[1] == "1 2.500000 0.100000"
    # continue after epoch 4 (first_epoch = 5): rates and best epoch come back from the files
    led2 = U.EpochLedger(str(model), params, 5)
    assert led2.rate(5) == 0.005 and led2.best.min_loss == 2.5 and led2.best.min_loss_epoch == 1
    with pytest.raises(AssertionError):
        U.EpochLedger(str(model), params, 3)             # learning_rate file does not match the resume point
    # stop once the rate would fall below min_learning_rate
    params.dict["min_learning_rate"] = 0.004
    assert led2.close_epoch(5, 2.9, 0.1) and led2.rate(6) == 0.0025
    # a rank that did not evaluate adopts the decision
    led3 = U.EpochLedger(str(tmp_path / "other"), params, 0) if (tmp_path / "other").mkdir() is None else None
    led3.adopt(0, 0.02)
    assert led3.rate(1) == 0.02
    # fixed schedule file: no tuning, never stops early
    sched = tmp_path / "lr.txt"
    sched.write_text("\n".join(str(0.1 / (i + 1)) for i in range(13)) + "\n")
    params.dict["learning_rate"] = str(sched)
    led4 = U.EpochLedger(str(tmp_path / "other"), params, 0)
    assert led4.fixed_schedule and led4.rate(3) == 0.1 / 4
    assert not led4.close_epoch(0, 9.9, 0.5) and led4.rate(1) == 0.05
    # the step the checkpoint index points at
    U.write_checkpoint_state(str(model), "model-700", ["model-600", "model-700"])
    assert U.checkpoint_step(str(model)) == 700 and U.checkpoint_step(str(tmp_path / "other")) is None


def test_chunk_split_and_weighted_average():
    """extract.py:69-93 on synthetic lengths: T=25001, chunk=10000 -> 5 chunks."""
    chunks = U.split_into_chunks(25001, 10000)
    assert chunks == [(0, 10000), (5000, 10000), (10000, 10000), (15000, 10000), (20000, 5001)]
    assert U.split_into_chunks(10000, 10000) == [(0, 10000)]
    assert U.split_into_chunks(10001, 10000) == [(0, 10000), (5000, 5001)]
    e = np.array([[1.0, 0.0], [0.0, 2.0]], np.float32)
    assert np.allclose(U.average_chunk_embeddings(e, [10000, 5000], False), [2 / 3, 2 / 3])
    assert np.allclose(U.average_chunk_embeddings(e, [10000, 5000], True), [2 / 3, 1 / 3])
    # the per-utterance driver of extract.py: one piece up to chunk_size, else full-length chunks as one batch + the shorter tail
    calls = []

    def predict(x):       # "embedding" = (first frame's first value, number of frames); batch in -> batch out
        x = np.asarray(x)
        calls.append(x.shape)
        return np.array([x[0, 0], x.shape[0]], np.float32) if x.ndim == 2 else np.stack([[c[0, 0], c.shape[0]] for c in x]).astype(np.float32)

    feat = np.arange(130, dtype=np.float32)[:, None] * np.ones((1, 3), np.float32)
    emb, pieces = U.utterance_embedding(predict, feat[:60], 60, False)
    assert pieces == 1 and calls == [(60, 3)] and np.allclose(emb, [0, 60])
    del calls[:]
    emb, pieces = U.utterance_embedding(predict, feat, 60, False)        # chunks (0,60) (30,60) (60,60) (90,40)
    assert pieces == 4 and calls == [(3, 60, 3), (40, 3)]
    assert np.allclose(emb, [(0 * 60 + 30 * 60 + 60 * 60 + 90 * 40) / 220.0, (3 * 60 * 60 + 40 * 40) / 220.0])
    emb, _ = U.utterance_embedding(predict, feat, 60, True)
    assert abs(np.linalg.norm(emb) - 1.0) < 1e-6


def test_prefetch_iter_keeps_order_and_forwards_errors():
    assert list(U.prefetch_iter(iter(range(100)), depth=3)) == list(range(100))
    assert list(U.prefetch_iter(iter([]))) == []

    def broken():
        yield 1
        yield 2
        raise ValueError("truncated ark")

    it = U.prefetch_iter(broken(), depth=2)
    assert next(it) == 1 and next(it) == 2
    with pytest.raises(ValueError, match="truncated ark"):
        next(it)
    it = U.prefetch_iter(iter(range(10 ** 9)), depth=2)      # abandoning the iterator stops the producer
    assert next(it) == 0
    it.close()


def test_cos_pairwise_eer():
    rs = np.random.RandomState(0)
    centres = rs.randn(10, 16) * 3
    emb = np.concatenate([c + rs.randn(8, 16) for c in centres])
    labels = np.repeat(np.arange(10), 8)
    eer = U.compute_cos_pairwise_eer(emb.copy(), labels)
    assert 0.0 <= eer < 0.1
    rand_eer = U.compute_cos_pairwise_eer(rs.randn(80, 16), labels)
    assert 0.35 < rand_eer < 0.65
    # subsampling path (> max_num_embeddings) keeps working with an integer stride (py2 '/' in the reference)
    assert 0.0 <= U.compute_cos_pairwise_eer(emb.copy(), labels, max_num_embeddings=30) < 0.2


def test_every_shipped_single_task_config_builds_an_engine_config():
    """All 81 single-task nnet_conf/*.json of the reference (voxceleb, sre, fisher recipes) map onto an engine configuration:
    4 loss families, statistics / self-attention pooling (key types 0-3), ring / MHE auxiliary losses, sgd / momentum.
    Needs the reference checkout (build container only); the 10 multitask configs (no network_type) are out of scope."""
    import glob
    ref = "/root/reference/egs"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present")
    from tf_kaldi_speaker_amd.misc.utils import Params
    from tf_kaldi_speaker_amd.model.tdnn import engine_config
    files = sorted(glob.glob(os.path.join(ref, "*", "*", "nnet_conf", "*.json")))
    assert len(files) >= 90
    n = 0
    for f in files:
        p = Params(f)
        if "network_type" not in p.dict:
            continue
        cfg = engine_config(p, 30, 100, p.dict["loss_func"], 64, 400)
        assert cfg.feat_dim == 30 and cfg.num_speakers == 100
        n += 1
    assert n == 81


def test_sample_validset_spk2utt_rules(tmp_path):
    """misc/tools/sample_validset_spk2utt.py (run.sh:179): speakers with >= n+2 utterances first, n utterances each, one
    utterance of every speaker always left for training, output in spk2utt format."""
    from tf_kaldi_speaker_amd.misc.tools.sample_validset_spk2utt import main, read_spk2utt, sample_validset
    import io
    import random
    from contextlib import redirect_stdout
    lines = ["spk%02d %s" % (i, " ".join("spk%02d-u%d" % (i, j) for j in range(n))) for i, n in enumerate([12, 9, 7, 3, 2, 1, 15, 5])]
    p = tmp_path / "spk2utt"
    p.write_text("\n".join(lines) + "\n")
    speakers = read_spk2utt(str(p))
    assert len(speakers) == 8 and speakers[3] == ("spk03", ["spk03-u0", "spk03-u1", "spk03-u2"])
    all_utts = dict(speakers)
    for seed in range(20):
        got = sample_validset(speakers, 4, 5, random.Random(seed))
        assert len(got) == 4 and len({s for s, _ in got}) == 4
        for spk, utts in got:
            assert len(all_utts[spk]) >= 7          # four speakers have >= 5 + 2 utterances: only those are drawn
            assert len(utts) == 5 and len(set(utts)) == 5 and set(utts) <= set(all_utts[spk])
    # more held-out speakers than rich ones: topped up with small speakers, which keep one utterance for training
    got = dict(sample_validset(speakers, 6, 5, random.Random(1)))
    assert len(got) == 6 and {"spk00", "spk01", "spk02", "spk06"} <= set(got)
    for spk, utts in got.items():
        n = len(all_utts[spk])
        assert len(utts) == (5 if n > 5 else n - 1)
    with pytest.raises(ValueError):
        sample_validset(speakers, 9, 5, random.Random(0))
    buf = io.StringIO()
    with redirect_stdout(buf):
        main(["2", "3", str(p), "--seed", "7"])
    out = [l.split() for l in buf.getvalue().strip().splitlines()]
    assert len(out) == 2 and all(len(l) == 4 and all(u.startswith(l[0] + "-") for u in l[1:]) for l in out)
    buf2 = io.StringIO()
    with redirect_stdout(buf2):
        main(["2", "3", str(p), "--seed", "7"])
    assert buf2.getvalue() == buf.getvalue()


def test_length_batch_planner_and_batched_utterance_embeddings():
    """Batched extraction, host side: utterances sorted into padded batches within the chunk / row / fill limits, every index once;
    the windowed driver gives the embeddings of the one-at-a-time driver (extract.py:64-93) in archive order."""
    rs = np.random.RandomState(0)
    lengths = [int(v) for v in rs.randint(25, 2001, 300)] + [10000, 9000, 26]
    plan = U.plan_length_batches(lengths, 49152, 128)
    seen = sorted(i for idx, _ in plan for i in idx)
    assert seen == list(range(len(lengths)))
    for idx, t in plan:
        assert t == max(lengths[i] for i in idx) and len(idx) <= 128 and (len(idx) == 1 or len(idx) * t <= 49152)
        short = [k for k, i in enumerate(idx) if lengths[i] < 0.9 * t]      # members in sorted order: padding beyond 10 % only while the batch is small
        assert all(k * t < 8192 for k in short), (t, short)
    padded = sum(len(idx) * t for idx, t in plan)
    assert padded <= 1.08 * sum(lengths), "more than 8 %% padding on a uniform length mix: %d vs %d" % (padded, sum(lengths))
    assert U.plan_length_batches([], 100, 4) == [] and U.plan_length_batches([7], 3, 4) == [([0], 7)]      # an utterance longer than the row budget goes alone

    def predict(x):       # "embedding" = (first value, frames) per matrix
        x = np.asarray(x)
        return np.array([x[0, 0], x.shape[0]], np.float32) if x.ndim == 2 else np.stack([[c[0, 0], c.shape[0]] for c in x]).astype(np.float32)

    calls = []

    def predict_batch(items):
        calls.append([it.shape[0] for it in items])
        return np.stack([predict(it.decode() if hasattr(it, "decode") else it) for it in items])

    feats = [np.arange(n, dtype=np.float32)[:, None] * np.ones((1, 3), np.float32) + k for k, n in enumerate((60, 130, 40, 61))]
    for normalize in (False, True):
        del calls[:]
        got = U.batched_utterance_embeddings(predict_batch, feats, 60, normalize)
        assert len(calls) == 1 and calls[0] == [60, 60, 60, 60, 40, 40, 60, 31]      # 1 + 4 + 1 + 2 pieces in one call
        for f, (emb, pieces) in zip(feats, got):
            ref, ref_pieces = U.utterance_embedding(predict, f, 60, normalize)
            assert pieces == ref_pieces and np.allclose(emb, ref, rtol=1e-6, atol=0)
    assert U.batched_utterance_embeddings(predict_batch, [], 60, False) == []


def test_packed_matrix_reader(tmp_path):
    """read_mat_ark_packed: 'CM ' matrices undecoded (the image the GPU decodes), other formats decoded; PackedMatrix.decode /
    row_range agree bit for bit with the (reference-pinned) host reader."""
    rs = np.random.RandomState(1)
    mats = {"a": rs.randn(77, 30).astype(np.float32) * 3, "b": rs.randn(5, 30).astype(np.float32), "c": rs.randn(300, 23).astype(np.float32),
            "d": rs.randn(40, 30).astype(np.float32)}
    ark = str(tmp_path / "mixed.ark")
    with open(ark, "wb") as f:
        kaldi_io.write_compressed_mat(f, mats["a"], key="a")
        kaldi_io.write_mat(f, mats["b"], key="b")                       # 'FM '
        kaldi_io.write_compressed_mat(f, mats["c"], key="c")
        kaldi_io.write_compressed_mat(f, mats["d"], key="d")
    ref = dict(kaldi_io.read_mat_ark(ark))
    got = list(kaldi_io.read_mat_ark_packed(ark))
    assert [k for k, _ in got] == ["a", "b", "c", "d"]
    for block in (7, 64, 1000, 5000):          # records that straddle the reader's blocks (also through a pipe: no readinto, short reads)
        for src in (ark, "cat %s |" % ark):
            again = list(kaldi_io.read_mat_ark_packed(src, block_bytes=block))
            assert [k for k, _ in again] == ["a", "b", "c", "d"]
            for (k, m), (_, m0) in zip(again, got):
                assert np.array_equal(m.decode() if isinstance(m, kaldi_io.PackedMatrix) else m, m0.decode() if isinstance(m0, kaldi_io.PackedMatrix) else m0), (block, k)
    for k, m in got:
        if k == "b":
            assert isinstance(m, np.ndarray) and np.array_equal(m, ref[k])
            continue
        assert isinstance(m, kaldi_io.PackedMatrix) and m.shape == ref[k].shape
        assert np.array_equal(m.decode(), ref[k])
        s, n = 3, min(17, m.rows - 3)
        part = m.row_range(s, n)
        assert part.shape == (n, m.cols) and np.array_equal(part.decode(), ref[k][s:s + n])
    with open(ark, "rb") as f:      # a truncated archive is an error, not a short matrix
        data = f.read()
    bad = str(tmp_path / "cut.ark")
    open(bad, "wb").write(data[:200])
    with pytest.raises(Exception):
        list(kaldi_io.read_mat_ark_packed(bad))


def _host_golden():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "host_golden.npz"))
    return g, json.loads(str(g["text_cases"]))


def test_cos_pairwise_eer_matches_the_reference_function():
    """misc/utils.py:compute_cos_pairwise_eer against the reference's own function (its source executed by tests/golden/make_host_golden.py:
    utils.py:273-312 with py2 integer division): separated and overlapping speakers, the subsampling branch (integer strides 2 and 1, a small
    cap), two speakers.  ROC + interp1d + brentq on the same scores give the same root: 1e-9."""
    g, cases = _host_golden()
    assert len(cases["eer"]) == len(g["eer"]) >= 7
    for i, c in enumerate(cases["eer"]):
        emb, labels = g["emb_%d" % i].astype(np.float64), g["labels_%d" % i]
        keep = emb.copy()
        got = U.compute_cos_pairwise_eer(emb, labels, max_num_embeddings=c["max_num_embeddings"])
        assert abs(got - float(g["eer"][i])) <= 1e-9, (i, c, got, float(g["eer"][i]))
        assert abs(float(g["eer"][i]) - c["eer"]) == 0.0
        assert np.array_equal(emb, keep)          # (the reference normalises its argument in place; callers here keep their embeddings)


def test_lr_and_valid_loss_files_parse_as_the_reference_parses_them(tmp_path):
    """load_lr / load_valid_loss (utils.py:193-214) on the files train.py appends to (`<epoch> <lr>`, `<epoch> <loss> <eer>`): ties keep the FIRST
    epoch of the minimum (strict <)."""
    _, cases = _host_golden()
    for c in cases["load_lr"]:
        p = tmp_path / "learning_rate"
        p.write_text(c["text"])
        assert U.load_lr(str(p)) == c["values"]
    for c in cases["load_valid_loss"]:
        p = tmp_path / "valid_loss"
        p.write_text(c["text"])
        v = U.load_valid_loss(str(p))
        assert (v.min_loss, v.min_loss_epoch) == (c["min_loss"], c["min_loss_epoch"])


def test_small_dict_helpers_match_the_reference_functions():
    """substring_in_list (:315-330), remove_params_prefix (:349-358: a NEW ParamsPlain, argument untouched), add_dict_prefix (:361-366)."""
    _, cases = _host_golden()
    for c in cases["substring_in_list"]:
        assert U.substring_in_list(c["s"], c["list"]) == c["result"], c
    for c in cases["remove_params_prefix"]:
        p = U.ParamsPlain()
        p.dict.update(c["dict"])
        q = U.remove_params_prefix(p, c["prefix"])
        assert q.dict == c["result"] and p.dict == c["input_after"] and q is not p, c
    for c in cases["add_dict_prefix"]:
        assert U.add_dict_prefix(dict(c["dict"]), c["prefix"]) == c["result"]


def test_sample_validset_script_draws_what_the_reference_script_draws(tmp_path):
    """misc/tools/sample_validset_spk2utt.py against the reference's script itself (run by tests/golden/make_host_golden.py with the interpreter's
    generator seeded): same pools, same order of draws -> the same speakers and utterances, byte for byte, for the same seed."""
    from tf_kaldi_speaker_amd.misc.tools.sample_validset_spk2utt import main
    import io
    from contextlib import redirect_stdout
    _, cases = _host_golden()
    assert len(cases["sample_validset"]) >= 5
    for c in cases["sample_validset"]:
        p = tmp_path / "spk2utt"
        p.write_text(c["spk2utt"])
        buf = io.StringIO()
        with redirect_stdout(buf):
            main([str(c["num_spks"]), str(c["num_utts"]), str(p), "--seed", str(c["seed"])])
        assert buf.getvalue() == c["stdout"], c


def test_get_speaker_info_matches_the_reference_function(tmp_path):
    """dataset/data_loader.py:get_speaker_info against the reference's function (:14-55, source executed by tests/golden/make_host_golden.py) on
    a directory whose spklist names a speaker the directory does not hold (a validation subset).  The reference keeps feats.scp's line end
    inside every "utt path:offset" string (its reader strips it later); here the strings are clean - compared modulo that newline."""
    from tf_kaldi_speaker_amd.dataset.data_loader import get_speaker_info
    _, cases = _host_golden()
    c = cases["speaker_info"]
    for name, text in c["files"].items():
        (tmp_path / name).write_text(text)
    s2f, f2s, s2i = get_speaker_info(str(tmp_path), str(tmp_path / "spklist"))
    assert s2i == c["spk2index"]
    assert {str(k): v for k, v in s2f.items()} == {k: [x.rstrip("\n") for x in v] for k, v in c["spk2features"].items()}
    assert f2s == {k.rstrip("\n"): v for k, v in c["features2spk"].items()}
