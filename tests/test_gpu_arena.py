"""The arena of a predict-only engine: no loss head, no backward pass - no dz slot per layer."""
import pytest

pytestmark = pytest.mark.gpu


def test_predict_only_engine_keeps_the_two_slot_arena():
    """an engine without a loss head never runs a backward pass - it must not pay for a dz slot per layer (Trainer.predict_batch
    builds one with 49 152 rows: nine slots of 296 MB against two)."""
    from tf_kaldi_speaker_amd import engine as E
    rows = 49152
    pred = E.Engine(E.make_config(30, 0, max_batch=128, max_frames=4000, max_rows=rows), device="cuda:0")
    train = E.Engine(E.make_config(30, 100, max_batch=128, max_frames=4000, max_rows=rows), device="cuda:0")
    slot = rows * 1504 * 4
    assert train.arena_bytes - pred.arena_bytes > 5 * slot, (train.arena_bytes, pred.arena_bytes)
    assert pred.arena_bytes < 4e9, pred.arena_bytes
    pred.close(); train.close()
