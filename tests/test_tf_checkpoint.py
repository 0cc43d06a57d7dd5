"""The TensorFlow V2 checkpoint codec of the drop-in (tf_kaldi_speaker_amd/misc/tf_checkpoint.py; reference: tf.train.Saver at
model/trainer.py:318,444).  Pinned here by: the RFC 3720 CRC32C known answers, LevelDB's documented mask, a table assembled by hand
from the LevelDB format description (independent of the writer), round trips, and - when tests/golden/tf_golden_ckpt.* written by a real
tf.train.Saver is present (tests/golden/make_tf_golden.py, needs TF 1.x, cannot run in the build container) - that file."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from tf_kaldi_speaker_amd.misc import tf_checkpoint as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("force_python", [True, False])
def test_crc32c_known_answers(force_python):
    # RFC 3720 appendix B.4
    assert T.crc32c(bytes(32), force_python=force_python) == 0x8A9136AA
    assert T.crc32c(bytes([0xFF] * 32), force_python=force_python) == 0x62A8AB43
    assert T.crc32c(bytes(range(32)), force_python=force_python) == 0x46DD794E
    assert T.crc32c(bytes(range(31, -1, -1)), force_python=force_python) == 0x113FDB5C
    assert T.crc32c(b"123456789", force_python=force_python) == 0xE3069283
    # continuation: crc(a + b) == crc(b, crc(a)), odd split (the native routine's 8-byte slices + tail)
    data = bytes((i * 131 + 7) & 0xFF for i in range(1003))
    assert T.crc32c(data[11:], T.crc32c(data[:11], force_python=force_python), force_python=force_python) == T.crc32c(data, force_python=True)


def test_crc_mask_is_leveldbs():
    # leveldb/util/crc32c.h: Mask(crc) = ((crc >> 15) | (crc << 17)) + 0xa282ead8
    for c in (0, 1, 0xE3069283, 0xFFFFFFFF, 0x12345678):
        assert T.mask_crc(c) == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF
        assert T.unmask_crc(T.mask_crc(c)) == c
    assert T.mask_crc(T.crc32c(b"foo")) != T.crc32c(b"foo")


def _hand_block(entries):
    """One table block assembled straight from leveldb/doc/table_format.md, every entry a restart point (shared = 0)."""
    body, restarts = b"", []
    for k, v in entries:
        restarts.append(len(body))
        body += bytes([0, len(k), len(v)]) + k + v          # all three lengths < 128: one-byte varints
    for r in restarts or [0]:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts or [0]))
    return body


def test_reads_a_hand_assembled_table(tmp_path):
    """Reader against bytes built here from the format description, not by write_table."""
    shape = b"\x12\x02\x08\x02" + b"\x12\x02\x08\x03"                                   # TensorShapeProto: dim{size:2} dim{size:3}
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    raw = a.tobytes()
    entry = b"\x08\x01" + b"\x12" + bytes([len(shape)]) + shape + b"\x18\x00" + b"\x20\x00" + b"\x28" + bytes([len(raw)]) + \
        b"\x35" + struct.pack("<I", T.mask_crc(T.crc32c(raw)))                          # dtype 1, shape, shard 0, offset 0, size, crc
    header = b"\x08\x01\x10\x00"                                                       # num_shards 1, little endian
    data_block = _hand_block([(b"", header), (b"tdnn/w", entry)])
    blob, handles = b"", []
    for body in (data_block, _hand_block([])):
        handles.append((len(blob), len(body)))
        t = body + b"\x00"
        blob += t + struct.pack("<I", T.mask_crc(T.crc32c(t)))
    index_block = _hand_block([(b"tdnn/x", bytes([handles[0][0], handles[0][1]]))])     # separator key >= last key of the block
    ioff = len(blob)
    t = index_block + b"\x00"
    blob += t + struct.pack("<I", T.mask_crc(T.crc32c(t)))
    footer = bytes([handles[1][0], handles[1][1], ioff, len(index_block)])
    blob += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    prefix = str(tmp_path / "model-7")
    open(prefix + ".index", "wb").write(blob)
    open(prefix + ".data-00000-of-00001", "wb").write(raw)
    got = T.read_checkpoint(prefix, verify=True)
    assert list(got) == ["tdnn/w"] and got["tdnn/w"].dtype == np.float32 and np.array_equal(got["tdnn/w"], a)
    # a flipped payload byte is caught by the tensor CRC, a flipped index byte by the block CRC
    open(prefix + ".data-00000-of-00001", "wb").write(raw[:5] + bytes([raw[5] ^ 1]) + raw[6:])
    with pytest.raises(ValueError, match="checksum"):
        T.read_checkpoint(prefix, verify=True)
    bad = bytearray(blob)
    bad[3] ^= 1
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="checksum"):
        T.read_checkpoint(prefix)


def test_round_trip_of_the_models_variables(tmp_path):
    rs = np.random.RandomState(0)
    variables = {"tdnn/tdnn1_conv/kernel": rs.randn(1, 5, 30, 512).astype(np.float32), "tdnn/tdnn1_conv/bias": rs.randn(512).astype(np.float32),
                 "tdnn/tdnn1_bn/moving_variance": rs.rand(512).astype(np.float32), "softmax/output/kernel": rs.randn(512, 77).astype(np.float32),
                 "global_step": np.array(1234, np.int64), "tdnn/tdnn1_conv/kernel/Momentum": rs.randn(1, 5, 30, 512).astype(np.float32)}
    for i in range(300):                      # enough keys for several table blocks and shared key prefixes
        variables["tdnn/extra_%03d/gamma" % i] = rs.randn(3).astype(np.float32)
    prefix = str(tmp_path / "model-42")
    T.write_checkpoint(prefix, variables)
    entries, shards = T.list_variables(prefix)
    assert shards == 1 and set(entries) == set(variables)
    assert entries["global_step"][1] == () and entries["tdnn/tdnn1_conv/kernel"][1] == (1, 5, 30, 512)
    got = T.read_checkpoint(prefix, verify=True)
    assert set(got) == set(variables)
    for k, v in variables.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    keys = [k for k, _ in T.read_table(prefix + ".index")]
    assert keys == sorted(keys) and keys[0] == b""
    assert not T.is_model_variable("tdnn/tdnn1_conv/kernel/Momentum") and not T.is_model_variable("global_step")
    assert T.is_model_variable("tdnn/tdnn1_conv/kernel") and T.is_model_variable("softmax/output/kernel")


def test_converter_tool_both_directions(tmp_path):
    rs = np.random.RandomState(1)
    variables = {"tdnn/tdnn6_dense/kernel": rs.randn(3000, 512).astype(np.float32), "tdnn/tdnn6_dense/kernel/Momentum": rs.randn(3000, 512).astype(np.float32),
                 "beta1_power": np.array(0.9, np.float32)}
    prefix = str(tmp_path / "model-9")
    T.write_checkpoint(prefix, variables)
    tool = os.path.join(ROOT, "tools", "tf_ckpt_to_npz.py")
    out = subprocess.run([sys.executable, tool, prefix, "--verify"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    data = np.load(prefix + ".npz")
    assert data.files == ["tdnn/tdnn6_dense/kernel"] and np.array_equal(data["tdnn/tdnn6_dense/kernel"], variables["tdnn/tdnn6_dense/kernel"])
    os.remove(prefix + ".index")
    os.remove(prefix + ".data-00000-of-00001")
    out = subprocess.run([sys.executable, tool, "--to-tf", prefix], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    back = T.read_checkpoint(prefix, verify=True)
    assert list(back) == ["tdnn/tdnn6_dense/kernel"] and np.array_equal(back["tdnn/tdnn6_dense/kernel"], variables["tdnn/tdnn6_dense/kernel"])
    out = subprocess.run([sys.executable, tool, "--list", prefix], capture_output=True, text=True)
    assert out.returncode == 0 and "tdnn/tdnn6_dense/kernel" in out.stdout and "[3000, 512]" in out.stdout


@pytest.mark.skipif(not os.path.isfile(os.path.join(GOLDEN, "tf_golden_ckpt.index")),
                    reason="tests/golden/tf_golden_ckpt.* is written by make_tf_golden.py on a box with TensorFlow 1.x (not available in the build container)")
def test_reads_a_checkpoint_written_by_tensorflow():
    got = T.read_checkpoint(os.path.join(GOLDEN, "tf_golden_ckpt"), verify=True)
    want = np.load(os.path.join(GOLDEN, "tf_golden_ckpt_values.npz"))
    assert set(want.files) <= set(got)
    for k in want.files:
        assert got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), k


def test_non_numeric_entries_are_skipped_and_writes_are_atomic(tmp_path):
    """An object-based saver adds the DT_STRING entry `_CHECKPOINTABLE_OBJECT_GRAPH` to the index: it is no model variable, and a
    checkpoint that holds it must still load.  write_checkpoint leaves no temporary file and never a lone index."""
    prefix = str(tmp_path / "model-7")
    want = {"tdnn/tdnn1_conv/bias": np.arange(5, dtype=np.float32), "global_step": np.array(7, np.int64)}
    T.write_checkpoint(prefix, want)
    assert sorted(os.listdir(tmp_path)) == ["model-7.data-00000-of-00001", "model-7.index"]
    items = [(k, v) for k, v in T.read_table(prefix + ".index")]
    size = os.path.getsize(prefix + ".data-00000-of-00001")
    blob = b"\x08\x01graph"                                             # what a string tensor's bytes might look like
    with open(prefix + ".data-00000-of-00001", "ab") as f:
        f.write(blob)
    string_entry = T._field(1, 0, 7) + T._field(2, 2, T._encode_shape(())) + T._field(3, 0, 0) + T._field(4, 0, size) + T._field(5, 0, len(blob)) + \
        T._field(6, 5, struct.pack("<I", T.mask_crc(T.crc32c(blob))))       # dtype 7 = DT_STRING
    items.append((b"_CHECKPOINTABLE_OBJECT_GRAPH", string_entry))
    T.write_table(prefix + ".index", sorted(items))
    entries, _ = T.list_variables(prefix)
    assert sorted(entries) == sorted(want) and T.list_variables.skipped == [("_CHECKPOINTABLE_OBJECT_GRAPH", 7)]
    got = T.read_checkpoint(prefix, verify=True)
    assert sorted(got) == sorted(want) and all(np.array_equal(got[k], want[k]) for k in want)
