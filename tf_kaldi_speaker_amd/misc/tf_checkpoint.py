"""TensorFlow "V2" checkpoints (`model-<step>.index` + `model-<step>.data-00000-of-00001`) read and written WITHOUT TensorFlow.

The reference saves and restores its models with `tf.train.Saver` (reference model/trainer.py:318,444; the pretrained models of its
README.md:86-104 are such checkpoints).  The payload format is small and public, so the drop-in keeps it readable:

  * `<prefix>.index` is an SSTable in the LevelDB table format (tensorflow/core/lib/io/table*.cc is a copy of LevelDB's table code):
    data blocks of prefix-compressed (key, value) entries + a restart array, each block followed by a 1-byte compression type
    (0 = none; the tensor-bundle writer never compresses) and a masked CRC32C; a metaindex block, an index block whose values
    are BlockHandles (varint offset, varint size) and a 48-byte footer ending in the magic 0xdb4775248b80fb57.
  * key "" -> BundleHeaderProto {1: num_shards, 2: endianness, 3: VersionDef}; every other key is a variable name ->
    BundleEntryProto {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset, 5: size, 6: crc32c (fixed32, masked), 7: slices}
    (tensorflow/core/protobuf/tensor_bundle.proto).
  * `<prefix>.data-XXXXX-of-YYYYY`: the raw little-endian tensor bytes at [offset, offset + size).

PINNING STATUS: restated from the published formats above; the CRC32C implementation is pinned by the RFC 3720 known-answer vectors and
the reader / writer by round trips and a hand-assembled table (tests/test_tf_checkpoint.py).  No TensorFlow-written file exists in the
build environment, so compatibility with a real `tf.train.Saver` file is UNPINNED here; `tests/golden/make_tf_golden.py` (run on a box
that has TF 1.x) drops a small Saver-written checkpoint next to its vectors and tests/test_tf_checkpoint.py checks this reader against
it when the file is present.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_, 17: np.uint16,
           19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}

# ---------------------------------------------------------------- CRC32C (Castagnoli), table driven; masked as LevelDB stores it
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t[i] = c
        _CRC_TABLE = t
    return _CRC_TABLE


def _native_crc():
    """xvio_crc32c of libxvector_io.so (host C++, slice-by-8) when the library is built; None otherwise."""
    try:
        try:
            from ..dataset import native_loader
        except (ImportError, ValueError):      # drop-in layout: PYTHONPATH=$TF_KALDI_ROOT
            from dataset import native_loader
        return native_loader.load().xvio_crc32c
    except Exception:
        return None


def crc32c(data, crc=0, force_python=False):
    """CRC32C of a bytes-like object continuing from `crc`.  Uses the native routine when libxvector_io.so is there (a 40 MB
    checkpoint in ~30 ms); the byte loop below is the portable restatement (about 1 s per 3 MB) and the one the known-answer
    tests pin first."""
    buf = bytes(data)
    fn = None if force_python else _native_crc()
    if fn is not None:
        return int(fn(crc & 0xFFFFFFFF, buf, len(buf)))
    t = _crc_table()
    c = (~crc) & 0xFFFFFFFF
    tl = t.tolist()
    for b in buf:
        c = tl[(c ^ b) & 0xFF] ^ (c >> 8)
    return (~c) & 0xFFFFFFFF


def mask_crc(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def unmask_crc(m):
    r = (m - 0xa282ead8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------- varints / minimal protobuf
def _get_varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """-> list of (field number, wire type, value): varint -> int, 64-bit / 32-bit -> raw bytes, length-delimited -> bytes."""
    out, pos, n = [], 0, len(buf)
    while pos < n:
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v, pos = bytes(buf[pos:pos + ln]), pos + ln
        elif wt == 5:
            v, pos = bytes(buf[pos:pos + 4]), pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((field, wt, v))
    return out


def _field(field, wt, payload):
    key = _put_varint((field << 3) | wt)
    if wt == 0:
        return key + _put_varint(payload)
    if wt == 2:
        return key + _put_varint(len(payload)) + payload
    return key + payload


def _parse_shape(buf):
    dims = []
    for f, wt, v in _parse_proto(buf):
        if f == 2 and wt == 2:                      # Dim {1: size}
            size = 0
            for g, gwt, gv in _parse_proto(v):
                if g == 1 and gwt == 0:
                    size = gv if gv < (1 << 63) else gv - (1 << 64)
            dims.append(size)
        elif f == 3 and wt == 0 and v:
            raise ValueError("tensor of unknown rank in a checkpoint")
    return tuple(dims)


def _encode_shape(shape):
    return b"".join(_field(2, 2, _field(1, 0, int(d))) for d in shape)


# ---------------------------------------------------------------- LevelDB table
def _read_block(f, offset, size, verify):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise ValueError("truncated table block at %d" % offset)
    body, ctype, crc = raw[:size], raw[size], struct.unpack("<I", raw[size + 1:])[0]
    if verify and unmask_crc(crc) != crc32c(raw[:size + 1]):
        raise ValueError("table block checksum mismatch at %d" % offset)
    if ctype != 0:
        raise ValueError("compressed table block (type %d): tensor-bundle index files are written uncompressed" % ctype)
    return body


def _block_entries(body):
    n_restarts = struct.unpack("<I", body[-4:])[0]
    limit = len(body) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _get_varint(body, pos)
        non_shared, pos = _get_varint(body, pos)
        vlen, pos = _get_varint(body, pos)
        key = key[:shared] + bytes(body[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(body[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """-> list of (key bytes, value bytes) of a LevelDB-format table file, in key order."""
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        size = f.tell()
        if size < 48:
            raise ValueError("%s: too short for a table file" % path)
        f.seek(size - 48)
        footer = f.read(48)
        if struct.unpack("<Q", footer[40:])[0] != TABLE_MAGIC:
            raise ValueError("%s: not a TensorFlow V2 checkpoint index (bad table magic)" % path)
        pos = 0
        _, pos = _get_varint(footer, pos)      # metaindex handle
        _, pos = _get_varint(footer, pos)
        ioff, pos = _get_varint(footer, pos)
        isize, pos = _get_varint(footer, pos)
        out = []
        for _, handle in _block_entries(_read_block(f, ioff, isize, verify)):
            boff, hp = _get_varint(handle, 0)
            bsize, hp = _get_varint(handle, hp)
            out.extend(_block_entries(_read_block(f, boff, bsize, verify)))
        return out


def _build_block(entries, restart_interval=16):
    body, restarts, prev, count = bytearray(), [], b"", 0
    for key, value in entries:
        shared = 0
        if count % restart_interval == 0:
            restarts.append(len(body))
        else:
            m = min(len(prev), len(key))
            while shared < m and prev[shared] == key[shared]:
                shared += 1
        body += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        prev, count = key, count + 1
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def write_table(path, items, block_bytes=4096):
    """items: (key bytes, value bytes) in strictly increasing key order."""
    with open(path, "wb") as f:
        handles = []

        def emit(body):
            off = f.tell()
            trailer = body + b"\x00"
            f.write(trailer + struct.pack("<I", mask_crc(crc32c(trailer))))
            return off, len(body)

        cur, cur_bytes, last_key = [], 0, None
        for key, value in items:
            if last_key is not None and key <= last_key:
                raise ValueError("table keys must be strictly increasing")
            cur.append((key, value))
            cur_bytes += len(key) + len(value) + 3
            last_key = key
            if cur_bytes >= block_bytes:
                handles.append((last_key,) + emit(_build_block(cur)))
                cur, cur_bytes = [], 0
        if cur or not handles:
            handles.append((last_key if last_key is not None else b"",) + emit(_build_block(cur)))
        meta = emit(_build_block([]))
        index = emit(_build_block([(k, _put_varint(off) + _put_varint(sz)) for k, off, sz in handles], restart_interval=1))
        footer = _put_varint(meta[0]) + _put_varint(meta[1]) + _put_varint(index[0]) + _put_varint(index[1])
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))


# ---------------------------------------------------------------- tensor bundle
def _shard_name(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def list_variables(prefix, verify=True):
    """-> {name: (dtype, shape, shard, offset, size, masked crc)} of the checkpoint `<prefix>.index`, and the shard count."""
    entries, num_shards, skipped = {}, 1, []
    for key, value in read_table(prefix + ".index", verify):
        fields = _parse_proto(value)
        if key == b"":
            for f, wt, v in fields:
                if f == 1 and wt == 0:
                    num_shards = v
                if f == 2 and wt == 0 and v != 0:
                    raise ValueError("big-endian checkpoints are not supported")
            continue
        dtype, shape, shard, offset, size, crc, sliced = 0, (), 0, 0, 0, None, False
        for f, wt, v in fields:
            if f == 1 and wt == 0:
                dtype = v
            elif f == 2 and wt == 2:
                shape = _parse_shape(v)
            elif f == 3 and wt == 0:
                shard = v
            elif f == 4 and wt == 0:
                offset = v
            elif f == 5 and wt == 0:
                size = v
            elif f == 6 and wt == 5:
                crc = struct.unpack("<I", v)[0]
            elif f == 7:
                sliced = True
        if sliced:
            raise ValueError("%s: partitioned variables are not supported" % key.decode())
        if dtype not in _DTYPES:
            # not a numeric tensor - e.g. the DT_STRING entry `_CHECKPOINTABLE_OBJECT_GRAPH` an object-based saver adds: no model
            # variable can be one, so it is skipped here and only a REQUESTED name that is missing is an error
            skipped.append((key.decode("utf-8", "replace"), dtype))
            continue
        entries[key.decode("utf-8")] = (np.dtype(_DTYPES[dtype]), shape, shard, offset, size, crc)
    list_variables.skipped = skipped
    return entries, num_shards


def read_checkpoint(prefix, verify=False):
    """-> {variable name: ndarray} of the TF V2 checkpoint `<prefix>` (e.g. `<model>/nnet/model-1200000`).  verify: also check the
    CRC32C of every tensor (slow in pure Python; the index blocks are always checked)."""
    entries, num_shards = list_variables(prefix)
    out, files = {}, {}
    try:
        for name, (dtype, shape, shard, offset, size, crc) in entries.items():
            if shard not in files:
                files[shard] = open(_shard_name(prefix, shard, num_shards), "rb")
            f = files[shard]
            f.seek(offset)
            raw = f.read(size)
            want = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize if shape else dtype.itemsize
            if len(raw) != size or size != want:
                raise ValueError("%s: %d bytes in the data file, %d expected for %s%s" % (name, len(raw), want, dtype, list(shape)))
            if verify and crc is not None and unmask_crc(crc) != crc32c(raw):
                raise ValueError("%s: tensor checksum mismatch" % name)
            out[name] = np.frombuffer(raw, dtype=dtype).reshape(shape).copy()
    finally:
        for f in files.values():
            f.close()
    return out


def write_checkpoint(prefix, variables, with_crc=True):
    """Write {name: ndarray} as a single-shard TF V2 checkpoint `<prefix>.index` / `<prefix>.data-00000-of-00001` that
    `tf.train.Saver().restore` / `tf.train.load_checkpoint` read (TF verifies the CRCs, hence with_crc)."""
    names = sorted(variables, key=lambda s: s.encode("utf-8"))
    items = [(b"", _field(1, 0, 1) + _field(2, 0, 0) + _field(3, 2, _field(1, 0, 1)))]       # 1 shard, little endian, producer 1
    offset = 0
    # both files are written under temporary names and moved into place (data first, the index last): a reader never sees a
    # half-written checkpoint, as with the .npz payload written beside it
    data_path, index_path = _shard_name(prefix, 0, 1), prefix + ".index"
    with open(data_path + ".tmp", "wb") as f:
        for name in names:
            a = np.asarray(variables[name], order="C")      # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DTYPE_CODES:
                raise ValueError("%s: dtype %s cannot be stored" % (name, a.dtype))
            raw = a.tobytes()
            f.write(raw)
            entry = _field(1, 0, _DTYPE_CODES[a.dtype]) + _field(2, 2, _encode_shape(a.shape)) + _field(3, 0, 0) + _field(4, 0, offset) + \
                _field(5, 0, len(raw)) + _field(6, 5, struct.pack("<I", mask_crc(crc32c(raw)) if with_crc else 0))
            items.append((name.encode("utf-8"), entry))
            offset += len(raw)
    write_table(index_path + ".tmp", items)
    os.replace(data_path + ".tmp", data_path)
    os.replace(index_path + ".tmp", index_path)


# optimiser slots and bookkeeping a Saver stores beside the model variables (trainer.py:332-346: GradientDescent / Momentum / Adam
# under name="optimizer"); they are not variables of the graph tdnn.py / loss.py build
def is_model_variable(name):
    tail = name.rsplit("/", 1)[-1]
    return not (tail in ("Momentum", "Adam", "Adam_1", "optimizer", "optimizer_1") or "/optimizer" in name or
                name in ("beta1_power", "beta2_power", "global_step") or name.startswith("optimizer/"))
