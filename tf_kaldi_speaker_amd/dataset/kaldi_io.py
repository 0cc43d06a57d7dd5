"""Kaldi ark/scp I/O for the x-vector engine - the I/O boundary north_star keeps.

Written from the Kaldi on-disk formats (kaldi/src/matrix/compressed-matrix.h, kaldi-io docs) and
the behaviour of the reference's dataset/kaldi_io.py (cited per function).  NumPy only - this module
must stay importable in loader subprocesses without touching torch / HIP.

Formats handled
  keys           "<utt-id> " followed by the object
  binary marker  "\\0B"
  float vector   "FV " \\x04 <int32 dim> <dim x f32>          (reference write_vec_flt, kaldi_io.py:624-653)
  float matrix   "FM " \\x04 <int32 rows> \\x04 <int32 cols> <rows*cols x f32>   (also "DM " f64)
  compressed     "CM " <f32 min> <f32 range> <int32 rows> <int32 cols>
                 cols x {uint16 p0,p25,p75,p100}  then cols*rows uint8, COLUMN-major
                 (reference _read_compressed_mat, kaldi_io.py:768-812; the only training format,
                 kaldi_io.py:743-749)
  text matrix    " [ ... ]"
"""
import gzip
import os
import random
import re
import struct
import subprocess
import sys

import numpy as np


class UnsupportedDataType(Exception):
    pass


class UnknownVectorHeader(Exception):
    pass


class UnknownMatrixHeader(Exception):
    pass


class BadSampleSize(Exception):
    pass


class BadInputFormat(Exception):
    pass


class SubprocessFailed(Exception):
    pass


# ------------------------------------------------------------------------------------------
# opening
# ------------------------------------------------------------------------------------------
_PREFIX = re.compile(r"^(ark|scp)(,scp|,b|,t|,n?f|,n?p|,b?o|,n?s|,n?cs)*:")


class _PipeReader(object):
    """stdout of `cmd` as a readable binary stream; close() waits and checks the exit status
    (reference popen(), kaldi_io.py:377-410)."""

    def __init__(self, cmd):
        self.cmd = cmd
        self.proc = subprocess.Popen(cmd, shell=True, stdout=subprocess.PIPE)
        self.stream = self.proc.stdout

    def read(self, n=-1):
        return self.stream.read(n)

    def readline(self):
        return self.stream.readline()

    def __iter__(self):
        return iter(self.stream)

    def close(self):
        self.stream.close()
        ret = self.proc.wait()
        if ret != 0:
            raise SubprocessFailed("cmd %s returned %d !" % (self.cmd, ret))


class _PipeWriter(object):
    mode = "wb"

    def __init__(self, cmd):
        self.cmd = cmd
        self.proc = subprocess.Popen(cmd, shell=True, stdin=subprocess.PIPE)
        self.stream = self.proc.stdin

    def write(self, b):
        return self.stream.write(b)

    def flush(self):
        self.stream.flush()

    def close(self):
        self.stream.close()
        ret = self.proc.wait()
        if ret != 0:
            raise SubprocessFailed("cmd %s returned %d !" % (self.cmd, ret))


def open_or_fd(file, mode="rb"):
    """Open a Kaldi rxfilename / wxfilename: optional "ark:"/"scp:" prefix, ":offset" suffix,
    "cmd |" input pipe, "| cmd" output pipe, ".gz", plain file, or an already-open stream
    (reference open_or_fd, kaldi_io.py:344-374)."""
    if not isinstance(file, str):
        return file
    offset = None
    if _PREFIX.search(file):
        file = file.split(":", 1)[1]
    if re.search(r":[0-9]+$", file):
        file, offset = file.rsplit(":", 1)
    file = file.strip()
    if file.endswith("|"):
        fd = _PipeReader(file[:-1])
    elif file.startswith("|"):
        fd = _PipeWriter(file[1:])
    elif file.endswith(".gz"):
        fd = gzip.open(file, mode)
    else:
        fd = open(file, mode)
    if offset is not None:
        fd.seek(int(offset))
    return fd


def read_key(fd):
    """Utterance key up to the first space; '' / None at EOF (reference read_key, kaldi_io.py:413-425)."""
    key = b""
    peek = getattr(fd, "peek", None)
    if peek is not None:          # buffered files: one look ahead instead of one read() per character
        buf = peek(256)
        pos = buf.find(b" ")
        if pos >= 0:
            key = fd.read(pos + 1)[:-1]
            key = key.decode("latin1").strip()
            if key == "":
                return None
            assert re.match(r"^\S+$", key) is not None
            return key
    while True:
        ch = fd.read(1)
        if ch == b"":
            break
        if ch == b" ":
            break
        key += ch
    key = key.decode("latin1").strip()
    if key == "":
        return None
    assert re.match(r"^\S+$", key) is not None
    return key


# ------------------------------------------------------------------------------------------
# vectors
# ------------------------------------------------------------------------------------------
def _read_vec_flt_binary(fd):
    header = fd.read(3).decode()
    if header == "FV ":
        dt = np.float32
    elif header == "DV ":
        dt = np.float64
    else:
        raise UnknownVectorHeader("The header contained '%s'" % header)
    assert fd.read(1) == b"\x04"
    n = struct.unpack("<i", fd.read(4))[0]
    return np.frombuffer(fd.read(n * np.dtype(dt).itemsize), dtype=dt).copy()


def read_vec_flt(file_or_fd):
    fd = open_or_fd(file_or_fd)
    try:
        binary = fd.read(2)
        if binary == b"\0B":
            return _read_vec_flt_binary(fd)
        arr = (binary + fd.readline()).decode().strip().split()
        arr = [a for a in arr if a not in ("[", "]")]
        return np.array(arr, dtype=np.float32)
    finally:
        if fd is not file_or_fd:
            fd.close()


def read_vec_flt_ark(file_or_fd):
    fd = open_or_fd(file_or_fd)
    try:
        key = read_key(fd)
        while key:
            yield key, read_vec_flt(fd)
            key = read_key(fd)
    finally:
        if fd is not file_or_fd:
            fd.close()


def write_vec_flt(file_or_fd, v, key=""):
    """"<key> " + "\\0B" + "FV " + \\x04 + int32 dim + data   (reference kaldi_io.py:624-653)."""
    fd = open_or_fd(file_or_fd, mode="wb")
    try:
        if key != "":
            fd.write((key + " ").encode("latin1"))
        fd.write(b"\0B")
        if v.dtype == np.float32:
            fd.write(b"FV ")
        elif v.dtype == np.float64:
            fd.write(b"DV ")
        else:
            raise UnsupportedDataType("'%s', please use 'float32' or 'float64'" % v.dtype)
        fd.write(b"\x04")
        fd.write(struct.pack("<I", v.shape[0]))
        fd.write(np.ascontiguousarray(v).tobytes())
    finally:
        if fd is not file_or_fd:
            fd.close()


# ------------------------------------------------------------------------------------------
# matrices
# ------------------------------------------------------------------------------------------
_GLOBAL_HDR = np.dtype([("minvalue", "<f4"), ("range", "<f4"), ("num_rows", "<i4"), ("num_cols", "<i4")])
_COL_HDR = np.dtype([("p0", "<u2"), ("p25", "<u2"), ("p75", "<u2"), ("p100", "<u2")])
_U16 = np.float32(1.52590218966964e-05)   # 1/65535, the constant of compressed-matrix.h


def _col_percentiles(col_headers, gmin, grange):
    """uint16 percentiles -> float32 (compressed-matrix.h Uint16ToFloat)."""
    out = np.empty((len(col_headers), 4), np.float32)
    for j, name in enumerate(("p0", "p25", "p75", "p100")):
        out[:, j] = np.float32(gmin) + np.float32(grange) * _U16 * col_headers[name].astype(np.float32)
    return out


def _decode_cm(data, pct):
    """data: uint8 [cols, n] (column-major block), pct: [cols, 4] -> float32 [n, cols].
    Piecewise-linear map of compressed-matrix.h CharToFloat: 0..64 | 65..192 | 193..255."""
    v = data.astype(np.float32)
    p0, p25, p75, p100 = (pct[:, j:j + 1] for j in range(4))
    lo = p0 + (p25 - p0) / np.float32(64.0) * v
    mid = p25 + (p75 - p25) / np.float32(128.0) * (v - np.float32(64.0))
    hi = p75 + (p100 - p75) / np.float32(63.0) * (v - np.float32(192.0))
    out = np.where(data <= 64, lo, np.where(data <= 192, mid, hi)).astype(np.float32)
    return np.ascontiguousarray(out.T)


def _read_cm_header(fd, fmt):
    assert fmt == "CM ", "The formats CM2, CM3 are not supported..."
    gmin, grange, rows, cols = np.frombuffer(fd.read(16), dtype=_GLOBAL_HDR, count=1)[0]
    return float(gmin), float(grange), int(rows), int(cols)


def _read_compressed_mat(fd, fmt):
    gmin, grange, rows, cols = _read_cm_header(fd, fmt)
    col_headers = np.frombuffer(fd.read(cols * 8), dtype=_COL_HDR, count=cols)
    data = np.frombuffer(fd.read(cols * rows), dtype=np.uint8, count=cols * rows).reshape(cols, rows)
    return _decode_cm(data, _col_percentiles(col_headers, gmin, grange))


def _read_compressed_submat(fd, fmt, start, length):
    """Rows [start, start+length) of a 'CM ' matrix without reading the rest
    (reference _read_compressed_submat, kaldi_io.py:814-867)."""
    gmin, grange, rows, cols = _read_cm_header(fd, fmt)
    assert rows >= (start + length), "The number of frames is not enough for length %d" % length
    col_headers = np.frombuffer(fd.read(cols * 8), dtype=_COL_HDR, count=cols)
    data = np.empty((cols, length), np.uint8)
    base = fd.tell()
    for i in range(cols):
        fd.seek(base + i * rows + start)
        data[i] = np.frombuffer(fd.read(length), dtype=np.uint8, count=length)
    fd.seek(base + cols * rows)
    return _decode_cm(data, _col_percentiles(col_headers, gmin, grange))


def _read_mat_binary(fd):
    header = fd.read(3).decode()
    if header.startswith("CM"):
        return _read_compressed_mat(fd, header)
    if header == "FM ":
        dt = np.float32
    elif header == "DM ":
        dt = np.float64
    else:
        raise UnknownMatrixHeader("The header contained '%s'" % header)
    s1, rows, s2, cols = np.frombuffer(fd.read(10), dtype="int8,int32,int8,int32", count=1)[0]
    buf = fd.read(int(rows) * int(cols) * np.dtype(dt).itemsize)
    return np.frombuffer(buf, dtype=dt).reshape(int(rows), int(cols)).copy()


def _read_submat_binary(fd, start, length):
    header = fd.read(3).decode()
    if header.startswith("CM"):
        return _read_compressed_submat(fd, header, start, length)
    raise ValueError("The features should be in the compressed format.")   # reference kaldi_io.py:743-749


def _read_mat_ascii(fd):
    rows = []
    while True:
        line = fd.readline().decode()
        if len(line) == 0:
            raise BadInputFormat
        if len(line.strip()) == 0:
            continue
        arr = line.strip().split()
        if arr[-1] != "]":
            rows.append(np.array(arr, dtype="float32"))
        else:
            rows.append(np.array(arr[:-1], dtype="float32"))
            return np.vstack(rows)


def read_mat(file_or_fd):
    fd = open_or_fd(file_or_fd)
    try:
        binary = fd.read(2)
        if binary == b"\0B":
            return _read_mat_binary(fd)
        assert binary == b" ["
        return _read_mat_ascii(fd)
    finally:
        if fd is not file_or_fd:
            fd.close()


def read_mat_ark(file_or_fd):
    """generator of (key, matrix) from an ark file / pipe (reference kaldi_io.py:683-703)."""
    fd = open_or_fd(file_or_fd)
    try:
        key = read_key(fd)
        while key:
            yield key, read_mat(fd)
            key = read_key(fd)
    finally:
        if fd is not file_or_fd:
            fd.close()


class PackedMatrix(object):
    """A Kaldi 'CM ' matrix kept exactly as it sits in the archive - [min f32][range f32][rows i32][cols i32][cols x (p0, p25, p75, p100)
    u16][cols x rows u8, column after column] - which is the image xv_cm_decode_ragged (include/xvector_hip.h) decodes on the GPU: the
    extraction driver ships a quarter of the fp32 bytes and the host never runs the codec.  `payload` is a bytes-like view (usually a
    slice of the block the reader pulled from the archive: no copy per utterance)."""
    __slots__ = ("rows", "cols", "payload", "block", "start")
    HEADER = 16

    def __init__(self, rows, cols, payload, block=None, start=0):
        """block / start: the bytes object `payload` is a view of and where it begins there (the reader's archive block) - lets a consumer
        ship whole blocks to the GPU once instead of gathering the matrices of a batch on the host."""
        self.rows, self.cols, self.payload = int(rows), int(cols), payload
        self.block, self.start = (payload, 0) if block is None else (block, int(start))
        assert len(payload) == self.HEADER + 8 * self.cols + self.cols * self.rows

    @property
    def shape(self):
        return (self.rows, self.cols)

    def decode(self):
        """float32 [rows, cols] through the host codec (the reference's arithmetic, _decode_cm)."""
        buf = np.frombuffer(self.payload, np.uint8)
        gmin, grange = np.frombuffer(buf[:8].tobytes(), "<f4")
        head = self.HEADER + 8 * self.cols
        hdr = np.frombuffer(buf[self.HEADER:head].tobytes(), dtype=_COL_HDR, count=self.cols)
        data = buf[head:].reshape(self.cols, self.rows)
        return _decode_cm(data, _col_percentiles(hdr, float(gmin), float(grange)))

    def row_range(self, start, length):
        """Rows [start, start + length) as a PackedMatrix of their own (same header: the codec is per element)."""
        assert 0 <= start and start + length <= self.rows
        buf = np.frombuffer(self.payload, np.uint8)
        head = self.HEADER + 8 * self.cols
        data = buf[head:].reshape(self.cols, self.rows)[:, start:start + length]
        return PackedMatrix(length, self.cols, buf[:8].tobytes() + struct.pack("<ii", length, self.cols) + buf[self.HEADER:head].tobytes() +
                            np.ascontiguousarray(data).tobytes())


def read_mat_ark_packed(file_or_fd, block_bytes=8 << 20):
    """read_mat_ark for the batched extraction driver: 'CM ' matrices come back undecoded as PackedMatrix, every other format as the
    float32 matrix read_mat gives.  The archive is pulled in blocks of `block_bytes` and the records are cut out of the block as views -
    a few microseconds of Python per utterance instead of a read() per field (this generator runs in the driver's prefetch thread, which
    shares the interpreter lock with the thread that feeds the GPU)."""
    fd = open_or_fd(file_or_fd)
    try:
        buf, pos, eof = b"", 0, False
        readinto = getattr(fd, "readinto", None)

        def need(n):      # make buf[pos : pos + n] available; False at end of input
            nonlocal buf, pos, eof
            while len(buf) - pos < n and not eof:
                # a new block = the unconsumed tail of the old one (a partial record, a few KB) + the next read, which lands in place
                # (readinto: no second copy of the 8 MB; the old block stays alive as long as records cut out of it do)
                rem = len(buf) - pos
                want = max(block_bytes, n - rem)
                new = bytearray(rem + want)
                new[:rem] = buf[pos:]
                if readinto is not None:
                    got = readinto(memoryview(new)[rem:]) or 0
                else:
                    more = fd.read(want)
                    got = len(more)
                    new[rem:rem + got] = more
                if got == 0:
                    eof = True
                    break
                if got < want:
                    del new[rem + got:]
                buf, pos = new, 0
            return len(buf) - pos >= n

        while True:
            # key: up to the first space
            while True:
                sp = buf.find(b" ", pos)
                if sp >= 0 or eof:
                    break
                need(len(buf) - pos + 1)
            if sp < 0:
                if buf[pos:].strip():
                    raise BadInputFormat
                return
            key = buf[pos:sp].decode("latin1").strip()
            pos = sp + 1
            if key == "":
                return
            assert re.match(r"^\S+$", key) is not None
            if not need(2):
                raise BadInputFormat
            if buf[pos:pos + 2] != b"\0B":
                assert buf[pos:pos + 2] == b" ["          # text archives: rare, read the slow way from here on
                rest = _Pushback(fd, buf[pos:])
                buf, pos, eof = b"", 0, True
                rest.read(2)
                yield key, _read_mat_ascii(rest)
                k = read_key(rest)
                while k:
                    yield k, read_mat(rest)
                    k = read_key(rest)
                return
            if not need(5):
                raise BadInputFormat
            fmt = buf[pos + 2:pos + 5]
            if fmt == b"CM ":
                if not need(5 + 16):
                    raise BadInputFormat
                rows, cols = struct.unpack_from("<ii", buf, pos + 5 + 8)
                size = PackedMatrix.HEADER + 8 * cols + cols * rows
                if not need(5 + size):
                    raise BadInputFormat
                yield key, PackedMatrix(rows, cols, memoryview(buf)[pos + 5:pos + 5 + size], buf, pos + 5)
                pos += 5 + size
            elif fmt in (b"FM ", b"DM "):
                if not need(5 + 10):
                    raise BadInputFormat
                _, rows, _, cols = struct.unpack_from("<bibi", buf, pos + 5)
                dt = np.float32 if fmt == b"FM " else np.float64
                size = rows * cols * np.dtype(dt).itemsize
                if not need(15 + size):
                    raise BadInputFormat
                yield key, np.frombuffer(buf, dtype=dt, count=rows * cols, offset=pos + 15).reshape(rows, cols).copy()
                pos += 15 + size
            else:
                raise UnknownMatrixHeader("The header contained '%s'" % fmt.decode("latin1"))
    finally:
        if fd is not file_or_fd:
            fd.close()


class _Pushback(object):
    """A stream with a few bytes put back in front of it (pipes cannot seek)."""

    def __init__(self, fd, head):
        self.fd, self.head = fd, bytes(head)

    def read(self, n=-1):
        if not self.head:
            return self.fd.read(n)
        if n < 0:
            out, self.head = self.head + self.fd.read(), b""
            return out
        out, self.head = self.head[:n], self.head[n:]
        if len(out) < n:
            out += self.fd.read(n - len(out))
        return out

    def readline(self):
        line = b""
        while True:
            ch = self.read(1)
            line += ch
            if ch in (b"", b"\n"):
                return line


def read_mat_scp(file_or_fd):
    fd = open_or_fd(file_or_fd)
    try:
        for line in fd:
            key, rxfile = line.decode().strip().split(" ", 1)
            yield key, read_mat(rxfile)
    finally:
        if fd is not file_or_fd:
            fd.close()


def write_mat(file_or_fd, m, key=""):
    """Uncompressed 'FM ' / 'DM ' matrix (reference write_mat, kaldi_io.py:870-905)."""
    fd = open_or_fd(file_or_fd, mode="wb")
    try:
        if key != "":
            fd.write((key + " ").encode("latin1"))
        fd.write(b"\0B")
        if m.dtype == np.float32:
            fd.write(b"FM ")
        elif m.dtype == np.float64:
            fd.write(b"DM ")
        else:
            raise UnsupportedDataType("'%s', please use 'float32' or 'float64'" % m.dtype)
        fd.write(b"\x04" + struct.pack("<I", m.shape[0]))
        fd.write(b"\x04" + struct.pack("<I", m.shape[1]))
        fd.write(np.ascontiguousarray(m).tobytes())
    finally:
        if fd is not file_or_fd:
            fd.close()


def write_compressed_mat(file_or_fd, m, key=""):
    """Kaldi 'CM ' (kSpeechFeature) writer.  The reference has no CM writer (it only reads what
    Kaldi's copy-feats --compress=true produced); this follows compressed-matrix.cc
    ComputeColHeader / FloatToChar so that synthetic training data can be produced without Kaldi."""
    m = np.asarray(m, np.float32)
    rows, cols = m.shape
    gmin, gmax = float(m.min()), float(m.max())
    if gmax == gmin:
        gmax = gmin + 1.0 + abs(gmin)
    grange = np.float32(gmax - gmin)
    gmin = np.float32(gmin)

    def to_u16(x):
        f = (x - gmin) / grange
        return np.clip(np.floor(f * 65535.0 + 0.499), 0, 65535).astype(np.uint16)

    srt = np.sort(m, axis=0)
    q = [0, rows // 4, (3 * rows) // 4, rows - 1] if rows >= 5 else None
    hdr = np.zeros(cols, _COL_HDR)
    if q is not None:
        p = [to_u16(srt[i]) for i in q]
    else:   # compressed-matrix.cc small-matrix branch: spread the available order statistics
        p0 = to_u16(srt[0])
        p25 = to_u16(srt[1]) if rows > 1 else p0 + 1
        p75 = to_u16(srt[2]) if rows > 2 else p25 + 1
        p100 = to_u16(srt[3]) if rows > 3 else p75 + 1
        p = [p0, p25, p75, p100]
    # enforce strictly increasing percentiles, as Kaldi does
    p0 = np.minimum(p[0].astype(np.int64), 65532)
    p25 = np.minimum(np.maximum(p[1].astype(np.int64), p0 + 1), 65533)
    p75 = np.minimum(np.maximum(p[2].astype(np.int64), p25 + 1), 65534)
    p100 = np.maximum(p[3].astype(np.int64), p75 + 1)
    hdr["p0"], hdr["p25"], hdr["p75"], hdr["p100"] = p0, p25, p75, p100
    pct = _col_percentiles(hdr, gmin, grange)
    v = m.T   # [cols, rows]
    f0, f25, f75, f100 = (pct[:, j:j + 1] for j in range(4))
    with np.errstate(divide="ignore", invalid="ignore"):
        lo = np.floor((v - f0) / (f25 - f0) * 64.0 + 0.5)
        mid = 64 + np.floor((v - f25) / (f75 - f25) * 128.0 + 0.5)
        hi = 192 + np.floor((v - f75) / (f100 - f75) * 63.0 + 0.5)
    lo = np.clip(lo, 0, 64)
    mid = np.clip(mid, 64, 192)
    hi = np.clip(hi, 192, 255)
    data = np.where(v < f25, lo, np.where(v < f75, mid, hi)).astype(np.uint8)
    fd = open_or_fd(file_or_fd, mode="wb")
    try:
        if key != "":
            fd.write((key + " ").encode("latin1"))
        fd.write(b"\0B" + b"CM ")
        fd.write(np.array([(gmin, grange, rows, cols)], dtype=_GLOBAL_HDR).tobytes())
        fd.write(hdr.tobytes())
        fd.write(np.ascontiguousarray(data).tobytes())
    finally:
        if fd is not file_or_fd:
            fd.close()


# ------------------------------------------------------------------------------------------
# FeatureReader - reference dataset/kaldi_io.py:40-150
# ------------------------------------------------------------------------------------------
class FeatureReader(object):
    """Random access to the matrices of a Kaldi data directory through feats.scp, keeping every
    ark open (one descriptor per archive).  `read_segment` decodes only the requested frames of a
    'CM ' matrix."""

    def __init__(self, data):
        self.fd = {}
        self.data = data
        self.utt2num_frames = {}
        path = os.path.join(data, "utt2num_frames")
        assert os.path.exists(path), "[Error] Expect utt2num_frames exists in %s " % data
        with open(path, "r") as f:
            for line in f:
                utt, length = line.strip().split(" ")
                self.utt2num_frames[utt] = int(length)
        self.dim = self.get_dim()

    def get_dim(self):
        with open(os.path.join(self.data, "feats.scp"), "r") as f:
            return self.read(f.readline().strip())[0].shape[1]

    def close(self):
        for name in self.fd:
            self.fd[name].close()
        self.fd = {}

    def _seek(self, file_or_fd):
        utt, rx = file_or_fd.split(" ")
        filename, offset = rx.rsplit(":", 1)
        if filename not in self.fd:
            self.fd[filename] = open(filename, "rb")
        fd = self.fd[filename]
        fd.seek(int(offset))
        return utt, fd

    def read(self, file_or_fd, length=None, shuffle=False, start=None):
        """(mat, start) - whole matrix, optionally cropped to `length` frames."""
        utt, fd = self._seek(file_or_fd)
        try:
            assert fd.read(2) == b"\0B"
            mat = _read_mat_binary(fd)
        except Exception:
            raise IOError("Cannot read features from %s" % file_or_fd)
        if length is not None:
            if start is None:
                n = mat.shape[0]
                length = n if length > n else length
                start = random.randint(0, n - length) if shuffle else 0
            else:
                assert not shuffle, "The start point is specified, thus shuffling is invalid."
            mat = mat[start:start + length, :]
        return mat, start

    def read_segment(self, file_or_fd, length=None, shuffle=False, start=None):
        """(mat, start) - only the requested frames are decoded (CM only)."""
        utt, fd = self._seek(file_or_fd)
        try:
            if fd.read(2) != b"\0B":
                raise IOError
            if length is not None:
                if start is None:
                    n = self.utt2num_frames[utt]
                    length = n if length > n else length
                    start = random.randint(0, n - length) if shuffle else 0
                else:
                    assert not shuffle, "The start point is specified, thus shuffling is invalid."
                mat = _read_submat_binary(fd, start, length)
            else:
                mat = _read_mat_binary(fd)
        except AssertionError:
            raise
        except Exception:
            raise IOError("Cannot read features from %s" % file_or_fd)
        return mat, start
