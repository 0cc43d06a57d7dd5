"""ctypes binding of libxvector_io.so (include/xvector_io.h): the native Kaldi minibatch loader.

NativeRandomQueue has the interface of KaldiDataRandomQueue (reference dataset/data_loader.py:310-414:
set_batch / set_length / start / fetch / stop) but decodes with C++ threads into caller-owned (pinned) host
buffers instead of pickling NumPy arrays through a multiprocessing.Queue - the path the reference's README
calls its bottleneck.  The sampling rules are the reference's (data_loader.py:271-298); batch i is a pure
function of (seed, i).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_HERE, "libxvector_io.so")


class XvioError(RuntimeError):
    pass


class XvioConfig(C.Structure):
    """Mirror of `struct xvio_config` (include/xvector_io.h)."""
    _fields_ = [
        ("data_dir", C.c_char_p),
        ("spklist", C.c_char_p),
        ("num_speakers", C.c_int32),
        ("num_segments", C.c_int32),
        ("min_len", C.c_int32),
        ("max_len", C.c_int32),
        ("shuffle", C.c_int32),
        ("num_threads", C.c_int32),
        ("queue_depth", C.c_int32),
        ("seed", C.c_uint64),
        ("packed", C.c_int32),
    ]


_VP = C.c_void_p
_I = C.c_int
_I64 = C.c_int64

SIGNATURES = {
    "xvio_last_error": (C.c_char_p, []),
    "xvio_abi_version": (_I, []),
    "xvio_crc32c": (C.c_uint32, [C.c_uint32, _VP, C.c_uint64]),
    "xvio_loader_create": (_I, [_VP, _VP]),
    "xvio_loader_destroy": (None, [_VP]),
    "xvio_loader_dim": (_I, [_VP]),
    "xvio_loader_total_speakers": (_I, [_VP]),
    "xvio_loader_num_utterances": (_I, [_VP]),
    "xvio_loader_next": (_I, [_VP, _VP, _VP, _VP]),
    "xvio_loader_stats": (_I, [_VP, _VP, _VP]),
    "xvio_packed_chunk_bytes": (_I64, [C.c_int32, C.c_int32]),
    "xvio_loader_next_packed": (_I, [_VP, _VP, _VP, _VP]),
    "xvio_read_rows": (_I, [C.c_char_p, _I64, C.c_int32, C.c_int32, _VP, _I64, _VP, _VP]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise XvioError("%s is missing: run `make -C tf_kaldi_speaker_amd/csrc`" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        raise XvioError("%s: %s" % (what, load().xvio_last_error().decode(errors="replace")))


def read_rows(ark_path, offset, start=0, length=-1, max_elems=1 << 24):
    """Rows [start, start+length) of the matrix at ark_path:offset as float32 [rows, cols] (CM: only those rows are decoded)."""
    lib = load()
    out = np.empty(max_elems, np.float32)
    rows, cols = C.c_int32(), C.c_int32()
    _check(lib.xvio_read_rows(ark_path.encode(), offset, start, length, out.ctypes.data, out.size, C.byref(rows), C.byref(cols)),
           "xvio_read_rows")
    return out[:rows.value * cols.value].reshape(rows.value, cols.value).copy()


def packed_chunk_bytes(dim, frames):
    return int(load().xvio_packed_chunk_bytes(int(dim), int(frames)))


def decode_packed(packed, batch, frames, dim):
    """NumPy restatement of xv_cm_decode (the GPU kernel) on a packed batch: [batch, frames, dim] float32, the arithmetic of the
    reference codec (kaldi_io.py:768-812) in float32 - what the host-decoding loader delivers for the same (seed, index)."""
    stride = packed_chunk_bytes(dim, frames)
    raw = np.frombuffer(np.ascontiguousarray(packed, np.uint8)[:batch * stride].tobytes(), np.uint8).reshape(batch, stride)
    out = np.empty((batch, frames, dim), np.float32)
    f32 = np.float32
    for i in range(batch):
        minv, rng = np.frombuffer(raw[i, :8].tobytes(), np.float32)
        hdr = np.frombuffer(raw[i, 8:8 + 8 * dim].tobytes(), np.uint16).reshape(dim, 4).astype(np.float32)
        b = raw[i, 8 + 8 * dim:8 + 8 * dim + dim * frames].reshape(dim, frames)
        gs = f32(rng) * f32(1.52590218966964e-05)
        p0, p25, p75, p100 = (f32(minv) + gs * hdr[:, j] for j in range(4))
        s_lo, s_mid, s_hi = (p25 - p0) / f32(64.0), (p75 - p25) / f32(128.0), (p100 - p75) / f32(63.0)
        v = b.astype(np.float32)
        y = np.where(b <= 64, p0[:, None] + s_lo[:, None] * v,
                     np.where(b <= 192, p25[:, None] + s_mid[:, None] * (v - f32(64.0)), p75[:, None] + s_hi[:, None] * (v - f32(192.0))))
        out[i] = y.astype(np.float32).T
    return out


class NativeRandomQueue(object):
    """Endless stream of random (features [B,T,D] f32, labels [B] i32) batches from C++ decoder threads.
    packed=True: the threads only gather the rows' undecoded 'CM ' bytes (a quarter of the traffic); `device_batches` then decodes on
    the GPU (xv_cm_decode), `fetch` through the NumPy restatement - same values either way."""

    def __init__(self, data_dir, spklist, num_parallel=1, max_qsize=10, num_speakers=None, num_segments=None,
                 min_len=None, max_len=None, shuffle=True, seed=None, packed=False):
        self.data = data_dir
        self.spklist = spklist
        self.num_speakers = num_speakers
        self.num_segments = num_segments
        self.min_len = min_len
        self.max_len = max_len
        self.num_parallel_datasets = num_parallel
        self.max_qsize = max_qsize
        self.shuffle = shuffle
        self.seed = int.from_bytes(os.urandom(8), "little") if seed is None else int(seed)
        self.packed = bool(packed)
        self.h = None
        self.dim = None
        self.num_total_speakers = None
        self._feat = self._lab = None

    def set_batch(self, num_speakers, num_segments):
        self.num_speakers = num_speakers
        self.num_segments = num_segments

    def set_length(self, min_len, max_len):
        self.min_len = min_len
        self.max_len = max_len

    def start(self):
        lib = load()
        cfg = XvioConfig(self.data.encode(), self.spklist.encode(), int(self.num_speakers), int(self.num_segments), int(self.min_len),
                         int(self.max_len), 1 if self.shuffle else 0, max(1, int(self.num_parallel_datasets)), max(1, int(self.max_qsize)),
                         self.seed & 0xFFFFFFFFFFFFFFFF, 1 if self.packed else 0)
        h = C.c_void_p()
        _check(lib.xvio_loader_create(C.byref(cfg), C.byref(h)), "xvio_loader_create")
        self.h = h
        self.dim = lib.xvio_loader_dim(h)
        self.num_total_speakers = lib.xvio_loader_total_speakers(h)
        self.batch = int(self.num_speakers) * int(self.num_segments)
        self._alloc()

    def _alloc(self):
        """Staging buffers the decoder copies into: pinned when torch + a GPU are around (async H2D), plain otherwise."""
        shape = (self.batch, int(self.max_len), self.dim)
        self._pinned = None
        if self.packed:
            self._pk = np.empty(self.batch * packed_chunk_bytes(self.dim, self.max_len), np.uint8)
            self._lab = np.empty(self.batch, np.int32)
            return
        try:
            import torch
            if torch.cuda.is_available():
                self._pinned = (torch.empty(shape, dtype=torch.float32).pin_memory(), torch.empty(self.batch, dtype=torch.int32).pin_memory())
                self._feat, self._lab = self._pinned[0].numpy(), self._pinned[1].numpy()
                return
        except ImportError:
            pass
        self._feat, self._lab = np.empty(shape, np.float32), np.empty(self.batch, np.int32)

    def fetch_into(self, features, labels):
        """Decode the next batch straight into caller buffers (float32 >= B*max_len*dim, int32 >= B); returns T."""
        frames = C.c_int32()
        _check(load().xvio_loader_next(self.h, features.ctypes.data, labels.ctypes.data, C.byref(frames)), "xvio_loader_next")
        return frames.value

    def fetch_packed_into(self, packed, labels):
        """Packed mode: the next batch's undecoded bytes into caller buffers (uint8 >= B*packed_chunk_bytes(dim, max_len), int32 >= B); returns T."""
        frames = C.c_int32()
        _check(load().xvio_loader_next_packed(self.h, packed.ctypes.data, labels.ctypes.data, C.byref(frames)), "xvio_loader_next_packed")
        return frames.value

    def fetch(self):
        """(features [B,T,D], labels [B]) as fresh NumPy arrays, like the reference queue."""
        if self.packed:
            t = self.fetch_packed_into(self._pk, self._lab)
            return decode_packed(self._pk, self.batch, t, self.dim), self._lab.copy()
        t = self.fetch_into(self._feat.reshape(-1), self._lab)
        feats = self._feat.reshape(-1)[:self.batch * t * self.dim].reshape(self.batch, t, self.dim).copy()
        return feats, self._lab.copy()

    def device_batches(self, device, depth=3, max_ahead=4):
        """Generator of (features, labels) as DEVICE tensors: the decoder threads fill one of `depth` pinned staging buffers,
        the host-to-device copy is enqueued on its own stream and the consumer's stream waits for it - decode, PCIe transfer
        and the previous training step overlap (the engine's calls only enqueue).  The host is kept at most `max_ahead` batches
        ahead of the consumer's stream: every batch is a fresh device tensor, and an unthrottled host (0.4 ms to enqueue a 6 ms
        step) would otherwise hold dozens of them in flight."""
        import collections
        import torch
        dev = torch.device(device)
        consumed = collections.deque()
        copy_stream = torch.cuda.Stream(device=dev)
        shape = (self.batch, int(self.max_len), self.dim)
        if self.packed:
            from .. import ops
            shape = (self.batch * packed_chunk_bytes(self.dim, self.max_len),)
        ring = [(torch.empty(shape, dtype=torch.uint8 if self.packed else torch.float32).pin_memory(),
                 torch.empty(self.batch, dtype=torch.int32).pin_memory(), torch.cuda.Event()) for _ in range(depth)]
        i = 0
        while True:
            pf, pl, ev = ring[i % depth]
            ev.synchronize()                       # the copy that last read this staging buffer has finished
            if self.packed:
                t = self.fetch_packed_into(pf.numpy(), pl.numpy())
            else:
                t = self.fetch_into(pf.numpy().reshape(-1), pl.numpy())
            with torch.cuda.stream(copy_stream):
                if self.packed:       # a quarter of the PCIe bytes; decoded by xv_cm_decode on the copy stream, off the step's critical path
                    raw = pf[:self.batch * packed_chunk_bytes(self.dim, t)].to(dev, non_blocking=True)
                    x = ops.cm_decode(raw, self.batch, t, self.dim)
                    raw.record_stream(copy_stream)
                else:
                    x = pf.view(-1)[:self.batch * t * self.dim].view(self.batch, t, self.dim).to(dev, non_blocking=True)
                y = pl.to(dev, non_blocking=True)
                ev.record(copy_stream)
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)
            x.record_stream(cur)
            y.record_stream(cur)
            yield x, y
            # back in the generator = the consumer has enqueued its work on batch i: mark that point on its stream
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
            consumed.append(done)
            if len(consumed) > max_ahead:
                consumed.popleft().synchronize()
            i += 1

    def stats(self):
        n, sec = C.c_int64(), C.c_double()
        _check(load().xvio_loader_stats(self.h, C.byref(n), C.byref(sec)), "xvio_loader_stats")
        return n.value, sec.value

    def stop(self):
        if self.h is not None:
            load().xvio_loader_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.stop()
        except Exception:
            pass
