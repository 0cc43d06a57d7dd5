"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and
exports exactly what include/xvector_hip.h declares; the ctypes table matches the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "xvector_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(xv_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_entry_points():
    names = header_functions()
    assert len(names) >= 40
    for must in ("xv_affine_forward", "xv_affine_dgrad", "xv_affine_wgrad", "xv_stat_pool_forward",
                 "xv_margin_softmax_rows", "xv_engine_create", "xv_engine_forward", "xv_engine_backward"):
        assert must in names


def test_library_loads_and_exports_every_symbol():
    from tf_kaldi_speaker_amd import _lib
    assert os.path.isfile(_lib.LIB_PATH), "build the HIP extension first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(lib, name), "libxvector_hip.so does not export %s" % name


def test_ctypes_table_matches_header():
    from tf_kaldi_speaker_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()
    lib = _lib.load()
    assert lib.xv_abi_version() == 1
    assert lib.xv_device_count() >= 0   # 0 here: no compute call is made without a GPU


def test_config_struct_layout_matches_header():
    """Field order of struct xv_config in the header == ctypes mirror."""
    from tf_kaldi_speaker_amd import _lib
    src = open(HEADER).read()
    body = src[src.index("typedef struct xv_config {") + len("typedef struct xv_config {"):src.index("} xv_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for line in body.split(";"):
        m = re.match(r"\s*(int32_t|float)\s+(.+)", line.strip(), flags=re.S)
        if m:
            for nm in m.group(2).split(","):
                fields.append((m.group(1), nm.strip()))
    def mirror_field(n, t):
        if t is ctypes.c_int32:
            return ("int32_t", n)
        if t is ctypes.c_float:
            return ("float", n)
        assert issubclass(t, ctypes.Array) and t._type_ is ctypes.c_int32, (n, t)      # int32_t name[len]
        return ("int32_t", "%s[%d]" % (n, t._length_))
    mirror = [mirror_field(n, t) for n, t in _lib.XvConfig._fields_]
    assert fields == mirror
    assert _lib.XV_MAX_FRAME_LAYERS == 12 and "#define XV_MAX_FRAME_LAYERS 12" in src


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tf_kaldi_speaker_amd import engine
    with pytest.raises(engine.XvError):
        engine.Engine(engine.make_config(30, 10))
