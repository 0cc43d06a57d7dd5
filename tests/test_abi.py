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
    assert lib.xv_abi_version() == _lib.ABI_VERSION
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


def test_environment_switches_are_value_checked_and_unknown_ones_named():
    """the library reads a few documented switches; a value it does not understand makes engine creation fail with
    the name - before any GPU call, so this runs here.  Any other XV_* variable (a typo, a switch of an earlier round, another program's) is
    named once on stderr and ignored: it must not stop a run."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r)\n"
            "from tf_kaldi_speaker_amd import _lib\n"
            "lib = _lib.load(); cfg = _lib.XvConfig(); cfg.feat_dim = 30; h = C.c_void_p()\n"
            "rc = lib.xv_engine_create(C.byref(cfg), C.byref(h)); print(rc, lib.xv_last_error().decode())\n" % ROOT)
    for extra, want in (({"XV_NT_SCHED": "fast"}, "XV_NT_SCHED=fast: expected dp or sk"), ({"XV_CONV_WR": "3"}, "XV_CONV_WR=3: expected 2 or 4"),
                        ({"XV_SEGMENT_FUSED": "yes"}, "XV_SEGMENT_FUSED=yes: expected 0 or 1")):
        env = {k: v for k, v in os.environ.items() if not k.startswith("XV_")}
        env.update(extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.startswith("2 ") and want in out.stdout, (out.stdout, out.stderr[-500:])
    env = {k: v for k, v in os.environ.items() if not k.startswith("XV_")}
    env.update(XV_NT_SCHED="dp", XV_PRECISION="f32", XV_SHARE_GPU="1")          # documented names pass the table (the call then fails later: no GPU here)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "environment switch" not in out.stdout + out.stderr and "expected" not in out.stdout, (out.stdout, out.stderr[-500:])
    env.update(XV_TN_WGS="768")                                                 # not a name of this package: named on stderr, not a failure
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "ignoring unknown environment switch XV_TN_WGS" in out.stderr and "environment switch" not in out.stdout, (out.stdout, out.stderr[-500:])


def test_config_struct_size_is_checked():
    """xv_config.struct_bytes (ABI version 2): a host built against another header's xv_config is refused at xv_engine_create instead of
    having fields read from beyond its struct."""
    import ctypes as C
    from tf_kaldi_speaker_amd import _lib
    lib = _lib.load()
    assert lib.xv_abi_version() == _lib.ABI_VERSION == 3
    cfg = _lib.XvConfig()
    assert cfg.struct_bytes == C.sizeof(_lib.XvConfig)
    cfg.feat_dim = 30
    cfg.struct_bytes -= 4
    h = C.c_void_p()
    assert lib.xv_engine_create(C.byref(cfg), C.byref(h)) != 0
    assert "struct_bytes" in lib.xv_last_error().decode()
