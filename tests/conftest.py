import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """The libraries and the C host are build products (git-ignored): a fresh checkout builds them once, as
    __graft_entry__.build() does, instead of failing every test that loads them."""
    import shutil
    need = [os.path.join(ROOT, "tf_kaldi_speaker_amd", "libxvector_hip.so"), os.path.join(ROOT, "tf_kaldi_speaker_amd", "libxvector_io.so"),
            os.path.join(ROOT, "tests", "c_abi", "engine_step")]
    if all(os.path.isfile(p) for p in need):
        return
    if shutil.which("hipcc") and shutil.which("make"):
        import __graft_entry__
        __graft_entry__.build()
