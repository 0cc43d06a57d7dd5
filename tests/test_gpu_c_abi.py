"""The C-ABI driven from a plain C program (tests/c_abi/engine_step.c, built by csrc/Makefile with gcc -std=c99): one full
optimiser step on hipMalloc'ed buffers, no Python and no torch in the process.  It must reproduce, bit for bit, what the
Python host (ctypes over the same library) computes from the same variables, features and labels - the boundary is the
library, not the binding."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "c_abi", "engine_step")


def test_c_host_binary_is_built_and_prints_usage():
    """CPU-side check: the C translation unit compiled against include/xvector_hip.h and linked against the library."""
    assert os.path.isfile(BIN), "run __graft_entry__.build() (csrc/Makefile builds tests/c_abi/engine_step)"
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 1 and "usage:" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plain", "allreduce"])
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_c_host_step_matches_python_host_bit_for_bit(tmp_path, precision, mode):
    """mode "allreduce": the C host runs the staged backward and hands every finished gradient slice to xv_engine_allreduce over an
    RCCL communicator it created itself (one rank: the box has one GPU) - the data-parallel call sequence of a host without
    torch.distributed; a one-rank sum must reproduce the plain step bit for bit."""
    import torch
    from tf_kaldi_speaker_amd import _lib, engine as E
    D, N, B, T, step, lr = 30, 41, 6, 50, 1234, 0.05
    eng = E.Engine(E.make_config(D, N, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, max_batch=B, max_frames=T,
                                 precision=precision), device="cuda:0")
    eng.init_variables(seed=5)
    rs = np.random.RandomState(9)
    x = rs.randn(B, T, D).astype(np.float32)
    y = rs.randint(0, N, B).astype(np.int32)
    v0 = eng.variables.cpu().numpy().copy()
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(v0.tobytes()); f.write(x.tobytes()); f.write(y.tobytes())

    raw, reg = eng.train_step(x, y, lr, step, fetch_losses=True)
    emb = eng.endpoint("tdnn6_dense").cpu().numpy()
    v1 = eng.variables.cpu().numpy()
    eng.close()
    torch.cuda.synchronize()

    r = subprocess.run([BIN, fin, fout, str(D), str(N), str(_lib.LOSS_KINDS["additive_margin_softmax"]), "0.2", str(B), str(T),
                        str(_lib.PRECISIONS[precision]), repr(lr), str(step)] + (["allreduce"] if mode == "allreduce" else []),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    buf = open(fout, "rb").read()
    c_raw, c_reg = np.frombuffer(buf, np.float32, 2, 0)
    rows, cols = np.frombuffer(buf, np.int32, 2, 8)
    c_emb = np.frombuffer(buf, np.float32, rows * cols, 16).reshape(rows, cols)
    c_v1 = np.frombuffer(buf, np.float32, v0.size, 16 + 4 * rows * cols)
    assert (rows, cols) == emb.shape == (B, 512)
    assert np.float32(raw) == c_raw, (raw, c_raw)
    # the reported regularisation loss is a float atomicAdd over workgroups (order-dependent in the last bit); it feeds nothing
    assert abs(float(c_reg) - reg) <= 1e-6 * abs(reg), (reg, c_reg)
    assert np.array_equal(c_emb, emb)
    assert np.array_equal(c_v1, v1) and not np.array_equal(c_v1, v0)
