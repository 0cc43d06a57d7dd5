#!/usr/bin/env python3
"""The clock the chip holds INSIDE the forward GEMM of a full S1 training step on random data (DESIGN.md quoted both
a GRBM_GUI_ACTIVE-derived 2.13 GHz under --pmc and 2.38 GHz from stamps of an isolated launch).  Needs a diagnostics build of the
library (csrc/xv_diag.h; tools/variant.sh unit xv_gemm.hip "diag:-DXV_DIAG=1"):

    XV_LIB=build_variants/diag/libxvector_hip.so XV_DIAG_M=24576 XV_DIAG_N=512 XV_DIAG_K=2560 python3 tools/step_clock.py

Runs >= 2 s of back-to-back S1 steps (128 x 200 x 30, 7 351 speakers, random features), then reads the per-workgroup stamps of the
last launch of that problem size - tdnn2's forward GEMM of the last step - and prints, over its workgroups: in-kernel clock =
d s_memtime / d s_memrealtime x 100 MHz (median, p10, p90), the launch's span, and the same for the launch one step earlier."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from tf_kaldi_speaker_amd import _lib, engine as E


def stamps(lib, which, nwg):
    buf = np.zeros((nwg, 8), np.uint64)
    fn = getattr(lib, which)
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_size_t]
    _lib.check(fn(buf.ctypes.data_as(C.c_void_p), buf.nbytes), which)
    return buf


def report(tag, st):
    cyc = (st[:, 3] - st[:, 0]).astype(np.float64)
    rt = (st[:, 6] - st[:, 5]).astype(np.float64)
    ok = (rt > 0) & (cyc > 0)
    clk = cyc[ok] / rt[ok] * 0.1          # GHz: realtime ticks at 100 MHz
    loop = (st[ok, 2] - st[ok, 1]).astype(np.float64)
    span = (st[ok, 6].max() - st[ok, 5].min()) / 100.0
    print("%s: %d workgroups, in-kernel clock median %.3f GHz (p10 %.3f, p90 %.3f); K loop = %.1f %% of a workgroup's cycles; launch span %.1f us"
          % (tag, int(ok.sum()), np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90), 100.0 * np.median(loop / cyc[ok]), span))


def main():
    lib = _lib.load()
    if not hasattr(lib, "xv_debug_read_stamps"):
        sys.exit("step_clock.py: %s is not a diagnostics build (csrc/xv_diag.h)" % _lib.LIB_PATH)
    B, T, D, N = 128, 200, 30, 7351
    eng = E.Engine(E.make_config(D, N, loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, max_batch=B, max_frames=T), device="cuda:0")
    eng.init_variables(seed=0)
    rs = np.random.RandomState(0)
    xs = [torch.from_numpy(rs.randn(B, T, D).astype(np.float32)).cuda() for _ in range(4)]
    ys = [torch.from_numpy(rs.randint(0, N, B).astype(np.int32)).cuda() for _ in range(4)]
    t0, i = time.time(), 0
    while time.time() - t0 < 2.5 or i < 50:
        eng.train_step(xs[i % 4], ys[i % 4], 0.01, i)
        i += 1
        if i % 50 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("%d S1 steps in %.2f s (%.3f ms/step incl. the stamps' own cost)" % (i, dt, dt / i * 1e3))
    M = int(os.environ.get("XV_DIAG_M", "24576"))
    nwg = min(4096, -(-M // 128) * 4)
    report("last step ", stamps(lib, "xv_debug_read_stamps", nwg))
    report("step before", stamps(lib, "xv_debug_read_stamps_prev", nwg))
    eng.close()


if __name__ == "__main__":
    main()
