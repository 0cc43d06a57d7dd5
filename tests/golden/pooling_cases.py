"""Seeded inputs of the statistics-pooling golden cases, shared by the generator (make_pooling_golden.py, build container) and the
tests (the 6 x 186 x 1500 case is 6.7 MB as data: pooling_golden.npz stores the expected outputs and a checksum of every input,
RandomState's rand / randn streams are stable across NumPy versions).  Rows 0-3 are the adversarial rows of the reference's
pooling self-test (model/pooling.py:503-506: x 1e-8, all-zero, x 100, constant 100)."""
import numpy as np

SEED = 20261003
# (chunks, frames, channels): the self-test's own shape family, tdnn5-wide rows, T = 1, a ragged width (the HIP kernels take multiples of 4), a long chunk
SHAPES = ((6, 100, 20), (6, 186, 1500), (5, 1, 64), (6, 37, 516), (6, 401, 96))


def pooling_cases():
    rs = np.random.RandomState(SEED)
    for (b, t, c) in SHAPES:
        x = rs.rand(b, t, c).astype(np.float32)          # the self-test feeds np.random.rand float32 (pooling.py:501)
        x[0] *= 1e-8
        x[1] = 0
        x[2] *= 100
        x[3] = 100.0
        if b > 5:
            x[5] = (rs.randn(t, c) * 3 + 5).astype(np.float32)   # signed with an offset: the cancellation case
        yield x
