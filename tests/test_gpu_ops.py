"""Per-kernel parity on a real MI355X: every op-level C-ABI entry point against the CPU
oracle (oracle/xvector_oracle.py, float64 as ground truth) on the same seeded inputs.

Tolerances (fp32 arithmetic, north_star: embeddings within 1e-4 relative):
  GEMM-backed ops: ||gpu - ref||_F / ||ref||_F <= 5e-6  and  max|gpu-ref| <= 2e-5 * max|ref|
  (v_mfma_f32_32x32x2_f32 is an exact-fp32 fma chain; the error is accumulation order only).
  elementwise / reductions: 2e-5 relative to the tensor scale.
"""
import os

import numpy as np
import pytest
import torch

from oracle import xvector_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(a, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(DEV)


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy().astype(np.float64)


def assert_close(got, ref, rel_f=5e-6, rel_max=2e-5, name=""):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.all(np.isfinite(got)), name
    scale = max(np.abs(ref).max(), 1e-30)
    fro = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
    mx = np.abs(got - ref).max() / scale
    assert fro <= rel_f and mx <= rel_max, "%s: rel_fro=%.3e (<=%.1e) rel_max=%.3e (<=%.1e)" % (name, fro, rel_f, mx, rel_max)


@pytest.fixture(scope="module")
def ops():
    from tf_kaldi_speaker_amd import ops as m
    return m


# (segs, t_in, c, k, o)  - ragged M/N/K tails, the feature layer (c=30 -> pad 32), dense k=1
AFFINE_CASES = [
    (3, 40, 30, 5, 512),      # tdnn1 shape: K = 160 after padding, M = 108 (< one tile)
    (5, 61, 512, 5, 512),     # tdnn2: M = 285 (ragged), K = 2560
    (4, 50, 512, 7, 512),     # tdnn3: K = 3584
    (1, 333, 512, 1, 1500),   # tdnn5 dense: N = 1500 (ragged N), one segment
    (7, 16, 64, 7, 96),       # tiny: t_out = 10 < K-step, N < tile
    (130, 1, 3000, 1, 512),   # tdnn6: segment-level, K = 3000 (ragged K), split path
]


@pytest.mark.parametrize("segs,t_in,c,k,o", AFFINE_CASES)
def test_affine_forward_and_bn_stats(ops, segs, t_in, c, k, o):
    rs = np.random.RandomState(segs * 1000 + t_in)
    x = rs.randn(segs, t_in, c).astype(np.float32)
    kern = (rs.randn(k, c, o) / np.sqrt(k * c)).astype(np.float32)
    bias = rs.randn(o).astype(np.float32)
    c_pad = (c + 3) // 4 * 4
    xp = ops.pad_channels(dev(x.reshape(-1, c)), c_pad).view(segs, t_in, c_pad)
    wt = ops.prep_weight_fwd(dev(kern), c_pad)
    ref = O.conv1d_valid_fwd(x.astype(np.float64), kern.astype(np.float64), bias.astype(np.float64)).reshape(-1, o)
    z = ops.affine_forward(xp, k, wt, dev(bias), o)
    assert_close(host(z), ref, name="affine_forward")
    rows = ref.shape[0]
    if rows >= 2:
        z2, part = ops.affine_forward(xp, k, wt, dev(bias), o, with_stats=True)
        assert_close(host(z2), ref, name="affine_forward(stats)")
        gamma, beta = rs.rand(o).astype(np.float32) + 0.5, rs.randn(o).astype(np.float32)
        mm, mv = dev(np.zeros(o)), dev(np.ones(o))
        mean, invstd, scale, shift = ops.bn_finalize(part, rows, dev(gamma), dev(beta), 1e-3, 0.99, True, mm, mv)
        rmean, rvar = ref.mean(0), ref.var(0)
        assert_close(host(mean), rmean, 2e-5, 1e-4, "bn mean")
        assert_close(host(invstd), 1 / np.sqrt(rvar + 1e-3), 2e-5, 1e-4, "bn invstd")
        assert_close(host(mm), 0.01 * rmean, 2e-5, 1e-4, "moving mean")
        assert_close(host(mv), 0.99 + 0.01 * rvar * rows / (rows - 1), 2e-5, 1e-4, "moving var (unbiased, N4)")
        a = ops.bn_apply(z2, scale, shift, True)
        yref, _ = O.batchnorm_train_fwd(ref, gamma.astype(np.float64), beta.astype(np.float64))
        assert_close(host(a), np.maximum(yref, 0), 2e-5, 1e-4, "bn_apply+relu")


# 129 ... 160 spliced input rows (tdnn1: 5 taps x 32 padded channels): the weight gradient runs in xv_gemm_tn160_kernel - many splits,
# fewer than 160 rows (28 channels -> 140), a ragged column tile
WIDE_ROW_CASES = [(40, 150, 28, 5, 512), (20, 100, 30, 5, 96), (64, 204, 30, 5, 512)]


@pytest.mark.parametrize("segs,t_in,c,k,o", AFFINE_CASES[:5] + WIDE_ROW_CASES)
def test_affine_dgrad_wgrad(ops, segs, t_in, c, k, o):
    rs = np.random.RandomState(segs * 77 + k)
    t_out = t_in - k + 1
    x = rs.randn(segs, t_in, c).astype(np.float32)
    kern = (rs.randn(k, c, o) / np.sqrt(k * c)).astype(np.float32)
    dz = rs.randn(segs, t_out, o).astype(np.float32)
    dx_ref, dk_ref, db_ref = O.conv1d_valid_bwd(x.astype(np.float64), kern.astype(np.float64), dz.astype(np.float64))
    pad = k - 1
    dzp = np.zeros((segs, t_out + 2 * pad, o), np.float32)
    dzp[:, pad:pad + t_out] = dz
    d_dzp = dev(dzp.reshape(-1, o))
    c_pad = (c + 3) // 4 * 4
    if c % 4 == 0:
        wf = ops.prep_weight_dgrad(dev(kern))
        dx = ops.affine_dgrad(d_dzp, segs, t_out, o, k, wf, c)
        assert_close(host(dx), dx_ref.reshape(-1, c), name="affine_dgrad")
    xp = ops.pad_channels(dev(x.reshape(-1, c)), c_pad).view(segs, t_in, c_pad)
    l2 = 1e-2
    dk = ops.affine_wgrad(xp, k, c, d_dzp, t_out + 2 * pad, pad, o, dev(kern), l2)
    assert_close(host(dk), dk_ref + l2 * kern.astype(np.float64), name="affine_wgrad")
    assert_close(host(ops.colsum(d_dzp)), db_ref, 2e-5, 1e-4, "colsum (bias grad)")


def test_gemm_identity_asymmetric(ops):
    """A = I check with an ASYMMETRIC B (catches swapped row/col maps of the MFMA C/D layout)."""
    n = 256
    x = np.eye(n, dtype=np.float32)[None]                       # [1, n, n] dense rows
    kern = (np.arange(n)[:, None] * 1000 + np.arange(n)[None, :]).astype(np.float32)[None]   # [1, n, n]
    wt = ops.prep_weight_fwd(dev(kern), n)
    z = ops.affine_forward(dev(x).view(n, 1, n), 1, wt, dev(np.zeros(n)), n)
    assert np.array_equal(host(z), kern[0].astype(np.float64))


@pytest.mark.parametrize("relu,pad", [(1, 0), (1, 6), (0, 0)])
def test_bn_relu_backward(ops, relu, pad):
    rs = np.random.RandomState(5 + pad)
    segs, t, n = 6, 37, 512
    z = (rs.randn(segs * t, n) * 2 + 0.5).astype(np.float32)
    da = rs.randn(segs * t, n).astype(np.float32)
    gamma, beta = (rs.rand(n) + 0.5).astype(np.float32), (0.3 * rs.randn(n)).astype(np.float32)
    y, cache = O.batchnorm_train_fwd(z.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64))
    dy = da.astype(np.float64) * (y > 0) if relu else da.astype(np.float64)
    dz_ref, dg_ref, db_ref = O.batchnorm_train_bwd(dy, cache, gamma.astype(np.float64))
    part = ops.col_stats(dev(z))
    mean, invstd, scale, shift = ops.bn_finalize(part, segs * t, dev(gamma), dev(beta), 1e-3, 0.99, False, None, None)
    dz, dg, db, dbias = ops.bn_relu_backward(dev(da), dev(z), segs, t, dev(gamma), mean, invstd, scale, shift, relu, pad,
                                             with_dbias=True)
    # gradient of a bias in front of the BN: zero in exact arithmetic, rounding noise in fp32 (TF: reduce_sum(dz))
    assert np.abs(host(dbias)).max() <= 1e-4 * np.abs(dz_ref).sum(axis=0).max()
    dzh = host(dz).reshape(segs, t + 2 * pad, n)
    if pad:
        assert np.all(dzh[:, :pad] == 0) and np.all(dzh[:, pad + t:] == 0)
    assert_close(dzh[:, pad:pad + t].reshape(-1, n), dz_ref, 2e-5, 2e-4, "bn dz")
    assert_close(host(dg), dg_ref, 2e-5, 1e-4, "dgamma")
    assert_close(host(db), db_ref, 2e-5, 1e-4, "dbeta")


@pytest.mark.parametrize("kind", ["prelu", "lrelu"])
def test_bn_activation_through_the_op_level_abi(ops, kind):
    """network_relu_type prelu / lrelu (tdnn.py:24-30, common.py:27-42) on the OP-level entry points: the slope travels through
    xv_set_activation (include/xvector_hip.h), as a foreign binding would pass it - not only through the engine."""
    rs = np.random.RandomState(11)
    segs, t, n = 5, 41, 512
    z = (rs.randn(segs * t, n) * 2 + 0.3).astype(np.float32)
    da = rs.randn(segs * t, n).astype(np.float32)
    gamma, beta = (rs.rand(n) + 0.5).astype(np.float32), (0.3 * rs.randn(n)).astype(np.float32)
    alpha = (0.01 + 0.3 * rs.rand(n)).astype(np.float32) if kind == "prelu" else np.full(n, O.LRELU_ALPHA, np.float32)
    y, cache = O.batchnorm_train_fwd(z.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64))
    a_ref = O.act_fwd(y, kind, alpha.astype(np.float64))
    dy, dalpha_ref = O.act_bwd(y, a_ref, da.astype(np.float64), kind, alpha.astype(np.float64))
    dz_ref, dg_ref, db_ref = O.batchnorm_train_bwd(dy, cache, gamma.astype(np.float64))
    part = ops.col_stats(dev(z))
    mean, invstd, scale, shift = ops.bn_finalize(part, segs * t, dev(gamma), dev(beta), 1e-3, 0.99, False, None, None)
    d_alpha = dev(alpha)
    d_dalpha = dev(np.zeros(n, np.float32)) if kind == "prelu" else None
    with ops.activation(d_alpha, d_dalpha):
        a = ops.bn_apply(dev(z), scale, shift, 1)
        dz, dg, db = ops.bn_relu_backward(dev(da), dev(z), segs, t, dev(gamma), mean, invstd, scale, shift, 1, 0)
    assert_close(host(a), a_ref, 2e-5, 2e-4, "bn + %s" % kind)
    assert (host(a) < 0).any()                                             # the negative side is really there
    assert_close(host(dz).reshape(-1, n), dz_ref, 2e-5, 2e-4, "bn dz (%s)" % kind)
    assert_close(host(dg), dg_ref, 2e-5, 1e-4, "dgamma")
    assert_close(host(db), db_ref, 2e-5, 1e-4, "dbeta")
    if kind == "prelu":
        assert_close(host(d_dalpha), dalpha_ref, 2e-5, 1e-4, "dalpha")
    # outside the scope the same calls are plain ReLU again
    a0 = ops.bn_apply(dev(z), scale, shift, 1)
    assert_close(host(a0), np.maximum(y, 0), 2e-5, 2e-4, "bn + relu after the scope")


def test_statistics_pooling_adversarial(ops):
    """Rows from the reference's pooling self-test (pooling.py:478-510 style): tiny, zero, huge, constant."""
    rs = np.random.RandomState(9)
    b, t, c = 6, 186, 1500
    x = rs.rand(b, t, c).astype(np.float32)
    x[0] *= 1e-8
    x[1] = 0
    x[2] *= 100
    x[3] = 100.0
    ref, cache = O.statistics_pooling_fwd(x.astype(np.float64))
    out = ops.stat_pool_forward(dev(x))
    got = host(out)
    assert_close(got[:, :c], ref[:, :c], 2e-6, 1e-5, "pool mean")
    # std of the constant / zero rows is exactly sqrt(1e-12); elsewhere relative agreement
    assert np.allclose(got[[1, 3], c:], 1e-6, rtol=1e-5)
    assert_close(got[[0, 2, 4, 5], c:], ref[[0, 2, 4, 5], c:], 1e-5, 1e-4, "pool std")
    dout = rs.randn(b, 2 * c).astype(np.float32)
    dref = O.statistics_pooling_bwd(x.astype(np.float64), cache, dout.astype(np.float64))
    dx = host(ops.stat_pool_backward(dev(x), out, dev(dout)))
    assert np.all(np.isfinite(dx)), "Gradient should not be nan"
    for i in (2, 4, 5):
        assert_close(dx[i], dref[i], 2e-5, 2e-4, "pool dx row %d" % i)
    # clamped-variance rows: only the mean path carries gradient
    assert_close(dx[1], np.broadcast_to(dout[1, None, :c] / t, (t, c)), 1e-6, 1e-5, "pool dx zero row")


def test_statistics_pooling_matches_reference_numpy(ops):
    """xv_stat_pool_forward vs the reference's own NumPy (compute_self_attention with a zero query = uniform weights over time,
    model/test_utils.py:320-372; tests/golden/make_pooling_golden.py) on inputs with the reference self-test's adversarial rows."""
    from test_oracle_golden import _pooling_golden_cases, check_pooling_against_reference
    n = 0
    for i, x, ref in _pooling_golden_cases():
        got = host(ops.stat_pool_forward(dev(x)))
        check_pooling_against_reference(got, ref, x, 2e-6, 2e-5, "case %d" % i)
        n += 1
    assert n == 5


@pytest.mark.parametrize("t", [1, 7, 15, 186, 401])
def test_statistics_pooling_lengths(ops, t):
    rs = np.random.RandomState(t)
    x = (rs.randn(3, t, 1500) * 3 + 5).astype(np.float32)
    ref, _ = O.statistics_pooling_fwd(x.astype(np.float64))
    got = host(ops.stat_pool_forward(dev(x)))
    assert_close(got, ref, 5e-6, 5e-5, "pool T=%d" % t)


def _pooled_case(seed, b, t, n):
    rs = np.random.RandomState(seed)
    z = (rs.randn(b * t, n) * 2 + 0.3).astype(np.float32)
    gamma, beta = (rs.rand(n) + 0.5).astype(np.float32), (0.3 * rs.randn(n)).astype(np.float32)
    dout = rs.randn(b, 2 * n).astype(np.float32)
    z64, g64, b64 = z.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64)
    y, cache = O.batchnorm_train_fwd(z64, g64, b64)
    a = np.maximum(y, 0).reshape(b, t, n)
    pool_ref, pcache = O.statistics_pooling_fwd(a)
    da = O.statistics_pooling_bwd(a, pcache, dout.astype(np.float64)).reshape(b * t, n)
    dz_ref, dg_ref, db_ref = O.batchnorm_train_bwd(da * (y > 0), cache, g64)
    return z, gamma, beta, dout, pool_ref, dz_ref, dg_ref, db_ref


@pytest.mark.parametrize("b,t,n", [(5, 37, 1500), (3, 186, 512)])
def test_pooling_fused_into_bn(ops, b, t, n):
    """tdnn5 -> statistics pooling without the activation in memory: forward pools relu(bn(z)) on the fly, backward
    evaluates pooling backward + ReLU backward + BN backward from (pool, d pool) - against the composed oracle."""
    z, gamma, beta, dout, pool_ref, dz_ref, dg_ref, db_ref = _pooled_case(17 + t, b, t, n)
    part = ops.col_stats(dev(z))
    mean, invstd, scale, shift = ops.bn_finalize(part, b * t, dev(gamma), dev(beta), 1e-3, 0.99, False, None, None)
    pool = ops.stat_pool_forward_bn(dev(z), b, t, scale, shift, True)
    assert_close(host(pool), pool_ref, 5e-6, 5e-5, "fused pool")
    dz, dg, db, dbias = ops.bn_relu_backward_pooled(pool, dev(dout), b, t, dev(z), dev(gamma), mean, invstd, scale, shift, True)
    assert_close(host(dz), dz_ref, 2e-5, 2e-4, "pooled bn dz")
    assert_close(host(dg), dg_ref, 2e-5, 1e-4, "pooled dgamma")
    assert_close(host(db), db_ref, 2e-5, 1e-4, "pooled dbeta")


@pytest.mark.parametrize("b,t,n", [(5, 37, 1500), (128, 186, 512), (7, 64, 96)])
def test_pooled_bn_backward_closed_form(ops, b, t, n):
    """The closed-form reductions (xv_bn_relu_backward_pooled_aux: sum dy and sum dy*xhat per channel from the pooled statistics and
    the forward's by-product wpos, no pass over z) against the composed oracle and against the direct pass - also with chunks that
    sit on the special branches: a constant chunk (variance clamp: no std gradient), a chunk 100x larger, a channel whose ReLU is
    off for a whole chunk (wpos = 0) and one that is on everywhere (wpos = 1)."""
    rs = np.random.RandomState(b * 100 + t)
    z = (rs.randn(b, t, n) * 2 + 0.3).astype(np.float32)
    z[0] = z[0, :1]                      # chunk 0: every frame the same
    z[1] *= 100.0
    z[2, :, 3] = -50.0                   # off for the whole chunk
    z[2, :, 5] = 50.0 + rs.rand(t)       # on for the whole chunk
    z = z.reshape(b * t, n)
    gamma, beta = (rs.rand(n) + 0.5).astype(np.float32), (0.3 * rs.randn(n)).astype(np.float32)
    gamma[7] = -gamma[7]                 # a negative scale
    dout = rs.randn(b, 2 * n).astype(np.float32)
    z64, g64, b64 = z.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64)
    y, cache = O.batchnorm_train_fwd(z64, g64, b64)
    a = np.maximum(y, 0).reshape(b, t, n)
    pool_ref, pcache = O.statistics_pooling_fwd(a)
    da = O.statistics_pooling_bwd(a, pcache, dout.astype(np.float64)).reshape(b * t, n)
    dz_ref, dg_ref, db_ref = O.batchnorm_train_bwd(da * (y > 0), cache, g64)
    part = ops.col_stats(dev(z))
    mean, invstd, scale, shift = ops.bn_finalize(part, b * t, dev(gamma), dev(beta), 1e-3, 0.99, False, None, None)
    pool, wpos, amax = ops.stat_pool_forward_bn_aux(dev(z), b, t, scale, shift, True)
    assert_close(host(pool), pool_ref, 5e-6, 5e-5, "pool")
    ygpu = host(dev(z)) * host(scale) + host(shift)      # the GPU's own on/off pattern (fp32 scale / shift)
    on = (ygpu.astype(np.float32) > 0).reshape(b, t, n)
    assert np.abs(host(wpos) - on.mean(axis=1)).max() <= 2.0 / t + 1e-6          # rounding-level pre-activations may flip a frame or two
    assert_close(host(amax), a.max(axis=1), 2e-5, 2e-4, "amax")
    dz, dg, db, dbias = ops.bn_relu_backward_pooled_aux(pool, dev(dout), wpos, b, t, dev(z), dev(gamma), mean, invstd, scale, shift, True)
    dz2, dg2, db2, _ = ops.bn_relu_backward_pooled(pool, dev(dout), b, t, dev(z), dev(gamma), mean, invstd, scale, shift, True)
    for name, got, direct, ref, tol in (("dz", dz, dz2, dz_ref, 2e-4), ("dgamma", dg, dg2, dg_ref, 1e-4), ("dbeta", db, db2, db_ref, 1e-4)):
        assert_close(host(got), ref, 2e-5, tol, "closed-form " + name)
        assert_close(host(got), host(direct), 2e-5, tol, "closed-form vs direct " + name)
    assert np.abs(host(dbias)).max() <= 1e-4 * max(np.abs(dz_ref).sum(axis=0).max(), 1e-30)


@pytest.mark.parametrize("kind,weighted,b,t,n", [("prelu", False, 5, 37, 1500), ("lrelu", False, 3, 186, 512), ("relu", True, 4, 70, 512),
                                                  ("prelu", True, 3, 129, 96), ("none", False, 2, 64, 512)])
def test_pooled_bn_backward_direct_pass(ops, kind, weighted, b, t, n):
    """The pass over z that serves the cases without a closed form (bn_bwd_reduce_pooled_kernel: prelu / lrelu slopes, attention frame
    weights, no activation; common.py:27-42, pooling.py:148-155): T below, at and above the 64-row block, T not a multiple of 4."""
    rs = np.random.RandomState(b * 1000 + t)
    z = (rs.randn(b * t, n) * 2 + 0.3).astype(np.float32)
    gamma, beta = (rs.rand(n) + 0.5).astype(np.float32), (0.3 * rs.randn(n)).astype(np.float32)
    dout = rs.randn(b, 2 * n).astype(np.float32)
    alpha = {"prelu": (0.01 + 0.3 * rs.rand(n)).astype(np.float32), "lrelu": np.full(n, O.LRELU_ALPHA, np.float32)}.get(kind)
    w = None
    if weighted:
        w = rs.rand(b, t) + 0.05
        w = (w / w.sum(axis=1, keepdims=True)).astype(np.float32)
    z64, g64, b64 = z.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64)
    y, cache = O.batchnorm_train_fwd(z64, g64, b64)
    a = {"relu": np.maximum(y, 0), "none": y}.get(kind)
    if a is None:
        a = O.act_fwd(y, kind, alpha.astype(np.float64))
    a3 = a.reshape(b, t, n)
    w64 = np.full((b, t), 1.0 / t) if w is None else w.astype(np.float64)
    mean = (w64[:, :, None] * a3).sum(axis=1)
    var = (w64[:, :, None] * (a3 - mean[:, None]) ** 2).sum(axis=1)
    std = np.sqrt(np.maximum(var, 1e-12))
    pool = np.concatenate([mean, std], axis=1)
    d64 = dout.astype(np.float64)
    da = (d64[:, None, :n] * w64[:, :, None] + (d64[:, None, n:] / std[:, None]) * w64[:, :, None] * (a3 - mean[:, None])).reshape(b * t, n)
    if kind == "relu":
        dy, dalpha_ref = da * (y > 0), None
    elif kind == "none":
        dy, dalpha_ref = da, None
    else:
        dy, dalpha_ref = O.act_bwd(y, a, da, kind, alpha.astype(np.float64))
    dz_ref, dg_ref, db_ref = O.batchnorm_train_bwd(dy, cache, g64)
    part = ops.col_stats(dev(z))
    mean_d, invstd, scale, shift = ops.bn_finalize(part, b * t, dev(gamma), dev(beta), 1e-3, 0.99, False, None, None)
    d_dalpha = dev(np.zeros(n, np.float32)) if kind == "prelu" else None
    args = (dev(pool.astype(np.float32)), dev(dout), b, t, dev(z), dev(gamma), mean_d, invstd, scale, shift, kind != "none")
    kw = dict(weights=dev(w.reshape(-1)) if weighted else None)
    if alpha is not None:
        with ops.activation(dev(alpha), d_dalpha):
            dz, dg, db, _ = ops.bn_relu_backward_pooled(*args, **kw)
    else:
        dz, dg, db, _ = ops.bn_relu_backward_pooled(*args, **kw)
    assert_close(host(dz), dz_ref, 2e-5, 2e-4, "pooled dz (%s)" % kind)
    assert_close(host(dg), dg_ref, 2e-5, 1e-4, "pooled dgamma (%s)" % kind)
    assert_close(host(db), db_ref, 2e-5, 1e-4, "pooled dbeta (%s)" % kind)
    if kind == "prelu":
        assert_close(host(d_dalpha), dalpha_ref, 2e-5, 1e-4, "pooled dalpha")


def test_l2_scaling(ops):
    rs = np.random.RandomState(3)
    x = rs.randn(100, 512).astype(np.float32)
    x[2] *= 1e-8
    x[3] *= 100
    y_ref, cache = O.l2_scaling_fwd(x.astype(np.float64), 30.0)
    y = host(ops.l2_scaling_forward(dev(x), 30.0))
    assert_close(y, y_ref, 2e-6, 1e-5, "l2_scaling")
    assert np.allclose(np.linalg.norm(y[3:], axis=1), 30.0, rtol=1e-5)   # the reference's own check, tdnn.py:246-247
    dy = rs.randn(100, 512).astype(np.float32)
    dx_ref = O.l2_scaling_bwd(x.astype(np.float64), cache, dy.astype(np.float64), 30.0)
    dx = host(ops.l2_scaling_backward(dev(x), dev(dy), 30.0))
    for r in (0, 1, 3, 50):
        assert_close(dx[r], dx_ref[r], 2e-5, 2e-4, "l2_scaling bwd row %d" % r)


G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_golden.npz"))
KIND_ID = {"asoftmax": 1, "additive_margin_softmax": 2, "additive_angular_margin_softmax": 3}


def test_loss_family_against_reference_golden_vectors(ops):
    """HIP loss path (normalise W -> MFMA logits -> margin/xent rows) against the reference's own
    NumPy oracle outputs (tests/golden/loss_golden.npz), tolerance = the reference's np.allclose
    bar scaled for fp32 (rtol 2e-5 + atol 1e-6), and gradients against the pinned oracle."""
    n_cases = len(G["loss"])
    worst = 0.0
    for i in range(n_cases):
        kind = str(G["kind"][i])
        m = float(G["m"][i])
        emb, w, labels = G["emb"][i], G["w"][i], G["labels"][i]
        lam = O.margin_lambda(*G["sched"][i], int(G["step"][i]))
        e64 = emb.astype(np.float64)
        x = dev(emb)
        if G["feature_norm"][i]:
            x = ops.l2_scaling_forward(x, 0.1)
            e64, _ = O.l2_scaling_fwd(e64, 0.1)
        inv, wn, wnt = ops.loss_prep_weight(dev(w), True)
        c, n = w.shape
        ldl = wn.shape[1]
        xp = x.view(x.shape[0], 1, c)
        logits = torch.zeros((x.shape[0], ldl), dtype=torch.float32, device=DEV)
        logits[:, :n] = ops.affine_forward(xp, 1, wnt, None, n)
        loss, dlogits, dnorm, _ = ops.margin_softmax_rows(KIND_ID[kind], logits, n, x, dev(labels, np.int32), m, lam)
        got = float(host(loss)[0])
        ref = float(G["loss"][i])
        assert abs(got - ref) <= 1e-6 + 2e-5 * abs(ref), (i, kind, m, got, ref)
        worst = max(worst, abs(got - ref) / max(abs(ref), 1e-9))
        # gradients vs the pinned oracle (float64)
        mm = int(m) if kind == "asoftmax" else m
        _, lg_ref, (df_ref, dk_ref) = O.margin_softmax_loss(kind, e64, labels, w.astype(np.float64), mm, lam)
        assert_close(host(logits)[:, :n], lg_ref, 5e-6, 2e-5, "logits")
        assert np.all(host(dlogits)[:, n:] == 0)
        wf = wn[:, :ldl]
        dx = ops.affine_forward(dlogits.view(dlogits.shape[0], 1, ldl), 1, wf, None, c)
        dx = ops.add_norm_grad(x, dnorm, dx)
        assert np.all(np.isfinite(host(dx))), "Gradient should not be nan (tdnn.py:282)"
        # rows 0/1 sit at theta ~ 0 / pi where d phi / d cos ~ cos/sin is ill-conditioned (fp32 cos rounds
        # to exactly +-1 or one ulp inside): the reference only asserts NaN-freeness there (tdnn.py:282).
        gscale = max(np.abs(df_ref[2:]).max(), 1e-12)
        assert np.abs(host(dx)[2:] - df_ref[2:]).max() <= 2e-4 * gscale + 1e-7, (i, kind, m)
    assert worst < 5e-5


def test_loss_weight_backward_and_softmax_kind(ops):
    rs = np.random.RandomState(21)
    rows, c, n = 32, 512, 1001     # odd N: logits pitch padded to 1004
    x = rs.randn(rows, c).astype(np.float32)
    w = (rs.randn(c, n) * 0.05).astype(np.float32)
    b = (rs.randn(n) * 0.1).astype(np.float32)
    labels = rs.randint(0, n, rows).astype(np.int32)
    for kind, name, m in ((0, "softmax", 0.0), (2, "additive_margin_softmax", 0.2), (3, "additive_angular_margin_softmax", 0.3),
                          (1, "asoftmax", 4)):
        normalize = kind != 0
        inv, wn, wnt = ops.loss_prep_weight(dev(w), normalize)
        ldl = wn.shape[1]
        logits = torch.zeros((rows, ldl), dtype=torch.float32, device=DEV)
        logits[:, :n] = ops.affine_forward(dev(x).view(rows, 1, c), 1, wnt, dev(b) if kind == 0 else None, n)
        lam = 0.5
        loss, dlogits, dnorm, _ = ops.margin_softmax_rows(kind, logits, n, dev(x), dev(labels, np.int32), m, lam)
        if kind == 0:
            lref, _, (df_ref, dk_ref, db_ref) = O.softmax_loss(x.astype(np.float64), labels, w.astype(np.float64), b.astype(np.float64))
            assert_close(host(ops.colsum(dlogits[:, :n])), db_ref, 2e-5, 1e-4, "softmax bias grad")
        else:
            lref, _, (df_ref, dk_ref) = O.margin_softmax_loss(name, x.astype(np.float64), labels, w.astype(np.float64), m, lam)
        assert abs(float(host(loss)[0]) - lref) <= 2e-5 * abs(lref) + 1e-6, name
        # d wn = x^T dlogits through the weight-gradient GEMM, then through l2_normalize
        dwn = ops.affine_wgrad(dev(x).view(rows, 1, c), 1, c, dlogits, 1, 0, ldl, None, 0.0)[0]      # [c, ldl]
        dw = ops.loss_weight_backward(dwn, wn, inv, dev(w), normalize, 1e-2)
        assert_close(host(dw), dk_ref + 1e-2 * w.astype(np.float64), 2e-5, 2e-4, "loss dW " + name)


def test_optimizers_and_reductions(ops):
    rs = np.random.RandomState(1)
    n = 100003
    p0, g = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
    p = dev(p0)
    ops.sgd_update(p, dev(g), 0.1, 0.5)
    assert_close(host(p), O.sgd_update(p0.astype(np.float64), 0.5 * g.astype(np.float64), 0.1), 1e-6, 1e-6, "sgd")
    for nesterov in (False, True):
        p, acc = dev(p0), dev(np.zeros(n))
        pr, ar = p0.astype(np.float64), np.zeros(n)
        for _ in range(3):
            ops.momentum_update(p, dev(g), acc, 0.01, 0.9, nesterov)
            pr, ar = O.momentum_update(pr, g.astype(np.float64), ar, 0.01, 0.9, nesterov)
        assert_close(host(p), pr, 1e-6, 1e-6, "momentum")
    p, m, v = dev(p0), dev(np.zeros(n)), dev(np.zeros(n))
    pr, mr, vr = p0.astype(np.float64), np.zeros(n), np.zeros(n)
    for t in (1, 2, 3):
        ops.adam_update(p, dev(g), m, v, 0.001, t)
        pr, mr, vr = O.adam_update(pr, g.astype(np.float64), mr, vr, t, 0.001)
    assert_close(host(p), pr, 1e-6, 1e-5, "adam")
    acc = dev(np.zeros(1))
    ops.l2_reg_loss(dev(g), 1e-2, acc)
    assert abs(float(host(acc)[0]) - 0.005 * float((g.astype(np.float64) ** 2).sum())) < 1e-4 * 0.005 * n


@pytest.mark.parametrize("act,use_scale", [(3, True), (0, False)])
def test_self_attention_pieces(ops, act, use_scale):
    """The kernels around the key network's GEMMs (pooling.py:37-192, shipped single-head form) against the oracle:
    score -> softmax over frames -> weighted statistics of relu(bn(z)) on the fly; backward: d weights, softmax backward,
    the key-layer gradient (dzk, d query, d bias) and the value path through tdnn5's BN backward with frame weights."""
    rs = np.random.RandomState(31 + act)
    b, t, n, dk = 5, 37, 1500, 1500
    z = (rs.randn(b * t, n) * 2 + 0.3).astype(np.float32)          # tdnn5 pre-BN output
    zk = (rs.randn(b * t, dk) * 0.7).astype(np.float32)            # key layer pre-activation
    query = (rs.randn(1, dk) * 0.1).astype(np.float32)
    gamma, beta = (rs.rand(n) + 0.5).astype(np.float32), (0.3 * rs.randn(n)).astype(np.float32)
    dout = rs.randn(b, 2 * n).astype(np.float32)
    z64, zk64, q64 = z.astype(np.float64), zk.astype(np.float64), query.astype(np.float64)
    y, cache = O.batchnorm_train_fwd(z64, gamma.astype(np.float64), beta.astype(np.float64))
    value = np.maximum(y, 0).reshape(b, t, n)
    key = (np.tanh(zk64) if act == 3 else zk64).reshape(b, t, dk)
    pool_ref, pc = O.self_attention_fwd(value, key, q64, use_scale)
    dv, dkey, dq_ref = O.self_attention_bwd(value, key, q64, pc, dout.astype(np.float64))
    dzk_ref = dkey.reshape(b * t, dk) * ((1 - key.reshape(b * t, dk) ** 2) if act == 3 else 1.0)
    dz_ref, dg_ref, db_ref = O.batchnorm_train_bwd(dv.reshape(b * t, n) * (y > 0), cache, gamma.astype(np.float64))
    scale_f = 1.0 / np.sqrt(dk) if use_scale else 1.0

    part = ops.col_stats(dev(z))
    mean, invstd, sc, sh = ops.bn_finalize(part, b * t, dev(gamma), dev(beta), 1e-3, 0.99, False, None, None)
    score = ops.att_score(dev(zk), act, dev(query), scale_f)
    assert_close(host(score), (key.reshape(b * t, dk) @ q64[0]) * scale_f, 2e-6, 2e-5, "att score")
    w = ops.softmax_segments(score, b, t)
    assert_close(host(w), pc[0], 2e-6, 2e-5, "attention weights")
    assert np.allclose(host(w).sum(axis=1), 1.0, atol=1e-5)
    pool = ops.stat_pool_forward_bn(dev(z), b, t, sc, sh, True, weights=w)
    assert_close(host(pool), pool_ref, 5e-6, 5e-5, "attention pooling")
    dw = ops.att_pool_backward_weights(dev(z), b, t, sc, sh, True, pool, dev(dout))
    dw_ref = np.einsum("btc,bc->bt", value, dout[:, :n].astype(np.float64)) + \
        np.einsum("btc,bc->bt", (value - pc[1][:, None, :]) ** 2, dout[:, n:] * 0.5 / pc[2] * (1 - pc[3]))
    assert_close(host(dw), dw_ref, 5e-6, 5e-5, "d weights")
    ds = ops.softmax_segments_backward(w, dw)
    dzk, dq, dbias = ops.att_key_backward(dev(zk), act, dev(query), scale_f, ds.reshape(-1))
    assert_close(host(dzk), dzk_ref, 1e-5, 1e-4, "d key pre-activation")
    assert_close(host(dq), dq_ref[0], 1e-5, 1e-4, "d query")
    # the frame gradients of a chunk sum to zero (softmax), so for the affine key the bias gradient is 0 + rounding noise
    assert np.abs(host(dbias) - dzk_ref.sum(axis=0)).max() <= 1e-5 * np.abs(dzk_ref).sum(axis=0).max(), "d key bias"
    dz, dg, db, _ = ops.bn_relu_backward_pooled(pool, dev(dout), b, t, dev(z), dev(gamma), mean, invstd, sc, sh, True, weights=w)
    assert_close(host(dz), dz_ref, 2e-5, 2e-4, "value path: tdnn5 dz")
    assert_close(host(dg), dg_ref, 2e-5, 1e-4, "value path: dgamma")
    assert_close(host(db), db_ref, 2e-5, 1e-4, "value path: dbeta")
    y2 = host(dzk).copy()
    ops._lib.call("xv_add_inplace", ops._s(), ops._p(dzk), ops._p(dzk), ops.C.c_size_t(dzk.numel()))
    assert np.array_equal(host(dzk), 2 * y2)


def test_auxiliary_losses_against_reference_golden_vectors(ops):
    """xv_ring_loss / xv_mhe_loss vs the reference's NumPy oracles (tests/golden/aux_golden.npz) and, for the gradient pieces,
    the float64 oracle; then the functional form model.loss.additive_margin_softmax with aux_loss_func set."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "aux_golden.npz"))
    for i in range(int(g["num_cases"])):
        feats, w, labels = g["features_%d" % i], g["w_%d" % i], g["labels_%d" % i].astype(np.int32)
        r, lam = float(g["r_%d" % i]), float(g["lambda_%d" % i])
        x = dev(feats.astype(np.float32))
        rows, n = feats.shape
        out, dn, dr = dev(np.zeros(1, np.float32)), dev(np.zeros(rows, np.float32)), dev(np.zeros(1, np.float32))
        ops._lib.call("xv_ring_loss", ops._s(), ops._p(x), rows, n, n, ops._p(dev(np.array([r], np.float32))), lam, ops._p(out), ops._p(dn),
                      ops._p(dr))
        assert np.isclose(host(out)[0], float(g["ring_%d" % i]), rtol=2e-5)
        _, (dfeat, dr_ref) = O.ring_loss(feats, r, lam)
        norm = np.sqrt((feats ** 2).sum(axis=1))
        assert_close(host(dn), (dfeat * feats).sum(axis=1) / norm, 2e-5, 1e-4, "ring d||x||")
        assert np.isclose(host(dr)[0], dr_ref, rtol=1e-4)
        inv, wn, _ = ops.loss_prep_weight(dev(w.astype(np.float32)), True)
        assert np.isclose(float(ops.mhe_loss(wn, w.shape[1], dev(labels, np.int32), lam).cpu()), float(g["mhe_%d" % i]), rtol=5e-5)
        # gradient w.r.t. the normalised weights: g*(u + cnt*v)
        c = w.shape[0]
        coef, counts = dev(np.zeros(1 + 2 * c, np.float32)), torch.zeros(w.shape[1], dtype=torch.int32, device="cuda")
        ops._lib.call("xv_mhe_loss", ops._s(), ops._p(wn), c, w.shape[1], wn.shape[1], ops._p(dev(labels, np.int32)), rows, lam, ops._p(out), ops._p(coef),
                      ops._p(counts))
        dwn = torch.zeros_like(wn)
        ops._lib.call("xv_mhe_add_grad", ops._s(), ops._p(dwn), c, w.shape[1], wn.shape[1], ops._p(coef), ops._p(counts))
        wn64 = w / np.sqrt((w ** 2).sum(axis=0, keepdims=True))
        u, v = wn64[:, labels].sum(axis=1), wn64.sum(axis=1)
        M = 2 - 2 * (u @ v) / (rows * w.shape[1]) + 1e-6
        ref = 2 * lam / (M * M * rows * w.shape[1]) * (u[:, None] + v[:, None] * np.bincount(labels, minlength=w.shape[1])[None, :])
        assert_close(host(dwn)[:, :w.shape[1]], ref, 2e-5, 1e-4, "mhe d wn")
    from tf_kaldi_speaker_amd.misc.utils import Params
    from tf_kaldi_speaker_amd.model import loss as L
    L.reset_variables() if hasattr(L, "reset_variables") else L._VARIABLES.clear()
    p = Params.__new__(Params)
    p.__dict__.update(dict(amsoftmax_m=0.2, amsoftmax_lambda_min=0, amsoftmax_lambda_base=1000, amsoftmax_lambda_gamma=1e-4,
                           amsoftmax_lambda_power=5, weight_l2_regularizer=1e-2, aux_loss_func=["ring_loss", "mhe_loss"],
                           ring_loss_init=2.0, ring_loss_lambda=0.05, mhe_lambda=0.05))
    rs = np.random.RandomState(4)
    feats, labels = rs.randn(12, 32).astype(np.float32), rs.randint(0, 9, 12).astype(np.int32)
    total, ep = L.additive_margin_softmax(feats, labels, 9, p)
    w = ep["w"].cpu().numpy().astype(np.float64)
    base, _, _ = O.margin_softmax_loss("additive_margin_softmax", feats.astype(np.float64), labels, w, 0.2,
                                       O.margin_lambda(0, 1000, 1e-4, 5, 0))
    ref = base + O.ring_loss(feats.astype(np.float64), 2.0, 0.05)[0] + O.mhe_loss(w, labels, 0.05)[0]
    assert abs(float(total.cpu()) - ref) <= 2e-5 * abs(ref), (float(total.cpu()), ref)
    assert "ring_loss_r" in ep
    L._VARIABLES.clear()


# (rows, n, k): the segment-level problems of the reference topology (tdnn6 K=3000, tdnn7, logits N=7351, d out K=7352, d pool N=3000)
# plus ragged rows / columns / K tails (K % 8 == 4, N % 32 != 0, one row, a single split)
SEGMENT_CASES = [(128, 512, 3000), (64, 512, 512), (128, 7351, 512), (128, 512, 7352), (128, 3000, 512),
                 (1, 512, 3000), (37, 100, 68), (128, 33, 4), (2, 600, 1204)]


# (segs, t_in, c, k, o): tile counts between one and four per CU with a remainder -> whole tiles + shares (csrc/xv_gemm.hip xv_nt_shares)
SHARE_CASES = [
    (130, 135, 192, 5, 512),     # forward 17 030 rows x 512 = 536 tiles (24 remaining), K = 960; data gradient 17 550 rows = 552 tiles
    (66, 151, 64, 5, 512),       # forward 9 702 rows = 304 tiles: one whole tile per CU - shared in the forward launch only
    (1, 5600, 512, 1, 1500),     # dense, ragged N: 44 x 12 = 528 tiles, K = 512 (32 K-steps)
]


@pytest.mark.parametrize("segs,t_in,c,k,o", SHARE_CASES)
def test_whole_tiles_plus_shares(ops, segs, t_in, c, k, o):
    """The remainder tiles of these launches are summed from K shares by their last block (slab hand-over inside xv_gemm_nt_kernel): values and
    BatchNorm partials against the oracle, and every repetition - beside an uneven load on another stream - bit-identical to the first
    (a stale slab word or a ticket left non-zero would show)."""
    import torch
    rs = np.random.RandomState(segs + t_in)
    t_out = t_in - k + 1
    x = rs.randn(segs, t_in, c).astype(np.float32)
    kern = (rs.randn(k, c, o) / np.sqrt(k * c)).astype(np.float32)
    bias = rs.randn(o).astype(np.float32)
    xp = dev(x)
    wt = ops.prep_weight_fwd(dev(kern), c)
    ref = O.conv1d_valid_fwd(x.astype(np.float64), kern.astype(np.float64), bias.astype(np.float64)).reshape(-1, o)
    z, part = ops.affine_forward(xp, k, wt, dev(bias), o, with_stats=True)
    z, part = z.clone(), part.clone()
    assert_close(host(z), ref, name="affine_forward (shares)")
    rows = ref.shape[0]
    gamma, beta = rs.rand(o).astype(np.float32) + 0.5, rs.randn(o).astype(np.float32)
    mean, invstd, _, _ = ops.bn_finalize(part, rows, dev(gamma), dev(beta), 1e-3, 0.99, True, dev(np.zeros(o)), dev(np.ones(o)))
    assert_close(host(mean), ref.mean(0), 2e-5, 1e-4, "bn mean")
    assert_close(host(invstd), 1 / np.sqrt(ref.var(0) + 1e-3), 2e-5, 1e-4, "bn invstd")
    dz = rs.randn(segs, t_out, o).astype(np.float32)
    pad = k - 1
    dzp = np.zeros((segs, t_out + 2 * pad, o), np.float32)
    dzp[:, pad:pad + t_out] = dz
    d_dzp = dev(dzp.reshape(-1, o))
    wf = ops.prep_weight_dgrad(dev(kern))
    dx = ops.affine_dgrad(d_dzp, segs, t_out, o, k, wf, c).clone()
    dx_ref = O.conv1d_valid_bwd(x.astype(np.float64), kern.astype(np.float64), dz.astype(np.float64))[0]
    assert_close(host(dx), dx_ref.reshape(-1, c), name="affine_dgrad (shares)")
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device="cuda")
    bad = 0
    for rep in range(3):
        with torch.cuda.stream(side):
            for _ in range(3):
                big @ big
        for _ in range(6):
            z2, part2 = ops.affine_forward(xp, k, wt, dev(bias), o, with_stats=True)
            dx2 = ops.affine_dgrad(d_dzp, segs, t_out, o, k, wf, c)
            bad += int(not torch.equal(z2, z)) + int(not torch.equal(part2, part)) + int(not torch.equal(dx2, dx))
        torch.cuda.synchronize()
    assert bad == 0, "%d of 54 results differ from the first launch" % bad


def test_split_handoff_stress(ops):
    """The split-K hand-over of the one-launch segment kernels (relaxed agent-scope stores / ticket / loads: the xv_handoff_* contract of
    csrc/xv_common.h, an architecture property of gfx950 rather than a HIP memory-model guarantee - ADVICE r02) replayed many times
    back to back, the consumer L1-warm, under UNEVEN load (a large GEMM on another stream occupies part of the chip): every launch must
    reproduce the first result bit for bit - a stale slab word would show as a different sum - and leave its tickets at zero."""
    import torch
    rs = np.random.RandomState(3)
    rows, n, k = 128, 512, 3000                        # tdnn6: the largest split count of the chain
    x, wt, bias = rs.randn(rows, k).astype(np.float32), (rs.randn(n, k) / 50).astype(np.float32), rs.randn(n).astype(np.float32)
    dx, dw, db = dev(x), dev(wt), dev(bias)
    first = ops.segment_gemm(dx, dw, db).clone()
    torch.cuda.synchronize()
    assert_close(host(first), x.astype(np.float64) @ wt.astype(np.float64).T + bias, name="segment_gemm")
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device="cuda")
    bad = 0
    for rep in range(8):
        with torch.cuda.stream(side):
            for _ in range(3):
                big @ big                               # uneven load beside the hand-overs
        outs = [ops.segment_gemm(dx, dw, db).clone() for _ in range(150)]
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, first)) for o in outs)
    assert bad == 0, "%d of 1200 launches differ from the first" % bad
    # the fused BatchNorm form consumes the summed columns inside the same launch: same replay on it
    gamma, beta = dev((rs.rand(n) + 0.5).astype(np.float32)), dev((0.3 * rs.randn(n)).astype(np.float32))
    def bn():
        mm, mv = dev(np.zeros(n)), dev(np.ones(n))
        return ops.segment_affine_bn_forward(dx, dw, db, gamma, beta, 1e-3, 0.99, True, mm, mv, 1)[1].clone()
    a0 = bn()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(6):
            big @ big
    outs = [bn() for _ in range(200)]
    torch.cuda.synchronize()
    assert all(torch.equal(o, a0) for o in outs)


@pytest.mark.parametrize("rows,n,k", SEGMENT_CASES)
def test_segment_gemm_and_fused_batchnorm(ops, rows, n, k):
    """xv_segment_gemm / _affine_bn_forward / _dgrad_bn_backward (GEMM + split-K sum + the consumer's BatchNorm in one launch)
    against the oracle's dense / BatchNorm forward and backward; twice on the same tickets (they must come back zeroed)."""
    rs = np.random.RandomState(rows * 7 + n)
    x = rs.randn(rows, k).astype(np.float32)
    wt = (rs.randn(n, k) / np.sqrt(k)).astype(np.float32)
    bias = rs.randn(n).astype(np.float32)
    x64, wt64 = x.astype(np.float64), wt.astype(np.float64)
    ref = x64 @ wt64.T + bias
    for _ in range(2):
        assert_close(host(ops.segment_gemm(dev(x), dev(wt), dev(bias))), ref, name="segment_gemm")
    # the row term (gradient through ||x||): + (coef / norm) * xrow, 0 where norm == 0
    coef, xrow = rs.randn(rows).astype(np.float32), rs.randn(rows, n).astype(np.float32)
    norm = np.abs(rs.randn(rows)).astype(np.float32) + 0.1
    norm[0] = 0.0
    kf = np.where(norm > 0, coef / np.maximum(norm, 1e-30), 0.0).astype(np.float64)
    got = ops.segment_gemm(dev(x), dev(wt), None, row_term=(dev(coef), dev(norm), dev(xrow)))
    assert_close(host(got), x64 @ wt64.T + kf[:, None] * xrow, name="segment_gemm(row term)")
    if rows < 2:
        return
    gamma, beta = (rs.rand(n) + 0.5).astype(np.float32), (0.3 * rs.randn(n)).astype(np.float32)
    for relu in (1, 0):
        mm, mv = dev(np.zeros(n)), dev(np.ones(n))
        z, a, mean, invstd, scale, shift = ops.segment_affine_bn_forward(dev(x), dev(wt), dev(bias), dev(gamma), dev(beta), 1e-3, 0.99,
                                                                         True, mm, mv, relu)
        yref, cache = O.batchnorm_train_fwd(ref, gamma.astype(np.float64), beta.astype(np.float64))
        assert_close(host(z), ref, name="z")
        assert_close(host(a), np.maximum(yref, 0) if relu else yref, 2e-5, 1e-4, "a")
        assert_close(host(mean), ref.mean(0), 2e-5, 1e-4, "mean")
        assert_close(host(invstd), 1 / np.sqrt(ref.var(0) + 1e-3), 2e-5, 1e-4, "invstd")
        assert_close(host(mm), 0.01 * ref.mean(0), 2e-5, 1e-4, "moving mean")
        assert_close(host(mv), 0.99 + 0.01 * ref.var(0) * rows / (rows - 1), 2e-5, 1e-4, "moving var")
        # backward: dy [rows][k2] . w2^T [n][k2] = d a of this layer, then its BN (+ReLU) backward
        k2 = 64
        dy = rs.randn(rows, k2).astype(np.float32)
        w2 = (rs.randn(n, k2) / 8).astype(np.float32)
        da = dy.astype(np.float64) @ w2.astype(np.float64).T + kf[:, None] * xrow
        dyy = da * (yref > 0) if relu else da
        dz_ref, dg_ref, db_ref = O.batchnorm_train_bwd(dyy, cache, gamma.astype(np.float64))
        dz, dg, db, dbias = ops.segment_dgrad_bn_backward(dev(dy), dev(w2), z, dev(gamma), mean, invstd, scale, shift, relu,
                                                          row_term=(dev(coef), dev(norm), dev(xrow)))
        assert_close(host(dz), dz_ref, 2e-5, 2e-4, "dz")
        assert_close(host(dg), dg_ref, 2e-5, 1e-4, "dgamma")
        assert_close(host(db), db_ref, 2e-5, 1e-4, "dbeta")
        assert np.abs(host(dbias)).max() <= 1e-4 * max(np.abs(dz_ref).sum(axis=0).max(), 1e-30)
    from tf_kaldi_speaker_amd import ops as m
    assert int(m._TICKETS[str(z.device)].abs().sum().item()) == 0
