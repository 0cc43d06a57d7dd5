"""Split-precision (f16x3) kernels on a real MI355X against the float64 oracle: same shapes and the same
tolerances as the fp32-MFMA tests of test_gpu_ops.py (rel-Frobenius 5e-6, max 2e-5 of the tensor scale) -
the split path must be indistinguishable at that level.  Adversarial value ranges for the power-of-two
scaling: tiny gradients (1e-7), huge activations (1e3), all-zero tensors, outliers 1e4x the typical value."""
import numpy as np
import pytest
import torch

from oracle import xvector_oracle as O
from tests.test_gpu_ops import assert_close, dev, host, AFFINE_CASES, _pooled_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from tf_kaldi_speaker_amd import ops as m
    return m


def _bits_to_float(t):
    return float(t.view(torch.float32).cpu().numpy()[0])


def test_split_planes_reconstruct_and_scale(ops):
    rs = np.random.RandomState(0)
    for scale in (1.0, 1e-7, 1e3):
        x = (rs.randn(37, 30) * scale).astype(np.float32)
        x[3, 5] = 1e4 * scale                               # outlier sets the scale
        p = ops.split_planes(dev(x))
        assert p.ld == 32 and abs(_bits_to_float(p.amax) - np.abs(x).max()) == 0
        planes = p.data.cpu().numpy().view(np.float16).astype(np.float64)
        amax = np.abs(x).max()
        s = 2.0 ** (12 - np.floor(np.log2(amax)))
        rec = (planes[0] + planes[1])[:, :30] / s
        assert np.all(np.isfinite(planes)) and np.abs(planes).max() < 2 ** 13
        assert np.abs(rec - x).max() <= 2.0 ** -21 * amax     # 22-bit pieces relative to the scale
        assert np.all(planes[:, :, 30:] == 0)
    z = ops.split_planes(dev(np.zeros((4, 8), np.float32)))    # all-zero tensor: scale 1, planes 0
    assert np.all(z.data.cpu().numpy() == 0)


@pytest.mark.parametrize("segs,t_in,c,k,o", AFFINE_CASES[:5])
def test_affine_forward_dgrad_wgrad_f16x3(ops, segs, t_in, c, k, o):
    rs = np.random.RandomState(segs * 31 + k)
    t_out = t_in - k + 1
    x = np.maximum(rs.randn(segs, t_in, c), 0).astype(np.float32) * 1.7          # ReLU-like activations
    kern = (rs.randn(k, c, o) / np.sqrt(k * c)).astype(np.float32)
    bias = rs.randn(o).astype(np.float32)
    dz = (rs.randn(segs, t_out, o) * 3e-4).astype(np.float32)                    # gradient-like magnitudes
    x64, k64 = x.astype(np.float64), kern.astype(np.float64)
    ref = O.conv1d_valid_fwd(x64, k64, bias.astype(np.float64)).reshape(-1, o)
    dx_ref, dk_ref, _ = O.conv1d_valid_bwd(x64, k64, dz.astype(np.float64))

    xp = ops.split_planes(dev(x.reshape(-1, c)))
    c_ld = xp.ld
    wt = ops.prep_weight_fwd(dev(kern), c_ld)                                    # [o][k*c_ld] fp32 kernel layout
    wtp = ops.split_planes(wt)
    z, part = ops.affine_forward_f16x3(xp, segs, t_in, k, wtp, dev(bias), o, with_stats=True)
    assert_close(host(z), ref, name="affine_forward_f16x3")
    rows = ref.shape[0]
    if rows >= 2:
        gamma, beta = rs.rand(o).astype(np.float32) + 0.5, rs.randn(o).astype(np.float32)
        mean, invstd, scale, shift, zmin, zmax, amax = ops.bn_finalize(part, rows, dev(gamma), dev(beta), 1e-3, 0.99, False, None, None,
                                                                        with_range=True, relu=True)
        assert_close(host(mean), ref.mean(0), 2e-5, 1e-4, "bn mean")
        assert_close(host(zmin), ref.min(0), 1e-5, 2e-5, "zmin")
        assert_close(host(zmax), ref.max(0), 1e-5, 2e-5, "zmax")
        y, _ = O.batchnorm_train_fwd(ref, gamma.astype(np.float64), beta.astype(np.float64))
        a_ref = np.maximum(y, 0)
        assert abs(float(host(amax)[0]) - a_ref.max()) <= 1e-4 * a_ref.max()     # exact output range from the epilogue min/max
        ap = ops.bn_apply_split(z, scale, shift, True, amax.view(torch.int32))
        planes = ap.data.cpu().numpy().view(np.float16).astype(np.float64)
        s = 2.0 ** (12 - np.floor(np.log2(float(host(amax)[0]))))
        assert np.abs(planes).max() < 2 ** 13 + 8
        assert_close((planes[0] + planes[1])[:, :o] / s, a_ref, 2e-5, 1e-4, "bn_apply_split")

    # backward: dz planes in the padded layout, tap-flipped weight planes
    pad = k - 1
    dzp_host = np.zeros((segs, t_out + 2 * pad, o), np.float32)
    dzp_host[:, pad:pad + t_out] = dz
    dzp = ops.split_planes(dev(dzp_host.reshape(-1, o)))
    if c % 8 == 0:
        if k > 1:
            wf = ops.prep_weight_dgrad(dev(kern))                                # [c][k*o]
        else:
            wf = dev(kern[0])
        o_ld = dzp.ld
        if o_ld != o:                                                            # re-pitch rows of wf to k*o_ld
            wf = torch.nn.functional.pad(wf.view(c, k, o), (0, o_ld - o)).reshape(c, k * o_ld).contiguous()
        wfp = ops.split_planes(wf)
        dx = ops.affine_dgrad_f16x3(dzp, segs, t_out, k, wfp, c)
        assert_close(host(dx), dx_ref.reshape(-1, c), name="affine_dgrad_f16x3")
    dk = ops.affine_wgrad_f16x3(xp, segs, t_in, k, c, dzp, t_out + 2 * pad, pad, o, dev(kern), 1e-2)
    assert_close(host(dk), dk_ref + 1e-2 * k64, name="affine_wgrad_f16x3")


def test_gemm16_identity_asymmetric(ops):
    """A = I with an asymmetric B: exact in the split representation (integers < 2^11 per piece)."""
    n = 256
    x = np.eye(n, dtype=np.float32)
    kern = (np.arange(n)[:, None] * 7 + np.arange(n)[None, :] % 5).astype(np.float32)[None]
    xp = ops.split_planes(dev(x))
    wtp = ops.split_planes(ops.prep_weight_fwd(dev(kern), n))
    z = ops.affine_forward_f16x3(xp, n, 1, 1, wtp, None, n)
    assert np.array_equal(host(z), kern[0].astype(np.float64))


@pytest.mark.parametrize("relu,pad,n", [(1, 0, 512), (1, 6, 512), (1, 0, 1500)])
def test_bn_relu_backward_split(ops, relu, pad, n):
    rs = np.random.RandomState(5 + pad + n)
    segs, t = 6, 37
    z = (rs.randn(segs * t, n) * 2 + 0.5).astype(np.float32)
    da = (rs.randn(segs * t, n) * 1e-5).astype(np.float32)                       # tiny upstream gradient
    da[7, 11] = 3e-2                                                             # with one large entry
    gamma, beta = (rs.rand(n) + 0.5).astype(np.float32), (0.3 * rs.randn(n)).astype(np.float32)
    y, cache = O.batchnorm_train_fwd(z.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64))
    dy = da.astype(np.float64) * (y > 0) if relu else da.astype(np.float64)
    dz_ref, dg_ref, db_ref = O.batchnorm_train_bwd(dy, cache, gamma.astype(np.float64))
    part = ops.col_stats(dev(z))
    mean, invstd, scale, shift, zmin, zmax, _ = ops.bn_finalize(part, segs * t, dev(gamma), dev(beta), 1e-3, 0.99, False, None, None,
                                                                with_range=True)
    dzp, dg, db, dbias = ops.bn_relu_backward_split(dev(da), dev(z), segs, t, dev(gamma), mean, invstd, scale, shift, zmin, zmax, relu, pad)
    bound = _bits_to_float(dzp.amax)
    true_max = np.abs(dz_ref).max()
    assert true_max <= bound <= 64 * true_max, (true_max, bound)                  # a true upper bound, not absurdly loose
    planes = dzp.data.cpu().numpy().view(np.float16).astype(np.float64)
    assert np.all(np.isfinite(planes))
    s = 2.0 ** (12 - np.floor(np.log2(bound)))
    rec = ((planes[0] + planes[1]) / s).reshape(segs, t + 2 * pad, dzp.ld)
    if pad:
        assert np.all(rec[:, :pad] == 0) and np.all(rec[:, pad + t:] == 0)
    assert np.all(rec[:, :, n:] == 0)
    assert_close(rec[:, pad:pad + t, :n].reshape(-1, n), dz_ref, 2e-5, 2e-4, "dz planes")
    assert_close(host(dg), dg_ref, 2e-5, 1e-4, "dgamma")
    assert_close(host(db), db_ref, 2e-5, 1e-4, "dbeta")


def test_pooling_fused_into_bn_split(ops):
    b, t, n = 4, 37, 1500
    z, gamma, beta, dout, pool_ref, dz_ref, dg_ref, db_ref = _pooled_case(23, b, t, n)
    part = ops.col_stats(dev(z))
    mean, invstd, scale, shift, zmin, zmax, _ = ops.bn_finalize(part, b * t, dev(gamma), dev(beta), 1e-3, 0.99, False, None, None,
                                                                with_range=True)
    pool = ops.stat_pool_forward_bn(dev(z), b, t, scale, shift, True)
    dzp, dg, db, dbias = ops.bn_relu_backward_pooled_split(pool, dev(dout), b, t, dev(z), dev(gamma), mean, invstd, scale, shift, zmin, zmax)
    bound = _bits_to_float(dzp.amax)
    assert np.abs(dz_ref).max() <= bound
    planes = dzp.data.cpu().numpy().view(np.float16).astype(np.float64)
    s = 2.0 ** (12 - np.floor(np.log2(bound)))
    rec = (planes[0] + planes[1]) / s
    assert np.all(rec[:, n:] == 0)
    assert_close(rec[:, :n], dz_ref, 2e-5, 2e-4, "pooled split dz")
    assert_close(host(dg), dg_ref, 2e-5, 1e-4, "pooled split dgamma")
    assert_close(host(db), db_ref, 2e-5, 1e-4, "pooled split dbeta")


@pytest.mark.parametrize("segs,t_out,k", [(5, 57, 5), (3, 300, 1)])
def test_dgrad_epilogue_produces_the_bn_backward_partials(ops, segs, t_out, k):
    """xv_affine_dgrad_bnstats_f16x3: dx identical to the plain data gradient, partials = the sums the BN backward needs
    (sum dd, sum dd*xhat, max |dd| per 128-row tile), checked against float64 on the kernel's own dx."""
    rs = np.random.RandomState(k)
    c, o = 512, 512
    rows = segs * (t_out + k - 1)
    kern = (rs.randn(k, c, o) / np.sqrt(k * c)).astype(np.float32)
    pad = k - 1
    dzp_host = np.zeros((segs, t_out + 2 * pad, o), np.float32)
    dzp_host[:, pad:pad + t_out] = (rs.randn(segs, t_out, o) * 1e-3).astype(np.float32)
    dzp = ops.split_planes(dev(dzp_host.reshape(-1, o)))
    wf = ops.prep_weight_dgrad(dev(kern)) if k > 1 else dev(kern[0])
    wfp = ops.split_planes(wf)
    z = (rs.randn(rows, c) * 2 + 0.3).astype(np.float32)                       # pre-BN output of the layer that owns dx
    gamma, beta = (rs.rand(c) + 0.5).astype(np.float32), (0.3 * rs.randn(c)).astype(np.float32)
    part_z = ops.col_stats(dev(z))
    mean, invstd, scale, shift, zmin, zmax, _ = ops.bn_finalize(part_z, rows, dev(gamma), dev(beta), 1e-3, 0.99, False, None, None,
                                                                with_range=True)
    dx_plain = ops.affine_dgrad_f16x3(dzp, segs, t_out, k, wfp, c)
    dx, part = ops.affine_dgrad_bnstats_f16x3(dzp, segs, t_out, k, wfp, c, dev(z), scale, shift, mean, invstd)
    # same products, possibly another accumulation grouping (the plain launch may take the 16x16x32 context-window kernel, the
    # epilogue launch the 32x32x16 one): equal to fp32 rounding, not bit for bit
    assert float((dx - dx_plain).abs().max()) <= 2e-6 * float(dx_plain.abs().max())
    dxh, zh = host(dx), z.astype(np.float64)
    sc, sh, mu, istd = host(scale), host(shift), host(mean), host(invstd)
    dd = dxh * ((zh * sc + sh) > 0)
    xh = (zh - mu) * istd
    p = host(part)
    tiles = (rows + 127) // 128
    for t in range(tiles):
        r = slice(128 * t, min(rows, 128 * t + 128))
        assert_close(p[t, 0], dd[r].sum(axis=0), 2e-5, 2e-4, "sum dd tile %d" % t)
        assert_close(p[t, 1], (dd[r] * xh[r]).sum(axis=0), 2e-5, 2e-4, "sum dd*xhat tile %d" % t)
        assert_close(p[t, 2], np.abs(dd[r]).max(axis=0), 1e-6, 1e-6, "max |dd| tile %d" % t)


def test_context_window_kernel_256_row_tiles():
    """The 256-row-tile build of the context-window kernel (XV_CONV_WR=4, off by default because it measured slower at
    S1) stays parity-clean: rerun the affine cases and the fused-statistics test with it forced.  The switch is read once
    per process, hence the child process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, XV_CONV_WR="4")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_gpu_ops_f16x3.py"),
                        "-k", "test_affine_forward_dgrad_wgrad_f16x3 or test_dgrad_epilogue"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "passed" in r.stdout
