"""Op-level Python wrappers over the C-ABI (include/xvector_hip.h, "Op level").

Inputs/outputs are torch CUDA tensors used purely as device buffers; every
function enqueues HIP kernels on torch's current stream.  No arithmetic is done
by torch here.  These are what the per-kernel parity tests call.
"""
import ctypes as C

import torch

try:
    from . import _lib
except ImportError:
    import _lib

TILE_M = 128


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


_WS = {}


def workspace(device, nbytes=256 << 20):
    key = (str(device), nbytes)
    if key not in _WS:
        _WS[key] = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    return _WS[key]


def _ws(t):
    w = workspace(t.device)
    return _p(w), C.c_size_t(w.numel() * 4)


def pad_channels(x2d, c_dst):
    rows, c = x2d.shape
    out = _f32((rows, c_dst), x2d)
    _lib.call("xv_pad_channels", _s(), _p(x2d), rows, c, _p(out), c_dst)
    return out


def cm_decode(packed, b, t, d):
    """packed: uint8 device tensor of b chunks (xvector_io.h packed layout) -> [b, t, d] float32 (Kaldi 'CM ' decode on the GPU)."""
    out = torch.empty((b, t, d), dtype=torch.float32, device=packed.device)
    stride = (8 + 8 * d + d * t + 15) // 16 * 16
    _lib.call("xv_cm_decode", _s(), _p(packed), b, t, d, C.c_size_t(stride), _p(out))
    return out


def prep_weight_fwd(kernel, c_pad):
    """kernel: [k, C, O] -> [O, k*c_pad]"""
    k, c, o = kernel.shape
    wt = _f32((o, k * c_pad), kernel)
    _lib.call("xv_prep_weight_fwd", _s(), _p(kernel), k, c, o, _p(wt), c_pad)
    return wt


def prep_weight_dgrad(kernel):
    """kernel: [k, C, O] -> [C, k*O] (taps flipped)"""
    k, c, o = kernel.shape
    wf = _f32((c, k * o), kernel)
    _lib.call("xv_prep_weight_dgrad", _s(), _p(kernel), k, c, o, _p(wf))
    return wf


def affine_forward(x, k, wt, bias, o, with_stats=False):
    """x: [segs, t_in, c_pad] -> z [segs*(t_in-k+1), o] (+ bn_part)."""
    segs, t_in, c_pad = x.shape
    rows = segs * (t_in - k + 1)
    z = _f32((rows, o), x)
    part = _f32((4, (rows + TILE_M - 1) // TILE_M, o), x) if with_stats else None
    wp, wb = _ws(x)
    _lib.call("xv_affine_forward", _s(), _p(x), segs, t_in, c_pad, k, _p(wt), _p(bias), _p(z), o, o, _p(part), wp, wb)
    return (z, part) if with_stats else z


def affine_dgrad(dz_pad, segs, t_out, o, k, wf, c):
    dx = _f32((segs * (t_out + k - 1), c), dz_pad)
    wp, wb = _ws(dz_pad)
    _lib.call("xv_affine_dgrad", _s(), _p(dz_pad), segs, t_out, o, k, _p(wf), _p(dx), c, wp, wb)
    return dx


def affine_wgrad(x, k, c, dz, dz_seg_pitch, dz_row0, o, kernel, l2_scale):
    segs, t_in, c_pad = x.shape
    dk = _f32((k, c, o), x)
    wp, wb = _ws(x)
    _lib.call("xv_affine_wgrad", _s(), _p(x), segs, t_in, c_pad, k, c, _p(dz), dz_seg_pitch, dz_row0, o, _p(kernel),
              float(l2_scale), _p(dk), wp, wb)
    return dk


def colsum(a):
    rows, n = a.shape
    out = _f32((n,), a)
    wp, wb = _ws(a)
    _lib.call("xv_colsum", _s(), _p(a), rows, n, a.stride(0), _p(out), wp, wb)
    return out


def col_stats(z):
    rows, n = z.shape
    part = _f32((4, (rows + TILE_M - 1) // TILE_M, n), z)
    _lib.call("xv_col_stats", _s(), _p(z), rows, n, z.stride(0), _p(part))
    return part


def bn_finalize(part, rows, gamma, beta, eps, momentum, unbiased, moving_mean, moving_var, with_range=False, relu=True):
    n = gamma.numel()
    mean, invstd, scale, shift = (_f32((n,), gamma) for _ in range(4))
    zmin = _f32((n,), gamma) if with_range else None
    zmax = _f32((n,), gamma) if with_range else None
    amax = torch.zeros(1, dtype=torch.int32, device=gamma.device) if with_range else None
    _lib.call("xv_bn_finalize", _s(), _p(part), rows, n, _p(gamma), _p(beta), float(eps), float(momentum), int(unbiased),
              _p(moving_mean), _p(moving_var), _p(mean), _p(invstd), _p(scale), _p(shift), _p(zmin), _p(zmax), _p(amax), int(relu))
    if with_range:
        return mean, invstd, scale, shift, zmin, zmax, amax.view(torch.float32)
    return mean, invstd, scale, shift


def bn_inference_scale(gamma, beta, moving_mean, moving_var, eps):
    n = gamma.numel()
    scale, shift = _f32((n,), gamma), _f32((n,), gamma)
    _lib.call("xv_bn_inference_scale", _s(), n, _p(gamma), _p(beta), _p(moving_mean), _p(moving_var), float(eps), _p(scale), _p(shift))
    return scale, shift


def bn_apply(z, scale, shift, relu):
    rows, n = z.shape
    a = _f32((rows, n), z)
    _lib.call("xv_bn_apply", _s(), _p(z), rows, n, n, _p(scale), _p(shift), int(relu), _p(a), n)
    return a


def bn_relu_backward(da, z, segs, t, gamma, mean, invstd, scale, shift, relu, pad, with_dbias=False):
    n = z.shape[1]
    dz = _f32((segs * (t + 2 * pad), n), z)
    dgamma, dbeta = _f32((n,), z), _f32((n,), z)
    dbias = _f32((n,), z) if with_dbias else None
    wp, wb = _ws(z)
    _lib.call("xv_bn_relu_backward", _s(), _p(da), _p(z), segs, t, n, _p(gamma), _p(mean), _p(invstd), _p(scale), _p(shift),
              int(relu), int(pad), _p(dz), _p(dgamma), _p(dbeta), _p(dbias), wp, wb)
    return (dz, dgamma, dbeta, dbias) if with_dbias else (dz, dgamma, dbeta)


class activation:
    """`with ops.activation(slope, dalpha=None):` - the op-level calls inside take y > 0 ? y : slope[c] * y (prelu / lrelu,
    tdnn.py:24-30) wherever their `relu` flag is set, through the ABI's xv_set_activation; plain ReLU again on exit."""

    def __init__(self, slope, dalpha=None):
        self.slope, self.dalpha = slope, dalpha

    def __enter__(self):
        _lib.call("xv_set_activation", _p(self.slope), _p(self.dalpha))
        return self

    def __exit__(self, *exc):
        _lib.call("xv_set_activation", None, None)
        return False


def prelu_forward(x2d, alpha):
    """x > 0 ? x : alpha[c] * x over the last axis."""
    y = torch.empty_like(x2d)
    _lib.call("xv_prelu_forward", _s(), _p(x2d), x2d.shape[0], x2d.shape[1], _p(alpha), _p(y))
    return y


def relu_backward(da, a):
    dz = torch.empty_like(da)
    _lib.call("xv_relu_backward", _s(), _p(da), _p(a), C.c_size_t(da.numel()), _p(dz))
    return dz


def stat_pool_forward(x):
    b, t, c = x.shape
    out = _f32((b, 2 * c), x)
    _lib.call("xv_stat_pool_forward", _s(), _p(x), b, t, c, _p(out))
    return out


def stat_pool_forward_bn(z, b, t, scale, shift, relu=True, weights=None):
    """[b, 2c] statistics of relu?(z*scale+shift) without materialising it (z: [b*t, c]); weights [b, t]: self-attention."""
    c = z.shape[1]
    out = _f32((b, 2 * c), z)
    _lib.call("xv_stat_pool_forward_bn", _s(), _p(z), b, t, c, _p(scale), _p(shift), int(relu), _p(weights), _p(out))
    return out


def stat_pool_forward_bn_aux(z, b, t, scale, shift, relu=True, weights=None):
    """stat_pool_forward_bn that also returns wpos, amax [b, c] (see xv_stat_pool_forward_bn_aux)."""
    c = z.shape[1]
    out, wpos, amax = _f32((b, 2 * c), z), _f32((b, c), z), _f32((b, c), z)
    _lib.call("xv_stat_pool_forward_bn_aux", _s(), _p(z), b, t, c, _p(scale), _p(shift), int(relu), _p(weights), _p(out), _p(wpos), _p(amax))
    return out, wpos, amax


def att_score(zk, act, query, scale):
    rows, n = zk.shape
    score = _f32((rows,), zk)
    _lib.call("xv_att_score", _s(), _p(zk), rows, n, n, int(act), _p(query), float(scale), _p(score))
    return score


def key_activation(z, act):
    """act(z) with the key-layer codes: 0 identity, 1 relu, 3 tanh."""
    y = torch.empty_like(z)
    _lib.call("xv_key_activation", _s(), _p(z), C.c_size_t(z.numel()), int(act), _p(y))
    return y


def softmax_segments(score, b, t):
    w = _f32((b, t), score)
    _lib.call("xv_softmax_segments", _s(), _p(score), b, t, _p(w))
    return w


def softmax_segments_backward(w, dw):
    b, t = w.shape
    ds = _f32((b, t), w)
    _lib.call("xv_softmax_segments_backward", _s(), _p(w), _p(dw), b, t, _p(ds))
    return ds


def att_pool_backward_weights(z, b, t, scale, shift, relu, pool_out, dpool):
    c = z.shape[1]
    dw = _f32((b, t), z)
    _lib.call("xv_att_pool_backward_weights", _s(), _p(z), b, t, c, _p(scale), _p(shift), int(relu), _p(pool_out), _p(dpool), _p(dw))
    return dw


def att_key_backward(zk, act, query, scale, dscore):
    rows, n = zk.shape
    dzk, dq, db = _f32((rows, n), zk), _f32((n,), zk), _f32((n,), zk)
    wp, wb = _ws(zk)
    _lib.call("xv_att_key_backward", _s(), _p(zk), rows, n, int(act), _p(query), float(scale), _p(dscore), _p(dzk), _p(dq), _p(db), wp, wb)
    return dzk, dq, db


def bn_relu_backward_pooled(pool_out, dpool, b, t, z, gamma, mean, invstd, scale, shift, relu=True, weights=None):
    """BN(+ReLU) backward of the layer feeding statistics pooling, upstream gradient = pooling backward on the fly."""
    n = z.shape[1]
    dz = _f32((b * t, n), z)
    dgamma, dbeta, dbias = _f32((n,), z), _f32((n,), z), _f32((n,), z)
    wp, wb = _ws(z)
    _lib.call("xv_bn_relu_backward_pooled", _s(), _p(pool_out), _p(dpool), _p(weights), b, t, _p(z), n, _p(gamma), _p(mean), _p(invstd), _p(scale),
              _p(shift), int(relu), _p(dz), _p(dgamma), _p(dbeta), _p(dbias), wp, wb)
    return dz, dgamma, dbeta, dbias


def bn_relu_backward_pooled_aux(pool_out, dpool, wpos, b, t, z, gamma, mean, invstd, scale, shift, relu=True, weights=None):
    """bn_relu_backward_pooled with the reductions in closed form from (pool_out, dpool, wpos): no reduction pass over z."""
    n = z.shape[1]
    dz = _f32((b * t, n), z)
    dgamma, dbeta, dbias = _f32((n,), z), _f32((n,), z), _f32((n,), z)
    wp, wb = _ws(z)
    _lib.call("xv_bn_relu_backward_pooled_aux", _s(), _p(pool_out), _p(dpool), _p(weights), _p(wpos), b, t, _p(z), n, _p(gamma), _p(mean),
              _p(invstd), _p(scale), _p(shift), int(relu), _p(dz), _p(dgamma), _p(dbeta), _p(dbias), wp, wb)
    return dz, dgamma, dbeta, dbias


def stat_pool_backward(x, out, dout):
    b, t, c = x.shape
    dx = torch.empty_like(x)
    _lib.call("xv_stat_pool_backward", _s(), _p(x), _p(out), _p(dout), b, t, c, _p(dx))
    return dx


def l2_scaling_forward(x, factor):
    y = torch.empty_like(x)
    _lib.call("xv_l2_scaling_forward", _s(), _p(x), x.shape[0], x.shape[1], float(factor), _p(y))
    return y


def l2_scaling_backward(x, dy, factor):
    dx = torch.empty_like(x)
    _lib.call("xv_l2_scaling_backward", _s(), _p(x), _p(dy), x.shape[0], x.shape[1], float(factor), _p(dx))
    return dx


def loss_prep_weight(w, normalize):
    c, n = w.shape
    ldn = (n + 3) // 4 * 4
    inv = _f32((n,), w)
    wn = _f32((c, ldn), w)
    wnt = _f32((n, c), w)
    _lib.call("xv_loss_prep_weight", _s(), _p(w), c, n, int(normalize), _p(inv), _p(wn), ldn, _p(wnt))
    return inv, wn, wnt


def margin_softmax_rows(kind, logits, n, x, labels, m, lam):
    """logits: [rows, ldl] (ldl >= n).  Returns loss (1-elem tensor), dlogits, dnorm, row_loss."""
    rows, ldl = logits.shape
    dlogits = torch.empty_like(logits)
    dnorm, row_loss, loss = _f32((rows,), logits), _f32((rows,), logits), _f32((1,), logits)
    _lib.call("xv_margin_softmax_rows", _s(), int(kind), _p(logits), rows, n, ldl, _p(x), x.shape[1], _p(labels), float(m),
              float(lam), _p(dlogits), _p(dnorm), _p(row_loss), _p(loss))
    return loss, dlogits, dnorm, row_loss


def add_norm_grad(x, dnorm, dx):
    _lib.call("xv_add_norm_grad", _s(), _p(x), _p(dnorm), x.shape[0], x.shape[1], _p(dx))
    return dx


_TICKETS = {}


def _tickets(like, n):
    """Zeroed uint32 tickets for the segment-level launches (one per 32 output columns; every launch leaves them zero)."""
    key = str(like.device)
    need = (n + 31) // 32 + 8
    if key not in _TICKETS or _TICKETS[key].numel() < need:
        _TICKETS[key] = torch.zeros(max(need, 1024), dtype=torch.int32, device=like.device)
    return _TICKETS[key]


def _row_term(row_term):
    if row_term is None:
        return _p(None), _p(None), _p(None), 0
    coef, norm, xrow = row_term
    return _p(coef), _p(norm), _p(xrow), xrow.shape[1]


def segment_gemm(a, bt, bias=None, row_term=None):
    """c[m][n] = sum_k a[m][k] bt[n][k] + bias[n] (+ (coef[m]/norm[m]) xrow[m][n]); m <= 128 rows, one launch."""
    m, k = a.shape
    n = bt.shape[0]
    c = _f32((m, n), a)
    wp, wb = _ws(a)
    rc, rn, rx, ldx = _row_term(row_term)
    _lib.call("xv_segment_gemm", _s(), _p(a), a.stride(0), _p(bt), bt.stride(0), m, n, k, _p(bias), rc, rn, rx, ldx, _p(c), n,
              wp, wb, _p(_tickets(a, n)))
    return c


def segment_affine_bn_forward(x, wt, bias, gamma, beta, eps, momentum, unbiased, moving_mean, moving_var, relu, want_a=True):
    """dense + training-mode BatchNorm (+ ReLU) on m <= 128 rows in one launch.  Returns z, a, mean, invstd, scale, shift."""
    m, k = x.shape
    n = wt.shape[0]
    z = _f32((m, n), x)
    a = _f32((m, n), x) if want_a else None
    mean, invstd, scale, shift = (_f32((n,), x) for _ in range(4))
    wp, wb = _ws(x)
    _lib.call("xv_segment_affine_bn_forward", _s(), _p(x), x.stride(0), _p(wt), wt.stride(0), m, n, k, _p(bias), _p(gamma), _p(beta),
              float(eps), float(momentum), int(unbiased), _p(moving_mean), _p(moving_var), _p(z), _p(mean), _p(invstd), _p(scale),
              _p(shift), int(relu), _p(a), wp, wb, _p(_tickets(x, n)))
    return z, a, mean, invstd, scale, shift


def segment_dgrad_bn_backward(dy, wt, z, gamma, mean, invstd, scale, shift, relu, row_term=None):
    """d a = dy . wt^T (+ row term), then the BatchNorm (+ ReLU) backward of the layer with pre-BN tensor z.  Returns dz, dgamma, dbeta, dbias."""
    m, k = dy.shape
    n = wt.shape[0]
    dz = _f32((m, n), dy)
    dgamma, dbeta, dbias = (_f32((n,), dy) for _ in range(3))
    wp, wb = _ws(dy)
    rc, rn, rx, ldx = _row_term(row_term)
    _lib.call("xv_segment_dgrad_bn_backward", _s(), _p(dy), dy.stride(0), _p(wt), wt.stride(0), m, n, k, rc, rn, rx, ldx, _p(z),
              _p(gamma), _p(mean), _p(invstd), _p(scale), _p(shift), int(relu), _p(dz), _p(dgamma), _p(dbeta), _p(dbias), wp, wb,
              _p(_tickets(dy, n)))
    return dz, dgamma, dbeta, dbias


def loss_weight_backward(dwn, wn, inv, w, normalize, l2_scale):
    c, n = w.shape
    dw = torch.empty_like(w)
    wp, wb = _ws(w)
    _lib.call("xv_loss_weight_backward", _s(), _p(dwn), dwn.stride(0), _p(wn), wn.stride(0), _p(inv), _p(w), c, n, int(normalize),
              float(l2_scale), _p(dw), wp, wb)
    return dw


def ring_loss(x, r, lam):
    """lam * mean((||x|| - r)^2) as a device scalar (loss.py:1003-1017)."""
    rows, n = x.shape
    out, dn, dr = _f32((1,), x).zero_(), _f32((rows,), x).zero_(), _f32((1,), x)
    _lib.call("xv_ring_loss", _s(), _p(x), rows, n, n, _p(r), float(lam), _p(out), _p(dn), _p(dr))
    return out[0]


def mhe_loss(wn, n, labels, lam):
    """lam / (mean_{b,n}(2 - 2 wn[:,y_b].wn[:,n]) + 1e-6) on column-normalised weights wn [c, ldn] (loss.py:1018-1033)."""
    c, ldn = wn.shape
    out, coef = _f32((1,), wn).zero_(), _f32((1 + 2 * c,), wn)
    counts = torch.empty(n, dtype=torch.int32, device=wn.device)
    _lib.call("xv_mhe_loss", _s(), _p(wn), c, n, ldn, _p(labels), labels.shape[0], float(lam), _p(out), _p(coef), _p(counts))
    return out[0]


def l2_reg_loss(w, scale, accum):
    _lib.call("xv_l2_reg_loss", _s(), _p(w), C.c_size_t(w.numel()), float(scale), _p(accum))


def sumsq(g, accum):
    _lib.call("xv_sumsq", _s(), _p(g), C.c_size_t(g.numel()), _p(accum))


def sgd_update(p, g, lr, grad_scale=1.0):
    _lib.call("xv_sgd_update", _s(), _p(p), _p(g), C.c_size_t(p.numel()), float(lr), float(grad_scale))


def momentum_update(p, g, acc, lr, momentum, nesterov, grad_scale=1.0):
    _lib.call("xv_momentum_update", _s(), _p(p), _p(g), _p(acc), C.c_size_t(p.numel()), float(lr), float(momentum), int(nesterov),
              float(grad_scale))


def adam_update(p, g, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    _lib.call("xv_adam_update", _s(), _p(p), _p(g), _p(m), _p(v), C.c_size_t(p.numel()), float(lr), float(beta1), float(beta2),
              float(eps), int(t), float(grad_scale))


# ---- split precision (f16x3): fp32 tensors as two fp16 planes + a device-side max |x| -------------------------
class Planes(object):
    """[2][rows][ld] fp16 planes of an fp32 matrix plus the uint32 float-bits of its max |x| (device)."""

    def __init__(self, data, rows, ld, amax):
        self.data, self.rows, self.ld, self.amax = data, rows, ld, amax

    @property
    def stride(self):
        return self.rows * self.ld


def amax_of(x):
    a = torch.zeros(1, dtype=torch.int32, device=x.device)
    _lib.call("xv_amax", _s(), _p(x), C.c_size_t(x.numel()), _p(a))
    return a


def split_planes(x2d, amax=None):
    rows, c = x2d.shape
    ld = (c + 7) // 8 * 8
    if amax is None:
        amax = amax_of(x2d)
    data = torch.empty((2, rows, ld), dtype=torch.int16, device=x2d.device)
    _lib.call("xv_split_planes", _s(), _p(x2d), rows, c, x2d.stride(0), _p(data), ld, C.c_size_t(rows * ld), _p(amax))
    return Planes(data, rows, ld, amax)


def bn_apply_split(z, scale, shift, relu, amax):
    rows, n = z.shape
    ld = (n + 7) // 8 * 8
    data = torch.empty((2, rows, ld), dtype=torch.int16, device=z.device)
    _lib.call("xv_bn_apply_split", _s(), _p(z), rows, n, n, _p(scale), _p(shift), int(relu), _p(amax), _p(data), ld, C.c_size_t(rows * ld))
    return Planes(data, rows, ld, amax)


def bn_relu_backward_split(da, z, segs, t, gamma, mean, invstd, scale, shift, zmin, zmax, relu, pad):
    n = z.shape[1]
    ld = (n + 7) // 8 * 8
    rows = segs * (t + 2 * pad)
    data = torch.empty((2, rows, ld), dtype=torch.int16, device=z.device)
    amax = torch.zeros(1, dtype=torch.int32, device=z.device)
    dgamma, dbeta, dbias = _f32((n,), z), _f32((n,), z), _f32((n,), z)
    wp, wb = _ws(z)
    _lib.call("xv_bn_relu_backward_split", _s(), _p(da), _p(z), segs, t, n, _p(gamma), _p(mean), _p(invstd), _p(scale), _p(shift),
              _p(zmin), _p(zmax), int(relu), int(pad), _p(data), ld, C.c_size_t(rows * ld), _p(amax), _p(dgamma), _p(dbeta), _p(dbias),
              wp, wb)
    return Planes(data, rows, ld, amax), dgamma, dbeta, dbias


def bn_relu_backward_pooled_split(pool_out, dpool, b, t, z, gamma, mean, invstd, scale, shift, zmin, zmax, relu=True, weights=None):
    n = z.shape[1]
    ld = (n + 7) // 8 * 8
    rows = b * t
    data = torch.empty((2, rows, ld), dtype=torch.int16, device=z.device)
    amax = torch.zeros(1, dtype=torch.int32, device=z.device)
    dgamma, dbeta, dbias = _f32((n,), z), _f32((n,), z), _f32((n,), z)
    wp, wb = _ws(z)
    _lib.call("xv_bn_relu_backward_pooled_split", _s(), _p(pool_out), _p(dpool), _p(weights), b, t, _p(z), n, _p(gamma), _p(mean), _p(invstd), _p(scale),
              _p(shift), _p(zmin), _p(zmax), int(relu), _p(data), ld, C.c_size_t(rows * ld), _p(amax), _p(dgamma), _p(dbeta), _p(dbias),
              wp, wb)
    return Planes(data, rows, ld, amax), dgamma, dbeta, dbias


def affine_forward_f16x3(xp, segs, t_in, k, wtp, bias, o, with_stats=False):
    """xp: Planes of x [segs*t_in][c_ld]; wtp: Planes of Wt [o][k*c_ld]."""
    rows = segs * (t_in - k + 1)
    z = torch.empty((rows, o), dtype=torch.float32, device=xp.data.device)
    part = torch.empty((4, (rows + TILE_M - 1) // TILE_M, o), dtype=torch.float32, device=z.device) if with_stats else None
    _lib.call("xv_affine_forward_f16x3", _s(), _p(xp.data), C.c_size_t(xp.stride), _p(xp.amax), segs, t_in, xp.ld, k, _p(wtp.data),
              C.c_size_t(wtp.stride), _p(wtp.amax), _p(bias), _p(z), o, o, _p(part))
    return (z, part) if with_stats else z


def affine_dgrad_f16x3(dzp, segs, t_out, k, wfp, c):
    dx = torch.empty((segs * (t_out + k - 1), c), dtype=torch.float32, device=dzp.data.device)
    _lib.call("xv_affine_dgrad_f16x3", _s(), _p(dzp.data), C.c_size_t(dzp.stride), _p(dzp.amax), segs, t_out, dzp.ld, k, _p(wfp.data),
              C.c_size_t(wfp.stride), _p(wfp.amax), _p(dx), c)
    return dx


def affine_dgrad_bnstats_f16x3(dzp, segs, t_out, k, wfp, c, z_below, scale, shift, mean, invstd):
    """affine_dgrad_f16x3 + the per-tile BN-backward partials [ceil(rows/128), 3, c] of the layer that owns dx."""
    rows = segs * (t_out + k - 1)
    dx = torch.empty((rows, c), dtype=torch.float32, device=dzp.data.device)
    part = torch.empty(((rows + TILE_M - 1) // TILE_M, 3, c), dtype=torch.float32, device=dx.device)
    _lib.call("xv_affine_dgrad_bnstats_f16x3", _s(), _p(dzp.data), C.c_size_t(dzp.stride), _p(dzp.amax), segs, t_out, dzp.ld, k, _p(wfp.data),
              C.c_size_t(wfp.stride), _p(wfp.amax), _p(dx), c, _p(z_below), _p(scale), _p(shift), _p(mean), _p(invstd), _p(part))
    return dx, part


def affine_wgrad_f16x3(xp, segs, t_in, k, c, dzp, dz_seg_pitch, dz_row0, o, kernel, l2_scale):
    dk = torch.empty((k, c, o), dtype=torch.float32, device=xp.data.device)
    wp, wb = _ws(dk)
    _lib.call("xv_affine_wgrad_f16x3", _s(), _p(xp.data), C.c_size_t(xp.stride), _p(xp.amax), segs, t_in, xp.ld, k, c, _p(dzp.data),
              C.c_size_t(dzp.stride), _p(dzp.amax), dz_seg_pitch, dz_row0, dzp.ld, o, _p(kernel), float(l2_scale), _p(dk), wp, wb)
    return dk
