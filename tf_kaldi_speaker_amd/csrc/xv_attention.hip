// Self-attention pooling (model/pooling.py:37-192) in the shipped single-head form (nnet_conf/*_tdnn4_att.json):
//   key   = tanh?(dense(relu(bn(dense(tdnn4_relu)))))      - the two dense layers run on the frame-level GEMM kernels
//   score = key . query / sqrt(dk)                         - att_score_kernel (one wave per frame)
//   w     = softmax over the frames of a chunk             - softmax_segments_kernel
//   out   = [sum_t w v, sqrt(sum_t w (v - mean)^2)]        - stat_pool_fwd_kernel with weights (xv_elementwise.hip)
// and its backward pieces.  The value tensor v = relu(bn(z5)) is evaluated on the fly from tdnn5's pre-BN output, as
// for statistics pooling: neither v nor dv is ever written; dv enters tdnn5's BN backward through PoolGrad.w.
#include "xv_common.h"

#include <algorithm>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// att_key_network_type of the last key layer (pooling.py:84-96): 0 affine, 1 affine + relu, 3 affine + tanh
// (2 = affine + bn + relu runs through the BN kernels; these functions then see its output with act = 0)
__device__ __forceinline__ float key_act(float z, int act) { return act == 3 ? tanhf(z) : (act == 1 ? fmaxf(z, 0.f) : z); }

// score[r] = scale * sum_c act(zk[r][c]) * q[c]; block = 4 waves = 4 rows.  A lane takes column quads lane, lane + 64, ...: 16 bytes per
// lane and all of a row's loads (six for 1 500 columns) in flight before the first tanh - the 4-byte, one-load-at-a-time form of rounds 2-4
// read the 143 MB key tensor at 2 TB/s (71 us at S4).  Needs n % 4 == 0 and ldz % 4 == 0 (the launcher falls back to the scalar form).
#define ATT_SCORE_QUADS 8      // quads per lane held at once: rows of up to 2 048 columns in one pass
template <bool VEC>
__global__ __launch_bounds__(256) void att_score_kernel(const float* __restrict__ zk, int rows, int n, long ldz, int act,
                                                        const float* __restrict__ q, float scale, float* __restrict__ score) {
    XV_EW_PRIORITY();
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* z = zk + (long)row * ldz;
    float s = 0.f;
    if (VEC) {
        const int nq = n >> 2;
        for (int q0 = 0; q0 < nq; q0 += 64 * ATT_SCORE_QUADS) {
            f32x4 v[ATT_SCORE_QUADS], w[ATT_SCORE_QUADS];
#pragma unroll
            for (int u = 0; u < ATT_SCORE_QUADS; ++u) {
                const int qi = min(q0 + 64 * u + lane, nq - 1);      // (clamped: the loads stay unconditional, the sum is masked)
                v[u] = *(const f32x4*)(z + 4 * qi);
                w[u] = *(const f32x4*)(q + 4 * qi);
            }
#pragma unroll
            for (int u = 0; u < ATT_SCORE_QUADS; ++u) {
                if (q0 + 64 * u + lane < nq)
                    s += (key_act(v[u].x, act) * w[u].x + key_act(v[u].y, act) * w[u].y) + (key_act(v[u].z, act) * w[u].z + key_act(v[u].w, act) * w[u].w);
            }
        }
    } else {
        for (int c = lane; c < n; c += 64) s += key_act(z[c], act) * q[c];
    }
    s = wave_sum(s);
    if (lane == 0) score[row] = s * scale;
}

// w[b][t] = softmax_t(score[b][t]); one workgroup per chunk
// flen (batched extraction): chunk b has flen[b] - shrink valid frames; the padding behind them gets weight 0
__global__ __launch_bounds__(256) void softmax_segments_kernel(const float* __restrict__ score, int t, float* __restrict__ w,
                                                               const int* __restrict__ flen, int shrink) {
    XV_EW_PRIORITY();
    __shared__ float red[4];
    const float* s = score + (long)blockIdx.x * t;
    float* o = w + (long)blockIdx.x * t;
    if (flen) {
        const int tv = max(1, min(t, flen[blockIdx.x] - shrink));
        for (int i = tv + threadIdx.x; i < t; i += 256) o[i] = 0.f;
        t = tv;
    }
    float m = -INFINITY;
    for (int i = threadIdx.x; i < t; i += 256) m = fmaxf(m, s[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float z = 0.f;
    for (int i = threadIdx.x; i < t; i += 256) z += expf(s[i] - m);
    z = wave_sum(z);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = z;
    __syncthreads();
    z = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = threadIdx.x; i < t; i += 256) o[i] = expf(s[i] - m) / z;
}

// ds[b][t] = w (dw - sum_t' w dw)
__global__ __launch_bounds__(256) void softmax_segments_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dw, int t,
                                                                   float* __restrict__ ds) {
    XV_EW_PRIORITY();
    __shared__ float red[4];
    const long base = (long)blockIdx.x * t;
    float z = 0.f;
    for (int i = threadIdx.x; i < t; i += 256) z += w[base + i] * dw[base + i];
    z = wave_sum(z);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = z;
    __syncthreads();
    z = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = threadIdx.x; i < t; i += 256) ds[base + i] = w[base + i] * (dw[base + i] - z);
}

// dw[b][t] = sum_c ( dmean[c] * a + dvar[c] * (a - mean[c])^2 ),  a = relu?(z*scale + shift),
// dvar = dstd * 0.5 / std where the variance was not clamped (pooling.py:160-162).  One wave per frame.
__global__ __launch_bounds__(256) void att_pool_dw_kernel(const float* __restrict__ z, int rows, int t, int n,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                          const float* __restrict__ pool, const float* __restrict__ dpool,
                                                          float* __restrict__ dw, const float* __restrict__ slope) {
    XV_EW_PRIORITY();
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = row / t;
    const float* o = pool + (long)b * 2 * n;
    const float* g = dpool + (long)b * 2 * n;
    const float* zr = z + (long)row * n;
    const float sd_eps = 1e-6f;
    float s = 0.f;
    // all of a row's z loads first (six quads per lane for 1 500 columns), then the arithmetic: one load in flight per lane read the 143 MB
    // value tensor at 2.5 TB/s (57 us at S4)
    const int nq = n >> 2;
    for (int q0 = 0; q0 < nq; q0 += 64 * ATT_SCORE_QUADS) {
        f32x4 zq[ATT_SCORE_QUADS];
#pragma unroll
        for (int u = 0; u < ATT_SCORE_QUADS; ++u) zq[u] = *(const f32x4*)(zr + 4 * min(q0 + 64 * u + lane, nq - 1));
#pragma unroll
        for (int u = 0; u < ATT_SCORE_QUADS; ++u) {
            const int c = 4 * (q0 + 64 * u + lane);
            if (c >= n) continue;
            f32x4 zz = zq[u];
            f32x4 a = zz;
            if (scale) a = zz * *(const f32x4*)(scale + c) + *(const f32x4*)(shift + c);
            if (relu && slope) {      // prelu / leaky ReLU value tensor (act context, xv_common.h)
                const f32x4 sl = *(const f32x4*)(slope + c);
                a.x = a.x > 0.f ? a.x : a.x * sl.x; a.y = a.y > 0.f ? a.y : a.y * sl.y;
                a.z = a.z > 0.f ? a.z : a.z * sl.z; a.w = a.w > 0.f ? a.w : a.w * sl.w;
            } else if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
            f32x4 mean = *(const f32x4*)(o + c), sd = *(const f32x4*)(o + n + c);
            f32x4 dm = *(const f32x4*)(g + c), dsd = *(const f32x4*)(g + n + c);
            f32x4 dv;
            dv.x = sd.x <= sd_eps ? 0.f : dsd.x * 0.5f / sd.x; dv.y = sd.y <= sd_eps ? 0.f : dsd.y * 0.5f / sd.y;
            dv.z = sd.z <= sd_eps ? 0.f : dsd.z * 0.5f / sd.z; dv.w = sd.w <= sd_eps ? 0.f : dsd.w * 0.5f / sd.w;
            f32x4 cen = a - mean;
            f32x4 v = dm * a + dv * cen * cen;
            s += (v.x + v.y) + (v.z + v.w);
        }
    }
    s = wave_sum(s);
    if (lane == 0) dw[row] = s;
}

// Backward through score = scale * act(zk).q :  dzk[r][c] = ds[r]*scale*q[c]*act'(zk),
// per 64-row chunk partials of  dq[c] += ds[r]*scale*act(zk)  and  dbias[c] += dzk[r][c].
// block = 64 column quads x 4 row lanes (the layout of bn_bwd_reduce_kernel).
#define AKB_ROWS 64
__global__ __launch_bounds__(256) void att_key_bwd_kernel(const float* __restrict__ zk, int rows, int n, int act,
                                                          const float* __restrict__ q, float scale, const float* __restrict__ ds,
                                                          float* __restrict__ dzk, float* __restrict__ part /* [chunks][2][n] */) {
    XV_EW_PRIORITY();
    __shared__ f32x4 red[2][4][64];
    const int qx = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + qx) * 4;
    const int r0 = blockIdx.y * AKB_ROWS, r1 = min(rows, r0 + AKB_ROWS);
    f32x4 sq = {0, 0, 0, 0}, sb = {0, 0, 0, 0};
    if (col < n) {
        const f32x4 qq = *(const f32x4*)(q + col) * scale;
        for (int r = r0 + rl; r < r1; r += 4) {
            f32x4 zz = *(const f32x4*)(zk + (long)r * n + col);
            f32x4 k = zz, dk = {1.f, 1.f, 1.f, 1.f};
            if (act == 3) {
                k.x = tanhf(zz.x); k.y = tanhf(zz.y); k.z = tanhf(zz.z); k.w = tanhf(zz.w);
                dk = dk - k * k;
            } else if (act == 1) {
                dk.x = zz.x > 0.f ? 1.f : 0.f; dk.y = zz.y > 0.f ? 1.f : 0.f; dk.z = zz.z > 0.f ? 1.f : 0.f; dk.w = zz.w > 0.f ? 1.f : 0.f;
                k = k * dk;
            }
            const float d = ds[r];
            f32x4 dz = qq * d * dk;
            *(f32x4*)(dzk + (long)r * n + col) = dz;
            sq += k * d;
            sb += dz;
        }
        sq = sq * scale;
    }
    red[0][rl][qx] = sq; red[1][rl][qx] = sb;
    __syncthreads();
    if (rl == 0 && col < n) {
        *(f32x4*)(part + ((long)blockIdx.y * 2 + 0) * n + col) = (red[0][0][qx] + red[0][1][qx]) + (red[0][2][qx] + red[0][3][qx]);
        *(f32x4*)(part + ((long)blockIdx.y * 2 + 1) * n + col) = (red[1][0][qx] + red[1][1][qx]) + (red[1][2][qx] + red[1][3][qx]);
    }
}

__global__ void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, size_t count4) {
    XV_EW_PRIORITY();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 a = ((const f32x4*)y)[i], b = ((const f32x4*)x)[i];
        ((f32x4*)y)[i] = a + b;
    }
}

}  // namespace

extern "C" int xv_att_score(void* stream, const float* zk, int rows, int n, int ldz, int act, const float* query, float scale,
                            float* score) {
    XV_REQUIRE(zk && query && score && rows > 0 && n > 0 && ldz >= n && (act == 0 || act == 1 || act == 3), "att_score: bad arguments (act=%d)", act);
    const bool vec = n % 4 == 0 && ldz % 4 == 0 && ((uintptr_t)zk % 16) == 0 && ((uintptr_t)query % 16) == 0;
    if (vec) hipLaunchKernelGGL(att_score_kernel<true>, dim3(xv_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, zk, rows, n, (long)ldz, act, query, scale, score);
    else hipLaunchKernelGGL(att_score_kernel<false>, dim3(xv_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, zk, rows, n, (long)ldz, act, query, scale, score);
    XV_LAUNCH_CHECK();
    return 0;
}

int xv_softmax_segments_ex(hipStream_t s, const float* score, int b, int t, float* weights, const int32_t* frames, int shrink) {
    XV_REQUIRE(score && weights && b > 0 && t > 0, "softmax_segments: bad arguments");
    hipLaunchKernelGGL(softmax_segments_kernel, dim3(b), dim3(256), 0, s, score, t, weights, (const int*)frames, shrink);
    XV_LAUNCH_CHECK();
    return 0;
}
extern "C" int xv_softmax_segments(void* stream, const float* score, int b, int t, float* weights) {
    return xv_softmax_segments_ex((hipStream_t)stream, score, b, t, weights, nullptr, 0);
}

extern "C" int xv_softmax_segments_backward(void* stream, const float* weights, const float* dweights, int b, int t, float* dscore) {
    XV_REQUIRE(weights && dweights && dscore && b > 0 && t > 0, "softmax_segments_backward: bad arguments");
    hipLaunchKernelGGL(softmax_segments_bwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, weights, dweights, t, dscore);
    XV_LAUNCH_CHECK();
    return 0;
}

extern "C" int xv_att_pool_backward_weights(void* stream, const float* z, int b, int t, int c, const float* scale, const float* shift,
                                            int relu, const float* pool_out, const float* dpool, float* dweights) {
    XV_REQUIRE(z && pool_out && dpool && dweights && b > 0 && t > 0 && c > 0 && c % 4 == 0, "att_pool_backward_weights: bad arguments");
    hipLaunchKernelGGL(att_pool_dw_kernel, dim3(xv_cdiv(b * t, 4)), dim3(256), 0, (hipStream_t)stream, z, b * t, t, c, scale, shift, relu,
                       pool_out, dpool, dweights, relu ? xv_act_context().slope : nullptr);
    XV_LAUNCH_CHECK();
    return 0;
}

extern "C" int xv_att_key_backward(void* stream, const float* zk, int rows, int n, int act, const float* query, float scale,
                                   const float* dscore, float* dzk, float* dquery, float* dbias, void* ws, size_t ws_bytes) {
    XV_REQUIRE(zk && query && dscore && dzk && dquery && rows > 0 && n > 0 && n % 4 == 0 && (act == 0 || act == 1 || act == 3),
               "att_key_backward: bad arguments (n=%d must be a multiple of 4)", n);
    const int chunks = xv_cdiv(rows, AKB_ROWS);
    const size_t part_bytes = (size_t)chunks * 2 * n * sizeof(float);
    XV_REQUIRE(part_bytes + xv_op_workspace_bytes(chunks, n, n) <= ws_bytes, "att_key_backward: workspace too small");
    float* part = (float*)ws;
    void* ws2 = (char*)ws + xv_align(part_bytes, 256);
    const size_t ws2_bytes = ws_bytes - xv_align(part_bytes, 256);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(att_key_bwd_kernel, dim3(xv_cdiv(n / 4, 64), chunks), dim3(256), 0, s, zk, rows, n, act, query, scale, dscore, dzk,
                       part);
    XV_LAUNCH_CHECK();
    int rc = xv_colsum(stream, part, chunks, n, 2 * n, dquery, ws2, ws2_bytes);
    if (rc) return rc;
    if (dbias) rc = xv_colsum(stream, part + n, chunks, n, 2 * n, dbias, ws2, ws2_bytes);
    return rc;
}

// y = act(z) elementwise with the key-layer activation codes (0 identity, 1 relu, 3 tanh): the "<name>_relu" / "<name>_tanh"
// endpoints of common.py's dense_relu / dense_tanh for callers of model.pooling.self_attention (the score kernels apply
// the activation themselves and never need this tensor)
__global__ __launch_bounds__(256) void key_activation_kernel(const float* __restrict__ z, size_t count, int act, float* __restrict__ y) {
    XV_EW_PRIORITY();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) y[i] = key_act(z[i], act);
}

extern "C" int xv_key_activation(void* stream, const float* z, size_t count, int act, float* y) {
    XV_REQUIRE(z && y && (act == 0 || act == 1 || act == 3), "key_activation: act must be 0 (identity), 1 (relu) or 3 (tanh)");
    if (count == 0) return 0;
    int blocks = (int)std::min<size_t>((count + 255) / 256, 8192);
    hipLaunchKernelGGL(key_activation_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, count, act, y);
    XV_LAUNCH_CHECK();
    return 0;
}

extern "C" int xv_add_inplace(void* stream, float* y, const float* x, size_t count) {
    XV_REQUIRE(y && x && count % 4 == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)x % 16) == 0, "add_inplace: count and pointers must be 16-byte multiples");
    if (count == 0) return 0;
    size_t n4 = count / 4;
    int blocks = (int)std::min<size_t>((n4 + 255) / 256, 8192);
    hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, x, n4);
    XV_LAUNCH_CHECK();
    return 0;
}
