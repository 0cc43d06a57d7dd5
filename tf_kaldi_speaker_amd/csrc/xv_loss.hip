// Softmax loss family of model/loss.py:9-355 on gfx950: weight normalisation, the per-chunk
// margin transform + annealing blend + mean sparse cross entropy with its gradients, and the
// gradient through tf.nn.l2_normalize.  The [rows x N] logits GEMMs themselves run on the MFMA
// kernels of xv_gemm.hip; everything here is HBM/L2-bound row or column work.
#include "xv_common.h"

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// block-wide (256 threads) reductions through 4 LDS slots
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum_f(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max_f(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// inv_norm[n] = rsqrt(max(sum_c w[c][n]^2, 1e-12))   (tf.nn.l2_normalize(w, dim=0), loss.py:104)
// block = 256 threads = 32 columns x 8 row lanes (rows c = lane, lane+8, ...), fixed-order combine
__global__ __launch_bounds__(256) void col_inv_norm_kernel(const float* __restrict__ w, int C, int N, int normalize,
                                                           float* __restrict__ inv) {
    XV_EW_FILLER();
    __shared__ float red[8][32];
    const int cx = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + cx;
    float ss = 0.f;
    if (n < N && normalize)
        for (int c = rl; c < C; c += 8) { float v = w[(long)c * N + n]; ss += v * v; }
    red[rl][cx] = ss;
    __syncthreads();
    if (rl != 0 || n >= N) return;
    if (!normalize) { inv[n] = 1.f; return; }
    ss = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) ss += red[k][cx];
    inv[n] = rsqrtf(fmaxf(ss, 1e-12f));
}

// wn[c][ldn] = w[c][n]*inv[n] (pad columns zero), wnt[n][C] = wn^T through a 32x32 LDS tile
__global__ void loss_prep_weight_kernel(const float* __restrict__ w, int C, int N, const float* __restrict__ inv,
                                        float* __restrict__ wn, int ldn, float* __restrict__ wnt) {
    XV_EW_FILLER();
    __shared__ float tile[32][33];
    const int c0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        int c = c0 + r, n = n0 + tx;
        float v = 0.f;
        if (c < C && n < N) v = w[(long)c * N + n] * inv[n];
        if (c < C && n < ldn) wn[(long)c * ldn + n] = v;
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int n = n0 + r, c = c0 + tx;
        if (n < N && c < C) wnt[(long)n * C + c] = tile[tx][r];
    }
}

extern "C" int xv_loss_prep_weight(void* stream, const float* w, int c, int n, int normalize, float* inv_norm, float* wn, int ldn,
                                   float* wnt) {
    XV_REQUIRE(c > 0 && n > 0 && ldn >= n, "loss_prep_weight: bad shape");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(col_inv_norm_kernel, dim3(xv_cdiv(n, 32)), dim3(256), 0, s, w, c, n, normalize, inv_norm);
    XV_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_prep_weight_kernel, dim3(xv_cdiv(ldn, 32), xv_cdiv(c, 32)), dim3(256), 0, s, w, c, n,
                       (const float*)inv_norm, wn, ldn, wnt);
    XV_LAUNCH_CHECK();
    return 0;
}

// One workgroup per chunk (row).  phi / dphi follow loss.py:129-139 (A-Softmax sign-polynomial
// form), :225 (AM), :314-323 (ArcFace with the sqrt(max(1-c^2,1e-12)) guard).
// RQ > 0: the row's logits live in registers (RQ float4 per thread, rows of up to 1 024 RQ entries, ld % 4 == 0): ONE pass over memory with
// every load in flight instead of three dependent passes of 4-byte loads (22 -> ~9 us for 128 rows x 7 352 logits, in the serial chain between
// the forward and the backward pass); RQ = 0: any row length / alignment, three passes.
template <int RQ>
__global__ __launch_bounds__(256) void margin_softmax_rows_kernel(int kind, const float* __restrict__ logits, int rows, int N,
                                                                  long ldl, const float* __restrict__ x, int C,
                                                                  const int* __restrict__ labels, float m, float lambda,
                                                                  float* __restrict__ dlogits, float* __restrict__ dnorm,
                                                                  float* __restrict__ row_loss, float* __restrict__ xnorm,
                                                                  unsigned* __restrict__ ticket, float* __restrict__ loss_out) {
    XV_EW_PRIORITY();
    __shared__ float red[4];
    __shared__ float s_upd, s_dsel, s_dn;
    __shared__ int s_last;
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* lr = logits + (long)r * ldl;
    float* gr = dlogits + (long)r * ldl;
    // a label outside [0, N) is a caller error (TF raises / yields NaN): keep the memory accesses in range and poison the loss
    const int y_raw = labels[r];
    const bool bad_label = y_raw < 0 || y_raw >= N;
    const int y = bad_label ? 0 : y_raw;
    const bool margin = (kind != XV_LOSS_SOFTMAX) && !(kind == XV_LOSS_ASOFTMAX && m == 1.0f);
    float fa = 0.f, fs = 1.f;
    float ss = 0.f;
    if (margin || xnorm) {
        for (int c = tid; c < C; c += 256) { float v = x[(long)r * C + c]; ss += v * v; }
        ss = block_sum(ss, red);
        if (xnorm && tid == 0) xnorm[r] = sqrtf(ss);       // ||x[r]||: what the ||x|| gradient is divided by (xv_add_norm_grad / the segment GEMM's row term)
    }
    if (margin) {
        fa = 1.0f / (1.0f + lambda);
        fs = 1.0f - fa;
        if (tid == 0) {
            const float eps = 1e-12f;
            float sel = lr[y];
            float rawn = sqrtf(ss);
            float fn = fmaxf(rawn, eps);
            float craw = sel / fn;
            float lo = -1.0f + eps, hi = 1.0f - eps;          // == -1, 1 in fp32 (loss.py:125)
            float c = fminf(fmaxf(craw, lo), hi);
            float cm = (craw >= lo && craw <= hi) ? 1.f : 0.f;
            float phi, dphi;
            if (kind == XV_LOSS_ASOFTMAX) {
                float sg = (c > 0.f) - (c < 0.f);
                if (m == 2.0f) {
                    phi = 2.f * sg * c * c - 1.f;
                    dphi = 4.f * fabsf(c);
                } else {
                    float c2 = c * c, c4 = c2 * c2;
                    float t = 2.f * c2 - 1.f;
                    float s3 = ((t > 0.f) - (t < 0.f)) * sg;
                    float s4 = 2.f * sg + s3 - 3.f;
                    phi = s3 * (8.f * c4 - 8.f * c2 + 1.f) + s4;
                    dphi = s3 * (32.f * c2 * c - 16.f * c);
                }
            } else if (kind == XV_LOSS_AMSOFTMAX) {
                phi = c - m;
                dphi = 1.f;
            } else {
                float sin_sq = 1.f - c * c;
                bool live = sin_sq >= 1e-12f;
                float sn = sqrtf(fmaxf(sin_sq, 1e-12f));
                float cosm = cosf(m), sinm = sinf(m);
                float cpm = c * cosm - sn * sinm;
                float dcpm = cosm + (live ? c / sn : 0.f) * sinm;
                bool first = c > cosf(3.14159265358979323846f - m);
                phi = first ? cpm : -cpm - 2.f;
                dphi = first ? dcpm : -dcpm;
            }
            float scaled = phi * fn;
            // updated = fs*logits + fa*(logits + scatter(scaled - sel))      (loss.py:146-152)
            s_upd = fs * sel + fa * (sel + (scaled - sel));
            s_dsel = fa * (dphi * cm - 1.f);
            // d/d||x||: fa * (phi - dphi*cm*craw), routed to x only where ||x|| >= eps
            s_dn = (rawn >= eps) ? fa * (phi - dphi * cm * craw) : 0.f;
        }
        __syncthreads();
    }
    // log-sum-exp over the updated logits
    const float inv_rows = 1.0f / (float)rows;
    float lse;
    if (RQ > 0) {
        const int nq = (int)(ldl >> 2);
        f32x4 uq[RQ > 0 ? RQ : 1];
#pragma unroll
        for (int u = 0; u < RQ; ++u) uq[u] = *(const f32x4*)(lr + 4 * min(tid + 256 * u, nq - 1));
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < RQ; ++u) {
            const int j0 = 4 * (tid + 256 * u);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + e;
                const float l = uq[u][e];
                const float v = margin ? (j == y ? s_upd : fs * l + fa * l) : l;
                uq[u][e] = (tid + 256 * u < nq && j < N) ? v : -INFINITY;      // (entries beyond the row: exp(-inf) = 0 below, gradient 0)
                mx = fmaxf(mx, uq[u][e]);
            }
        }
        mx = block_max(mx, red);
        float se = 0.f;
#pragma unroll
        for (int u = 0; u < RQ; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) se += expf(uq[u][e] - mx);
        se = block_sum(se, red);
        lse = logf(se) + mx;
#pragma unroll
        for (int u = 0; u < RQ; ++u) {
            if (tid + 256 * u >= nq) continue;
            const int j0 = 4 * (tid + 256 * u);
            f32x4 g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + e;
                const float p = expf(uq[u][e] - lse);                          // 0 beyond N
                const float gj = (p - (j == y ? 1.f : 0.f)) * inv_rows;
                float gg = margin ? (fs + fa) * gj : gj;
                if (margin && j == y) gg += s_dsel * gj;
                g[e] = j < N ? gg : 0.f;
            }
            *(f32x4*)(gr + j0) = g;
        }
    } else {
        float mx = -INFINITY;
        for (int j = tid; j < N; j += 256) {
            float l = lr[j];
            float u = margin ? (j == y ? s_upd : fs * l + fa * l) : l;
            mx = fmaxf(mx, u);
        }
        mx = block_max(mx, red);
        float se = 0.f;
        for (int j = tid; j < N; j += 256) {
            float l = lr[j];
            float u = margin ? (j == y ? s_upd : fs * l + fa * l) : l;
            se += expf(u - mx);
        }
        se = block_sum(se, red);
        lse = logf(se) + mx;
        for (int j = tid; j < (int)ldl; j += 256) {
            float g = 0.f;
            if (j < N) {
                float l = lr[j];
                float u = margin ? (j == y ? s_upd : fs * l + fa * l) : l;
                float p = expf(u - lse);
                float gj = (p - (j == y ? 1.f : 0.f)) * inv_rows;
                g = margin ? (fs + fa) * gj : gj;
                if (margin && j == y) g += s_dsel * gj;
            }
            gr[j] = g;
        }
    }
    const float uy = margin ? s_upd : lr[y];
    if (tid == 0) {
        const float rl = bad_label ? NAN : lse - uy;
        row_loss[r] = rl;
        float py = expf(uy - lse);
        dnorm[r] = margin ? s_dn * (py - 1.f) * inv_rows : 0.f;
        if (ticket) {
            // the mean over the rows in the same launch: the last workgroup to finish sums row_loss in index order (mean_kernel's
            // arithmetic).  Hand-over by agent-scope atomics (the xv_handoff_* contract of xv_common.h), not __threadfence() (a
            // whole-L2 write-back per workgroup on gfx950)
            __hip_atomic_store(row_loss + r, rl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (t == (unsigned)(rows - 1));
            if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!ticket) return;
    __syncthreads();
    if (!s_last) return;
    float sum = 0.f;
    for (int i = tid; i < rows; i += 256) sum += __hip_atomic_load(row_loss + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sum = block_sum(sum, red);
    if (tid == 0) *loss_out = sum / (float)rows;
}

__global__ void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    XV_EW_PRIORITY();
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = s / (float)n;
}

static void launch_margin_softmax_rows(hipStream_t s, int kind, const float* logits, int rows, int n, int ldl, const float* x, int c,
                                       const int32_t* labels, float m, float lambda, float* dlogits, float* dnorm, float* row_loss, float* xnorm,
                                       uint32_t* ticket, float* loss_out) {
    const bool vec = ldl % 4 == 0 && ((uintptr_t)logits % 16) == 0 && ((uintptr_t)dlogits % 16) == 0;
    const int nq = ldl / 4;
#define XV_MSR_LAUNCH(RQ) hipLaunchKernelGGL(margin_softmax_rows_kernel<RQ>, dim3(rows), dim3(256), 0, s, kind, logits, rows, n, (long)ldl, x, c, \
                                             (const int*)labels, m, lambda, dlogits, dnorm, row_loss, xnorm, (unsigned*)ticket, loss_out)
    if (vec && nq <= 256 * 8) XV_MSR_LAUNCH(8);
    else if (vec && nq <= 256 * 16) XV_MSR_LAUNCH(16);
    else XV_MSR_LAUNCH(0);
#undef XV_MSR_LAUNCH
}

extern "C" int xv_margin_softmax_rows(void* stream, int kind, const float* logits, int rows, int n, int ldl, const float* x, int c,
                                      const int32_t* labels, float m, float lambda, float* dlogits, float* dnorm, float* row_loss,
                                      float* loss_out) {
    XV_REQUIRE(rows > 0 && n > 0 && ldl >= n && c > 0, "margin_softmax_rows: bad shape");
    XV_REQUIRE(kind >= XV_LOSS_SOFTMAX && kind <= XV_LOSS_ARCSOFTMAX, "Not implement loss kind %d", kind);
    if (kind == XV_LOSS_ASOFTMAX)
        XV_REQUIRE(m == 1.0f || m == 2.0f || m == 4.0f, "[ERROR] m=%d is not unsupported.", (int)m);
    hipStream_t s = (hipStream_t)stream;
    launch_margin_softmax_rows(s, kind, logits, rows, n, ldl, x, c, labels, m, lambda, dlogits, dnorm, row_loss, nullptr, nullptr, nullptr);
    XV_LAUNCH_CHECK();
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, (const float*)row_loss, rows, loss_out);
    XV_LAUNCH_CHECK();
    return 0;
}

// Engine form: also writes ||x[r]|| and folds the mean into the same launch (ticket: one zeroed uint32, left zeroed)
int xv_margin_softmax_rows_ex(hipStream_t s, int kind, const float* logits, int rows, int n, int ldl, const float* x, int c,
                              const int32_t* labels, float m, float lambda, float* dlogits, float* dnorm, float* row_loss,
                              float* loss_out, float* xnorm, uint32_t* ticket) {
    XV_REQUIRE(rows > 0 && n > 0 && ldl >= n && c > 0 && ticket, "margin_softmax_rows: bad shape");
    XV_REQUIRE(kind >= XV_LOSS_SOFTMAX && kind <= XV_LOSS_ARCSOFTMAX, "Not implement loss kind %d", kind);
    if (kind == XV_LOSS_ASOFTMAX)
        XV_REQUIRE(m == 1.0f || m == 2.0f || m == 4.0f, "[ERROR] m=%d is not unsupported.", (int)m);
    launch_margin_softmax_rows(s, kind, logits, rows, n, ldl, x, c, labels, m, lambda, dlogits, dnorm, row_loss, xnorm, ticket, loss_out);
    XV_LAUNCH_CHECK();
    return 0;
}

// dx[r][:] += dnorm[r] * x[r][:] / ||x[r]||
__global__ void add_norm_grad_kernel(const float* __restrict__ x, const float* __restrict__ dnorm, int rows, int C,
                                     float* __restrict__ dx) {
    XV_EW_PRIORITY();
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long)row * C;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) ss += xr[c] * xr[c];
    ss = wave_sum_f(ss);
    float nrm = sqrtf(ss);
    float k = nrm > 0.f ? dnorm[row] / nrm : 0.f;
    for (int c = lane; c < C; c += 64) dx[(long)row * C + c] += k * xr[c];
}
extern "C" int xv_add_norm_grad(void* stream, const float* x, const float* dnorm, int rows, int c, float* dx) {
    XV_REQUIRE(rows > 0 && c > 0, "add_norm_grad: bad shape");
    hipLaunchKernelGGL(add_norm_grad_kernel, dim3(xv_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, dnorm, rows, c, dx);
    XV_LAUNCH_CHECK();
    return 0;
}

// dot[n] = sum_c dwn[c][n] * wn[c][n]      (32 columns x 8 row lanes per block)
__global__ __launch_bounds__(256) void col_dot_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                      int C, int N, float* __restrict__ dot) {
    XV_EW_FILLER();
    __shared__ float red[8][32];
    const int cx = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + cx;
    float s = 0.f;
    if (n < N)
        for (int c = rl; c < C; c += 8) s += a[(long)c * lda + n] * b[(long)c * ldb + n];
    red[rl][cx] = s;
    __syncthreads();
    if (rl != 0 || n >= N) return;
    s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][cx];
    dot[n] = s;
}
__global__ void loss_weight_bwd_kernel(const float* __restrict__ dwn, long lddwn, const float* __restrict__ wn, long ldn,
                                       const float* __restrict__ inv, const float* __restrict__ dot, const float* __restrict__ w,
                                       int C, int N, int normalize, float l2, float* __restrict__ dw) {
    XV_EW_FILLER();
    long total = (long)C * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i / N), n = (int)(i - (long)c * N);
        float g = dwn[(long)c * lddwn + n];
        if (normalize) {
            // ss < eps columns are frozen by the maximum() of l2_normalize: only the scale term survives
            float ss_live = inv[n] < 0.99e6f ? 1.f : 0.f;   // inv == rsqrt(1e-12) ~ 1e6 when clamped
            g = inv[n] * (g - ss_live * wn[(long)c * ldn + n] * dot[n]);
        }
        dw[i] = g + l2 * w[i];
    }
}
extern "C" int xv_loss_weight_backward(void* stream, const float* dwn, int lddwn, const float* wn, int ldn, const float* inv_norm,
                                       const float* w, int c, int n, int normalize, float l2_scale, float* dw, void* ws,
                                       size_t ws_bytes) {
    XV_REQUIRE(c > 0 && n > 0 && lddwn >= n && ldn >= n, "loss_weight_backward: bad shape");
    XV_REQUIRE(!normalize || (size_t)n * sizeof(float) <= ws_bytes, "loss_weight_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* dot_buf = (float*)ws;
    if (normalize) {
        hipLaunchKernelGGL(col_dot_kernel, dim3(xv_cdiv(n, 32)), dim3(256), 0, s, dwn, (long)lddwn, wn, (long)ldn, c, n, dot_buf);
        XV_LAUNCH_CHECK();
    }
    long total = (long)c * n;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(loss_weight_bwd_kernel, dim3(blocks), dim3(256), 0, s, dwn, (long)lddwn, wn, (long)ldn, inv_norm,
                       (const float*)dot_buf, w, c, n, normalize, l2_scale, dw);
    XV_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------
// auxiliary losses (model/loss.py:985-1036; shipped in *_r0.01.json and *_mhe0.01.json)
// ------------------------------------------------------------------------------------
// ring loss: lambda * mean((||x|| - r)^2), r a trainable scalar.  One workgroup, rows walked in a fixed order:
// adds the loss to *loss_accum, lambda*2*(||x|| - r)/rows to dnorm[row] (the gradient w.r.t. ||x|| that
// xv_add_norm_grad turns into d x) and writes d r.
__global__ __launch_bounds__(256) void ring_loss_kernel(const float* __restrict__ x, int rows, int n, long ldx, const float* __restrict__ r,
                                                        float lambda, float* __restrict__ loss_accum, float* __restrict__ dnorm,
                                                        float* __restrict__ dr) {
    XV_EW_PRIORITY();
    __shared__ float s_sq[4], s_d[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float rr = *r;
    float sq = 0.f, sd = 0.f;
    for (int row = wave; row < rows; row += 4) {
        const float* xr = x + (long)row * ldx;
        float ss = 0.f;
        for (int c = lane; c < n; c += 64) ss += xr[c] * xr[c];
        ss = wave_sum_f(ss);
        const float diff = sqrtf(ss) - rr;
        sq += diff * diff;
        sd += diff;
        if (lane == 0) dnorm[row] += lambda * 2.0f * diff / (float)rows;
    }
    if (lane == 0) { s_sq[wave] = sq; s_d[wave] = sd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tq = (s_sq[0] + s_sq[1]) + (s_sq[2] + s_sq[3]), td = (s_d[0] + s_d[1]) + (s_d[2] + s_d[3]);
        *loss_accum += lambda * tq / (float)rows;
        *dr = -lambda * 2.0f * td / (float)rows;
    }
}

extern "C" int xv_ring_loss(void* stream, const float* x, int rows, int n, int ldx, const float* r, float lambda, float* loss_accum,
                            float* dnorm_accum, float* dr) {
    XV_REQUIRE(x && r && loss_accum && dnorm_accum && dr && rows > 0 && n > 0 && ldx >= n, "ring_loss: bad arguments");
    hipLaunchKernelGGL(ring_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, rows, n, (long)ldx, r, lambda, loss_accum,
                       dnorm_accum, dr);
    XV_LAUNCH_CHECK();
    return 0;
}

// MHE on the column-normalised speaker weights wn [c][ldn]:  M = mean_{b,n}(2 - 2 wn[:,y_b].wn[:,n]) = 2 - 2 u.v/(B N),
// u = sum_b wn[:,y_b], v = sum_n wn[:,n];  loss = lambda / (M + 1e-6);  d loss / d wn[:,n] = g (u + cnt_n v),
// g = 2 lambda / ((M + 1e-6)^2 B N), cnt_n = #{b : y_b = n}.   coef = [g | u[c] | v[c]], counts = int32 [n].
__global__ void mhe_counts_kernel(const int* __restrict__ labels, int rows, int n, int* __restrict__ counts) {
    XV_EW_FILLER();
    for (int i = threadIdx.x; i < n; i += blockDim.x) counts[i] = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < rows; b += blockDim.x) {
        const int y = labels[b];
        if (y >= 0 && y < n) atomicAdd(counts + y, 1);      // out-of-range labels already poison the main loss with NaN
    }
}
__global__ __launch_bounds__(256) void mhe_uv_kernel(const float* __restrict__ wn, int n, long ldn, const int* __restrict__ labels, int rows,
                                                     float* __restrict__ coef, int c_total) {
    XV_EW_PRIORITY();
    __shared__ float red[4];
    const int c = blockIdx.x;
    const float* w = wn + (long)c * ldn;
    float v = 0.f, u = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) v += w[i];
    for (int b = threadIdx.x; b < rows; b += 256) {
        const int y = labels[b];
        if (y >= 0 && y < n) u += w[y];
    }
    v = block_sum(v, red);
    u = block_sum(u, red);
    if (threadIdx.x == 0) { coef[1 + c] = u; coef[1 + c_total + c] = v; }
}
__global__ __launch_bounds__(256) void mhe_finalize_kernel(float* __restrict__ coef, int c_total, int rows, int n, float lambda,
                                                           float* __restrict__ loss_accum) {
    XV_EW_PRIORITY();
    __shared__ float red[4];
    float d = 0.f;
    for (int c = threadIdx.x; c < c_total; c += 256) d += coef[1 + c] * coef[1 + c_total + c];
    d = block_sum(d, red);
    if (threadIdx.x == 0) {
        const float bn = (float)rows * (float)n;
        const float M = 2.0f - 2.0f * d / bn + 1e-6f;
        *loss_accum += lambda / M;
        coef[0] = 2.0f * lambda / (M * M * bn);
    }
}
__global__ void mhe_add_grad_kernel(float* __restrict__ dwn, int c_total, int n, long ldn, const float* __restrict__ coef,
                                    const int* __restrict__ counts) {
    XV_EW_FILLER();
    const float g = coef[0];
    const long total = (long)c_total * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i / n), j = (int)(i - (long)c * n);
        dwn[(long)c * ldn + j] += g * (coef[1 + c] + (float)counts[j] * coef[1 + c_total + c]);
    }
}

extern "C" int xv_mhe_loss(void* stream, const float* wn, int c, int n, int ldn, const int32_t* labels, int rows, float lambda,
                           float* loss_accum, float* coef, int32_t* counts) {
    XV_REQUIRE(wn && labels && loss_accum && coef && counts && c > 0 && n > 0 && ldn >= n && rows > 0, "mhe_loss: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mhe_counts_kernel, dim3(1), dim3(1024), 0, s, (const int*)labels, rows, n, (int*)counts);
    XV_LAUNCH_CHECK();
    hipLaunchKernelGGL(mhe_uv_kernel, dim3(c), dim3(256), 0, s, wn, n, (long)ldn, (const int*)labels, rows, coef, c);
    XV_LAUNCH_CHECK();
    hipLaunchKernelGGL(mhe_finalize_kernel, dim3(1), dim3(256), 0, s, coef, c, rows, n, lambda, loss_accum);
    XV_LAUNCH_CHECK();
    return 0;
}

extern "C" int xv_mhe_add_grad(void* stream, float* dwn, int c, int n, int ldn, const float* coef, const int32_t* counts) {
    XV_REQUIRE(dwn && coef && counts && c > 0 && n > 0 && ldn >= n, "mhe_add_grad: bad arguments");
    long total = (long)c * n;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mhe_add_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dwn, c, n, (long)ldn, coef, (const int*)counts);
    XV_LAUNCH_CHECK();
    return 0;
}
