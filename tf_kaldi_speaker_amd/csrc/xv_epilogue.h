// Shared GEMM epilogue: per-128-row-tile column statistics of the output tile held in the 2 x NB v_mfma 32x32
// accumulators per wave of a workgroup of 2 x (4/NB) waves (NB = 2: four waves of 64x64, NB = 1: eight of 64x32).
//   part layout: [4][tiles_m][N] = column sum | sum of squares centred on the TILE mean (Chan-combinable,
//   no E[x^2]-E[x]^2 cancellation) | column min | column max  (valid rows only).
// sum/M2 feed BatchNorm (tdnn.py:46); min/max give the exact range of the BN output, which the
// split-precision path needs to pick the power-of-two scale of the next layer's fp16 operand planes.
// Deterministic: fixed-order combines through LDS, no atomics.
#pragma once
#include "xv_common.h"

template <int NB>
__device__ __forceinline__ void xv_tile_stats_epilogue(const f32x16 (&acc)[2][NB], float* red /* >= 1024 floats of LDS, free */,
                                                       int tid, int wr, int wc, int li, int lh, int m0, int n0, int M, int N,
                                                       int tile_m, int tiles_m, float* __restrict__ part) {
    // (opaque copies: called inside the tile loop of the evenly scheduled kernel, where hipcc otherwise hoists the lane constants derived from
    // these out of the loop and keeps them live - spilled - across the K loop)
    asm volatile("" : "+v"(tid), "+v"(wr), "+v"(wc), "+v"(li), "+v"(lh));
    float* r_sum = red;          // [2][128]
    float* r_m2 = red + 256;
    float* r_min = red + 512;
    float* r_max = red + 768;
    const int cnt = min(XV_TILE_M, M - m0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float v = 0.f, mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                bool ok = m < M;
                float x = acc[a][b][r];
                v += ok ? x : 0.f;
                mn = ok ? fminf(mn, x) : mn;
                mx = ok ? fmaxf(mx, x) : mx;
            }
        v += __shfl_xor(v, 32);
        mn = fminf(mn, __shfl_xor(mn, 32));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (lh == 0) {
            int col = (wc * NB + b) * 32 + li;
            r_sum[wr * 128 + col] = v;
            r_min[wr * 128 + col] = mn;
            r_max[wr * 128 + col] = mx;
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        int col = (wc * NB + b) * 32 + li;
        float mean = (r_sum[col] + r_sum[128 + col]) / (float)cnt;
        float v = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float d = acc[a][b][r] - mean;
                v += (m < M) ? d * d : 0.f;
            }
        v += __shfl_xor(v, 32);
        if (lh == 0) r_m2[wr * 128 + col] = v;
    }
    __syncthreads();
    if (tid < 128 && m0 < M) {       // (a 256-row workgroup's lower half may lie entirely below the last row)
        int n = n0 + tid;
        if (n < N) {
            const long plane = (long)tiles_m * N;
            const long o = (long)tile_m * N + n;
            part[o] = r_sum[tid] + r_sum[128 + tid];
            part[plane + o] = r_m2[tid] + r_m2[128 + tid];
            part[2 * plane + o] = fminf(r_min[tid], r_min[128 + tid]);
            part[3 * plane + o] = fmaxf(r_max[tid], r_max[128 + tid]);
        }
    }
}

// Data-gradient epilogue: the tile in `acc` is d a (the gradient w.r.t. a BN+ReLU output); together with the matching
// tile of that layer's pre-BN tensor z it yields the per-tile partials of the BN backward reductions
//   part[tile_m][0][n] = sum_rows dd,   [1] = sum_rows dd * xhat,   [2] = max_rows |dd|,     dd = (z*scale+shift > 0) ? d a : 0,
// i.e. what bn_bwd_reduce_kernel computes from memory - without re-reading d a (xv_elementwise.hip consumes the partials).
struct XvBwdStats { const float* z; const float* scale; const float* shift; const float* mean; const float* invstd; float* part; };

template <int NB>
__device__ __forceinline__ void xv_tile_bwd_stats_epilogue(const f32x16 (&acc)[2][NB], float* red /* >= 768 floats of LDS, free */, int tid,
                                                           int wr, int wc, int li, int lh, int m0, int n0, int M, int N, int tile_m,
                                                           const XvBwdStats& q) {
    float* r1 = red;             // [2][128]
    float* r2 = red + 256;
    float* r3 = red + 512;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int col = (wc * NB + b) * 32 + li, n = n0 + col;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (n < N) {
            const float sc = q.scale[n], sh = q.shift[n], mu = q.mean[n], is = q.invstd[n];
            // unconditional loads (row index clamped) issued as one batch: a predicated load compiles to a branch + vmcnt(0),
            // i.e. 32 serialised memory round trips per column block
            float zz[2][16];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    zz[a][r] = q.z[(long)min(m, M - 1) * N + n];
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float dd = (m < M && zz[a][r] * sc + sh > 0.f) ? acc[a][b][r] : 0.f;
                    s1 += dd;
                    s2 += dd * ((zz[a][r] - mu) * is);
                    s3 = fmaxf(s3, fabsf(dd));
                }
        }
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); s3 = fmaxf(s3, __shfl_xor(s3, 32));
        if (lh == 0) { r1[wr * 128 + col] = s1; r2[wr * 128 + col] = s2; r3[wr * 128 + col] = s3; }
    }
    __syncthreads();
    if (tid < 128 && m0 < M) {
        const int n = n0 + tid;
        if (n < N) {
            float* o = q.part + (long)tile_m * 3 * N + n;
            o[0] = r1[tid] + r1[128 + tid];
            o[N] = r2[tid] + r2[128 + tid];
            o[2 * N] = fmaxf(r3[tid], r3[128 + tid]);
        }
    }
}

// The same statistics for a tile held as 4 x 4 v_mfma_f32_16x16x32 accumulators per wave (4 waves, 2 x 2, 64 x 64 each):
// lane l owns column l & 15 of block b and rows (l >> 4) * 4 + j of block a.
__device__ __forceinline__ void xv_tile_stats_epilogue16(const f32x4 (&acc)[4][4], float* red /* >= 1024 floats of LDS, free */, int tid, int wr,
                                                         int wc, int lane, int m0, int n0, int M, int N, int tile_m, int tiles_m,
                                                         float* __restrict__ part) {
    float* r_sum = red;          // [2][128]
    float* r_m2 = red + 256;
    float* r_min = red + 512;
    float* r_max = red + 768;
    const int lc = lane & 15, lg = lane >> 4;
    const int cnt = min(XV_TILE_M, M - m0);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float v = 0.f, mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = m0 + wr * 64 + a * 16 + lg * 4 + j < M;
                const float x = acc[a][b][j];
                v += ok ? x : 0.f;
                mn = ok ? fminf(mn, x) : mn;
                mx = ok ? fmaxf(mx, x) : mx;
            }
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        mn = fminf(mn, __shfl_xor(mn, 16)); mn = fminf(mn, __shfl_xor(mn, 32));
        mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (lg == 0) {
            const int col = wc * 64 + b * 16 + lc;
            r_sum[wr * 128 + col] = v; r_min[wr * 128 + col] = mn; r_max[wr * 128 + col] = mx;
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int col = wc * 64 + b * 16 + lc;
        const float mean = (r_sum[col] + r_sum[128 + col]) / (float)cnt;
        float v = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = acc[a][b][j] - mean;
                v += (m0 + wr * 64 + a * 16 + lg * 4 + j < M) ? d * d : 0.f;
            }
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if (lg == 0) r_m2[wr * 128 + col] = v;
    }
    __syncthreads();
    if (tid < 128 && m0 < M) {
        int n = n0 + tid;
        if (n < N) {
            const long plane = (long)tiles_m * N;
            const long o = (long)tile_m * N + n;
            part[o] = r_sum[tid] + r_sum[128 + tid];
            part[plane + o] = r_m2[tid] + r_m2[128 + tid];
            part[2 * plane + o] = fminf(r_min[tid], r_min[128 + tid]);
            part[3 * plane + o] = fmaxf(r_max[tid], r_max[128 + tid]);
        }
    }
}

// power-of-two scale that brings a tensor with max |x| = amax (given as the uint bits of a non-negative
// float) to [2^12, 2^13): exact to apply and to undo, 3 bits of headroom below the fp16 maximum.
__device__ __forceinline__ float xv_pow2_scale(unsigned amax_bits) {
    float amax = __uint_as_float(amax_bits);
    if (!(amax > 0.f) || !(amax < INFINITY)) return 1.0f;
    int e = (int)((amax_bits >> 23) & 0xff) - 127;       // floor(log2(amax)) for normal numbers
    int s = 12 - e;
    s = max(-100, min(100, s));
    return __uint_as_float((unsigned)(s + 127) << 23);
}
