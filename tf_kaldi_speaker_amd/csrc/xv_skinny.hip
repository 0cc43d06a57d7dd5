// Segment-level GEMMs: C[M][N] = A[M][K] . Bt[N][K]^T with M <= 128 rows (the chunks of one batch), the split-K sum and the
// consumer's per-column work in ONE launch.
//
// Why: between the frame-level forward and backward sits a strictly serial chain of small problems (tdnn6, tdnn7, the logits,
// their gradients; model/tdnn.py:147-189, model/loss.py) - 128 x {512, 7351, 3000} outputs over K = 512 ... 7352.  As "GEMM
// (split-K slabs) -> slab sum -> BatchNorm" each layer was three launches of 6-25 us, all latency, with the MFMA pipes of the
// whole chip idle: 22 launches, ~0.25 ms per step.  Fusing the slab sum into the CONSUMERS made them slower (their
// one-wave-per-column shape reads slabs uncoalesced); this file fuses on the PRODUCER side:
//   * a workgroup owns all 128 rows x 32 columns of the output over one K-chunk; 4 waves = 4 x 32 rows, one
//     v_mfma_f32_32x32x2_f32 accumulator each; operands are staged 32 k at a time through registers into a swizzled LDS image with
//     three stages of loads in flight - the problem is latency-bound, so the loop is built for load parallelism (a first version
//     that loaded the MFMA operands straight from global memory, one row per lane, spent 2 us per 64 k in the texture addresser:
//     32 cache lines per instruction);
//   * K is split over gridDim.y workgroups that write [128][32] slabs; a ticket per column tile elects the LAST workgroup to
//     arrive, which sums the slabs in split order (fixed order -> bit-reproducible, unlike atomics) - 16 KB per split, so the
//     tail is a few microseconds on N/32 CUs in parallel;
//   * that workgroup holds complete columns (all rows of the batch), so the training-mode BatchNorm of the layer (statistics,
//     moving averages, scale/shift, activation - bn_small_fwd_kernel's arithmetic) or the BatchNorm backward of the layer
//     BELOW the gradient GEMM (bn_small_bwd_kernel's) runs on its registers before anything is written.
#include "xv_common.h"

#include <algorithm>
#include <stdlib.h>

namespace {

constexpr int SK_COLS = 32;      // columns per workgroup
constexpr int SK_ROWS = 128;     // rows per workgroup = the whole batch
constexpr int SK_KS = 32;        // k per stage: one 128-byte line per operand row
constexpr int SK_STAGE = (SK_ROWS + SK_COLS) * SK_KS;      // floats per LDS stage: A [128][32] then B [32][32] (20 KB), double-buffered
#define SK_SWZ(row) (((row) >> 1) & 7)                    // 16-byte chunk c of row r sits at chunk c ^ SK_SWZ(r) (xv_gemm.hip, BK = 32)

struct SkArgs {
    XvSkinny g;
    int tiles_n, splits, k_chunk;
    float* slab;
    const float* zero;
};

// sum of v over the (valid) rows of column li: 16 values per lane -> lane halves -> the 4 waves, fixed order
__device__ __forceinline__ float sk_col_sum(float v, float (*red)[SK_COLS], int wave, int li, int lh) {
    v += __shfl_xor(v, 32);
    __syncthreads();
    if (lh == 0) red[wave][li] = v;
    __syncthreads();
    return (red[0][li] + red[1][li]) + (red[2][li] + red[3][li]);
}

template <int EPI>
__global__ __launch_bounds__(256) void xv_skinny_kernel(SkArgs p) {
    XV_EW_PRIORITY();
    __shared__ __attribute__((aligned(16))) float smem[2 * SK_STAGE];
    __shared__ float red[4][SK_COLS];
    __shared__ int s_last;
    const XvSkinny& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int tile = blockIdx.x, z = blockIdx.y;
    const int n0 = tile * SK_COLS;
    const int k_begin = z * p.k_chunk;
    const int k_end = min(g.K, k_begin + p.k_chunk);
    const int nk = (k_end - k_begin + SK_KS - 1) / SK_KS;

    // ---- global -> VGPR -> LDS, three stages of loads in flight.  (LDS-DMA as in xv_gemm.hip does not pipeline here: hipcc drains
    // vmcnt(0) in front of every ds_read that follows a global_load_lds, which the big GEMMs hide behind 4 co-resident workgroups and
    // this kernel, at 1-2 workgroups per CU, cannot.  With register staging the compiler waits for exactly the stage it writes.)
    // Thread (lrow = tid / 8, lpos = tid % 8) owns the 16-byte chunk lpos of rows lrow + 32 i of A (i < 4) and of row lrow of B: a
    // row's 128 bytes are one coalesced request.  Out-of-range rows / chunks beyond k_end read the zero page.
    const int lrow = tid >> 3, lpos = tid & 7;
    const float* __restrict__ zp = p.zero;
    const float* srcp[5];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = lrow + 32 * i;
        srcp[i] = (row < g.M ? g.A + (long)row * g.lda : zp) + 4 * lpos;
    }
    srcp[4] = (n0 + lrow < g.N ? g.Bt + (long)(n0 + lrow) * g.ldb : zp) + 4 * lpos;
    auto gload = [&](f32x4 (&r)[5], int kt) {
        const int k0 = k_begin + kt * SK_KS;
        const bool kv = k0 + 4 * lpos < k_end;       // K % 4 == 0: a chunk is inside or outside as a whole
#pragma unroll
        for (int i = 0; i < 5; ++i) r[i] = *(const f32x4*)(kv ? srcp[i] + k0 : zp);
    };
    // LDS image per stage: A [128][32] then B [32][32] floats, chunk c of row r at position c ^ SK_SWZ(r)
    int dst[5];
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = (lrow + 32 * i) * SK_KS + ((lpos ^ SK_SWZ(lrow + 32 * i)) << 2);
    dst[4] = SK_ROWS * SK_KS + lrow * SK_KS + ((lpos ^ SK_SWZ(lrow)) << 2);
    auto lstore = [&](const f32x4 (&r)[5], int buf) {
#pragma unroll
        for (int i = 0; i < 5; ++i) *(f32x4*)(smem + buf * SK_STAGE + dst[i]) = r[i];
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int fsw = SK_SWZ(li);      // rows 32 wave + li and li share the swizzle (32 wave is a multiple of 16)
    const int a_off = (32 * wave + li) * SK_KS, b_off = SK_ROWS * SK_KS + li * SK_KS;
    auto compute = [&](int buf) {
        const float* sbuf = smem + buf * SK_STAGE;
#pragma unroll
        for (int q = 0; q < SK_KS / 8; ++q) {
            const int pos = ((2 * q + lh) ^ fsw) << 2;
            const f32x4 af = *(const f32x4*)(sbuf + a_off + pos);
            const f32x4 bf = *(const f32x4*)(sbuf + b_off + pos);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
        }
    };
    // stage kt: its registers -> LDS buffer kt & 1 (last read in iteration kt - 2, behind a barrier), the same registers reloaded
    // with stage kt + 3 (stages beyond the chunk read the zero page: the loads stay unconditional, so the compiler's vmcnt for the
    // ds_write is exactly "the two younger stages may still be in flight"), barrier, MFMAs.  Three register sets, named - rotating one
    // set by copies would make every copy wait for the loads it moves.
#define SK_STEP(R, kt_)                 \
    do {                                \
        lstore(R, (kt_) & 1);           \
        gload(R, (kt_) + 3);            \
        __syncthreads();                \
        compute((kt_) & 1);             \
    } while (0)
    f32x4 r0[5], r1[5], r2[5];
    gload(r0, 0);
    gload(r1, 1);
    gload(r2, 2);
    int kt = 0;
    for (; kt + 3 <= nk; kt += 3) {
        SK_STEP(r0, kt);
        SK_STEP(r1, kt + 1);
        SK_STEP(r2, kt + 2);
    }
    if (kt < nk) SK_STEP(r0, kt);
    if (kt + 1 < nk) SK_STEP(r1, kt + 1);
#undef SK_STEP

    // accumulator r of lane (li, lh) is C[row(r)][n0 + li], row(r) = 32 wave + (r & 3) + 8 (r >> 2) + 4 lh
    const int row0 = wave * 32 + 4 * lh;
    if (p.splits > 1) {
        // Slab hand-over without a device-wide fence (the xv_handoff_* contract of xv_common.h, gfx950 only).  __threadfence() here is "buffer_wbl2 sc1": every wave writes back its XCD's
        // whole L2 - measured 12 us per split on this kernel.  Instead the slab values are stored and loaded as relaxed agent-scope
        // atomics (sc1: written through to / read from the level all XCDs share), the stores are drained (vmcnt(0)) in front of the
        // workgroup barrier, and only then does thread 0 take the ticket.
        float* mine = p.slab + ((long)z * p.tiles_n + tile) * (SK_ROWS * SK_COLS) + li;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __hip_atomic_store(mine + (row0 + (r & 3) + 8 * (r >> 2)) * SK_COLS, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(&g.tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (t == (unsigned)(p.splits - 1));
            if (s_last) __hip_atomic_store(&g.tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every other split has taken its ticket: ready for the next launch
        }
        __syncthreads();
        if (!s_last) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // four splits' loads in flight at a time, added in split order
        const float* all = p.slab + (long)tile * (SK_ROWS * SK_COLS) + li;
        const long zstride = (long)p.tiles_n * (SK_ROWS * SK_COLS);
        int zz = 0;
        for (; zz + 4 <= p.splits; zz += 4) {
            float t[4][16];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    t[u][r] = __hip_atomic_load(all + (zz + u) * zstride + (row0 + (r & 3) + 8 * (r >> 2)) * SK_COLS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += t[u][r];
        }
        for (; zz < p.splits; ++zz) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[r] += __hip_atomic_load(all + zz * zstride + (row0 + (r & 3) + 8 * (r >> 2)) * SK_COLS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    const int n = n0 + li;
    const bool nv = n < g.N;
    const int nc = nv ? n : 0;
    // optional rank-one row term: acc[m][n] += (norm[m] > 0 ? coef[m] / norm[m] : 0) * X[m][n]   (loss.py: gradient through ||x||)
    if (g.row_coef) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = row0 + (r & 3) + 8 * (r >> 2);
            if (m < g.M) {
                const float nr = g.row_norm[m];
                const float kf = nr > 0.f ? g.row_coef[m] / nr : 0.f;
                acc[r] += kf * g.X[(long)m * g.ldx + nc];
            }
        }
    }
    if (EPI == XV_SK_PLAIN) {
        const float bias = g.bias ? g.bias[nc] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = row0 + (r & 3) + 8 * (r >> 2);
            if (m < g.M && nv) g.C[(long)m * g.ldc + n] = acc[r] + bias;
        }
        return;
    }
    const float rows_f = (float)g.M;
    if (EPI == XV_SK_BN_FWD) {
        // training-mode BatchNorm of this layer on complete columns (bn_small_fwd_kernel): z = acc + bias, biased two-pass variance
        const float bias = g.bias ? g.bias[nc] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = row0 + (r & 3) + 8 * (r >> 2);
            acc[r] += bias;
            if (m < g.M) {
                s += acc[r];
                if (nv) g.C[(long)m * g.ldc + n] = acc[r];
            }
        }
        const float mean = sk_col_sum(s, red, wave, li, lh) / rows_f;
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = row0 + (r & 3) + 8 * (r >> 2);
            const float d = acc[r] - mean;
            if (m < g.M) q += d * d;
        }
        const float var = sk_col_sum(q, red, wave, li, lh) / rows_f;
        const float invstd = 1.0f / sqrtf(var + g.eps);
        const float sc = g.gamma[nc] * invstd, sh = g.beta[nc] - mean * sc;
        if (wave == 0 && lh == 0 && nv) {
            g.mean[n] = mean; g.invstd[n] = invstd; g.scale[n] = sc; g.shift[n] = sh;
            if (g.mmean) {
                const float v = (g.unbiased && g.M > 1) ? var * (rows_f / (float)(g.M - 1)) : var;
                g.mmean[n] = g.mmean[n] * g.momentum + mean * (1.0f - g.momentum);
                g.mvar[n] = g.mvar[n] * g.momentum + v * (1.0f - g.momentum);
            }
        }
        if (g.a_out && nv) {
            const float sl = g.slope ? g.slope[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = row0 + (r & 3) + 8 * (r >> 2);
                const float y = acc[r] * sc + sh;
                if (m < g.M) g.a_out[(long)m * g.ldc + n] = g.relu ? (y > 0.f ? y : sl * y) : y;
            }
        }
        return;
    }
    if (EPI == XV_SK_BN_BWD) {
        // acc = d a of the BatchNorm(+activation) layer whose pre-BN tensor is g.z: its backward (bn_small_bwd_kernel) -> dz in C
        const float mu = g.mean[nc], is = g.invstd[nc], sc = g.scale[nc], sh = g.shift[nc];
        const float sl = g.slope ? g.slope[nc] : 0.f;
        float zn[16];
        float s1 = 0.f, s2 = 0.f, s4 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = row0 + (r & 3) + 8 * (r >> 2);
            const bool mv = m < g.M;
            const float zz = mv ? g.z[(long)m * g.ldc + nc] : 0.f;
            float dd = mv ? acc[r] : 0.f;
            if (g.relu) {
                const float y = zz * sc + sh;
                s4 += dd * fminf(y, 0.f);
                if (!(y > 0.f)) dd *= sl;
            }
            zn[r] = (zz - mu) * is;
            if (!mv) zn[r] = 0.f;
            acc[r] = dd;
            s1 += dd;
            s2 += dd * zn[r];
        }
        s1 = sk_col_sum(s1, red, wave, li, lh);
        s2 = sk_col_sum(s2, red, wave, li, lh);
        if (g.dalpha) s4 = sk_col_sum(s4, red, wave, li, lh);
        const float c1 = s1 / rows_f, c2 = s2 / rows_f;
        const float gg = g.gamma[nc] * is;
        if (wave == 0 && lh == 0 && nv) {
            g.dbeta[n] = s1; g.dgamma[n] = s2;
            if (g.dbias) g.dbias[n] = gg * (s1 - c1 * rows_f);
            if (g.dalpha) g.dalpha[n] = s4;
        }
        if (nv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = row0 + (r & 3) + 8 * (r >> 2);
                if (m < g.M) g.C[(long)m * g.ldc + n] = gg * (acc[r] - c1 - zn[r] * c2);
            }
        }
    }
}

int sk_target_wgs() { return 256; }      // workgroup target of the split policy: one per CU

}  // namespace

size_t xv_skinny_tickets(int max_n) { return (size_t)xv_cdiv(max_n, SK_COLS); }

int xv_launch_skinny(hipStream_t s, const XvSkinny& g) {
    XV_REQUIRE(g.M > 0 && g.M <= SK_ROWS && g.N > 0 && g.K > 0, "segment gemm: bad shape (M=%d N=%d K=%d, M <= %d)", g.M, g.N, g.K, SK_ROWS);
    XV_REQUIRE(g.K % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0, "segment gemm: K/lda/ldb must be multiples of 4 (K=%d lda=%ld ldb=%ld)", g.K, g.lda, g.ldb);
    XV_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.Bt % 16) == 0, "segment gemm: operands must be 16-byte aligned");
    XV_REQUIRE(g.epi >= XV_SK_PLAIN && g.epi <= XV_SK_BN_BWD, "segment gemm: bad epilogue %d", g.epi);
    SkArgs p;
    p.g = g;
    p.zero = xv_zero_page((size_t)g.K);
    if (!p.zero) return 1;
    p.tiles_n = xv_cdiv(g.N, SK_COLS);
    // two stages (64 k) per workgroup at least; as many splits as fill the chip once
    int splits = sk_target_wgs() / p.tiles_n;
    const int max_splits = std::max(1, g.K / (2 * SK_KS));
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    while (splits > 1 && (size_t)splits * p.tiles_n * SK_ROWS * SK_COLS * sizeof(float) > g.ws_bytes) --splits;
    p.k_chunk = (int)xv_align((size_t)xv_cdiv(g.K, splits), SK_KS);
    p.splits = xv_cdiv(g.K, p.k_chunk);
    p.slab = (float*)g.ws;
    XV_REQUIRE(p.splits == 1 || (g.tickets && g.ws), "segment gemm: split reduction needs a workspace and tickets");
    dim3 grid(p.tiles_n, p.splits, 1);
    switch (g.epi) {
    case XV_SK_PLAIN: hipLaunchKernelGGL(xv_skinny_kernel<XV_SK_PLAIN>, grid, dim3(256), 0, s, p); break;
    case XV_SK_BN_FWD:
        XV_REQUIRE(g.gamma && g.beta && g.mean && g.invstd && g.scale && g.shift, "segment gemm: BatchNorm forward needs its vectors");
        hipLaunchKernelGGL(xv_skinny_kernel<XV_SK_BN_FWD>, grid, dim3(256), 0, s, p);
        break;
    default:
        XV_REQUIRE(g.gamma && g.z && g.mean && g.invstd && g.scale && g.shift && g.dgamma && g.dbeta, "segment gemm: BatchNorm backward needs its vectors");
        hipLaunchKernelGGL(xv_skinny_kernel<XV_SK_BN_BWD>, grid, dim3(256), 0, s, p);
        break;
    }
    XV_LAUNCH_CHECK();
    return 0;
}

// C-ABI forms (include/xvector_hip.h)
extern "C" int xv_segment_gemm(void* stream, const float* a, long lda, const float* bt, long ldb, int m, int n, int k, const float* bias,
                               const float* row_coef, const float* row_norm, const float* xrow, long ldx, float* c, long ldc,
                               void* ws, size_t ws_bytes, uint32_t* tickets) {
    XV_REQUIRE(!row_coef || (row_norm && xrow), "segment_gemm: the row term needs row_norm and xrow");
    XvSkinny g = {};
    g.A = a; g.lda = lda; g.Bt = bt; g.ldb = ldb; g.M = m; g.N = n; g.K = k; g.bias = bias; g.C = c; g.ldc = ldc;
    g.row_coef = row_coef; g.row_norm = row_norm; g.X = xrow; g.ldx = ldx;
    g.epi = XV_SK_PLAIN; g.ws = ws; g.ws_bytes = ws_bytes; g.tickets = tickets;
    return xv_launch_skinny((hipStream_t)stream, g);
}

extern "C" int xv_segment_affine_bn_forward(void* stream, const float* x, long ldx, const float* wt, long ldw, int m, int n, int k,
                                            const float* bias, const float* gamma, const float* beta, float eps, float momentum,
                                            int unbiased_moving, float* moving_mean, float* moving_var, float* z, float* mean,
                                            float* invstd, float* scale, float* shift, int relu, float* a, void* ws, size_t ws_bytes,
                                            uint32_t* tickets) {
    XvSkinny g = {};
    g.A = x; g.lda = ldx; g.Bt = wt; g.ldb = ldw; g.M = m; g.N = n; g.K = k; g.bias = bias; g.C = z; g.ldc = n;
    g.epi = XV_SK_BN_FWD;
    g.gamma = gamma; g.beta = beta; g.eps = eps; g.momentum = momentum; g.unbiased = unbiased_moving; g.mmean = moving_mean; g.mvar = moving_var;
    g.mean = mean; g.invstd = invstd; g.scale = scale; g.shift = shift;
    g.relu = relu; g.slope = relu ? xv_act_context().slope : nullptr; g.a_out = a;
    g.ws = ws; g.ws_bytes = ws_bytes; g.tickets = tickets;
    return xv_launch_skinny((hipStream_t)stream, g);
}

extern "C" int xv_segment_dgrad_bn_backward(void* stream, const float* dy, long lddy, const float* wt, long ldw, int m, int n, int k,
                                            const float* row_coef, const float* row_norm, const float* xrow, long ldx, const float* z,
                                            const float* gamma, const float* mean, const float* invstd, const float* scale,
                                            const float* shift, int relu, float* dz, float* dgamma, float* dbeta, float* dbias,
                                            void* ws, size_t ws_bytes, uint32_t* tickets) {
    XV_REQUIRE(!row_coef || (row_norm && xrow), "segment_dgrad_bn_backward: the row term needs row_norm and xrow");
    XvSkinny g = {};
    g.A = dy; g.lda = lddy; g.Bt = wt; g.ldb = ldw; g.M = m; g.N = n; g.K = k; g.C = dz; g.ldc = n;
    g.row_coef = row_coef; g.row_norm = row_norm; g.X = xrow; g.ldx = ldx;
    g.epi = XV_SK_BN_BWD;
    g.z = z; g.gamma = gamma; g.mean = (float*)mean; g.invstd = (float*)invstd; g.scale = (float*)scale; g.shift = (float*)shift;
    const XvActContext act = xv_act_context();
    g.relu = relu; g.slope = relu ? act.slope : nullptr; g.dalpha = (relu && act.slope) ? act.dalpha : nullptr;
    g.dgamma = dgamma; g.dbeta = dbeta; g.dbias = dbias;
    g.ws = ws; g.ws_bytes = ws_bytes; g.tickets = tickets;
    return xv_launch_skinny((hipStream_t)stream, g);
}
