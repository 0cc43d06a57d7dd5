// Split-precision MFMA GEMM (experimental): fp32 operands carried as planes of 16-bit pieces so the
// contraction runs on the 16x faster 16-bit matrix cores while keeping fp32-class accuracy.
//
//   mode 3  "bf16x6": x = h + m + l, three bf16 pieces (8+8+8 mantissa bits); products hh, hm, mh, mm, hl, lh
//           (everything down to 2^-24 relative) accumulated in fp32 -> same error class as v_mfma_f32_32x32x2_f32.
//   mode 2  "f16x3":  x*s = h + l, two fp16 pieces (11+11 bits) with a power-of-two tensor scale s; products hh, hl, lh
//           (2^-22 relative), result multiplied by 1/(sA*sB).
//
// Operand planes: [PLANES][rows][ld] 16-bit, channel axis contiguous, ld a multiple of 8 (16-byte chunks),
// zero padded; the spliced (context-window) row map of xv_gemm.hip applies unchanged.
// Kernel: 128x128 tile, 4 waves (2x2) x 64x64 sub-tile = 2x2 v_mfma_f32_32x32x16_{bf16,f16}; K-step BK16 in
// {16,32}; LDS image per plane [128 rows][BK] 16-bit with an XOR chunk swizzle (conflict-free ds_read_b128),
// filled by LDS-DMA with the swizzle applied on the source address.
#include "xv_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#ifndef XV16_BK
#define XV16_BK 32
#endif
#ifndef XV16_NBUF
#define XV16_NBUF 2
#endif

// ---------------------------------------------------------------------------------------------
// fp32 -> planes
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u16 bf16_rne(float x) {
    __bf16 b = (__bf16)x;                       // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN-safe
    return __builtin_bit_cast(u16, b);
}
__device__ __forceinline__ float bf16_to_f32(u16 v) { return __uint_as_float((unsigned)v << 16); }

template <int MODE>
__global__ void split_planes_kernel(const float* __restrict__ src, long rows, int c, long lds, u16* __restrict__ dst, long ldd,
                                    long plane_stride, float scale) {
    long total = rows * ldd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / ldd;
        int col = (int)(i - r * ldd);
        float x = col < c ? src[r * lds + col] : 0.f;
        if (MODE == 3) {
            u16 h = bf16_rne(x);
            float r1 = x - bf16_to_f32(h);
            u16 m = bf16_rne(r1);
            float r2 = r1 - bf16_to_f32(m);
            u16 l = bf16_rne(r2);
            dst[i] = h; dst[plane_stride + i] = m; dst[2 * plane_stride + i] = l;
        } else {
            float xs = x * scale;
            _Float16 h = (_Float16)xs;
            _Float16 l = (_Float16)(xs - (float)h);
            dst[i] = __builtin_bit_cast(u16, h); dst[plane_stride + i] = __builtin_bit_cast(u16, l);
        }
    }
}

extern "C" int xvx_split_planes(void* stream, const float* src, int rows, int c, int lds, void* dst, int ldd, long plane_stride,
                                int mode, float scale) {
    XV_REQUIRE(mode == 2 || mode == 3, "split_planes: mode must be 2 (f16 pair) or 3 (bf16 triple)");
    XV_REQUIRE(ldd % 8 == 0 && ldd >= c, "split_planes: ld must be a multiple of 8");
    long total = (long)rows * ldd;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (mode == 3)
        hipLaunchKernelGGL(split_planes_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (long)rows, c, (long)lds,
                           (u16*)dst, (long)ldd, plane_stride, scale);
    else
        hipLaunchKernelGGL(split_planes_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (long)rows, c, (long)lds,
                           (u16*)dst, (long)ldd, plane_stride, scale);
    XV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// NT GEMM on planes
// ---------------------------------------------------------------------------------------------
struct NT16Args {
    const u16* A; long lda; long a_plane; int a_rps; int a_pitch;
    const u16* Bt; long ldb; long b_plane;
    float* C; long ldc;
    int M, N, K;
    int tiles_m, tiles_n;
    const float* bias;
    float out_scale;
    const u16* zero;
};

static u16* g_zero16 = nullptr;
static int ensure_zero16() {
    if (g_zero16) return 0;
    XV_CHECK_HIP(hipMalloc((void**)&g_zero16, 256));
    XV_CHECK_HIP(hipMemset(g_zero16, 0, 256));
    return 0;
}

__device__ __forceinline__ int xcd_swizzle16(int bid, int nwg) {
    int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int PLANES>
__global__ __launch_bounds__(256, 2) void xv_gemm16_nt_kernel(NT16Args p) {
    constexpr int BK16 = XV16_BK;                  // k per K-step (16-bit elements)
    constexpr int CQ = BK16 / 8;                   // 16-byte chunks per tile row (2 or 4)
    constexpr int RPI = 64 / CQ;                   // tile rows per LDS-DMA wave-instruction (32 or 16)
    constexpr int IPW = 128 / RPI / 4;             // DMA instructions per wave per plane per operand (1 or 2)
    constexpr int SW_SHIFT = (CQ == 4) ? 2 : 3;    // f(row) = (row >> SW_SHIFT) & (CQ-1)
    constexpr int PLANE_HALFS = 128 * BK16;        // 16-bit elements of one plane image
    constexpr int BUF_HALFS = 2 * PLANES * PLANE_HALFS;
    __shared__ __attribute__((aligned(16))) u16 smem[XV16_NBUF * BUF_HALFS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int t = xcd_swizzle16(blockIdx.x, gridDim.x);
    const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int nk = (p.K + BK16 - 1) / BK16;

    // DMA source bookkeeping: lane -> (row, chunk position); source chunk = pos ^ f(row)
    const int lrow = lane / CQ, lpos = lane % CQ;
    long aoff[IPW], boff[IPW];
    bool av[IPW], bv[IPW];
    int ksrc[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int row = RPI * (IPW * wave + i) + lrow;
        ksrc[i] = ((lpos ^ ((row >> SW_SHIFT) & (CQ - 1))) << 3);
        int m = m0 + row;
        av[i] = m < p.M;
        int mm = av[i] ? m : 0;
        int seg = mm / p.a_rps, tt = mm - seg * p.a_rps;
        aoff[i] = ((long)seg * p.a_pitch + tt) * p.lda;
        int n = n0 + row;
        bv[i] = n < p.N;
        boff[i] = (long)(bv[i] ? n : 0) * p.ldb;
    }
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto gstage = [&](int kt, int buf) {
        u16* base = smem + buf * BUF_HALFS + RPI * IPW * uwave * BK16;
        const int k0 = kt * BK16;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int k = k0 + ksrc[i];
            const bool kv = k < p.K;
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
                const u16* pa = (kv && av[i]) ? p.A + pl * p.a_plane + aoff[i] + k : p.zero;
                const u16* pb = (kv && bv[i]) ? p.Bt + pl * p.b_plane + boff[i] + k : p.zero;
                __builtin_amdgcn_global_load_lds((gptr_t)pa, (lptr_t)(base + pl * PLANE_HALFS + RPI * i * BK16), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)pb, (lptr_t)(base + (PLANES + pl) * PLANE_HALFS + RPI * i * BK16), 16, 0, 0);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int fsw = (li >> SW_SHIFT) & (CQ - 1);
    const int a_row = (wr * 64 + li) * BK16, b_row = (wc * 64 + li) * BK16;

    auto compute = [&](int buf) {
        const u16* base = smem + buf * BUF_HALFS;
#pragma unroll
        for (int kb = 0; kb < BK16 / 16; ++kb) {
            const int pos = (((2 * kb + lh) ^ fsw) << 3);
            f32x4 af[PLANES][2], bf[PLANES][2];
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
                af[pl][0] = *(const f32x4*)(base + pl * PLANE_HALFS + a_row + pos);
                af[pl][1] = *(const f32x4*)(base + pl * PLANE_HALFS + a_row + 32 * BK16 + pos);
                bf[pl][0] = *(const f32x4*)(base + (PLANES + pl) * PLANE_HALFS + b_row + pos);
                bf[pl][1] = *(const f32x4*)(base + (PLANES + pl) * PLANE_HALFS + b_row + 32 * BK16 + pos);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (PLANES == 3) {
#define MM3(i, j) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i][a]), __builtin_bit_cast(bf16x8, bf[j][b]), acc[a][b], 0, 0, 0)
                        MM3(0, 2); MM3(2, 0); MM3(1, 1); MM3(0, 1); MM3(1, 0); MM3(0, 0);   // smallest terms first
#undef MM3
                    } else {
#define MM2(i, j) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][a]), __builtin_bit_cast(f16x8, bf[j][b]), acc[a][b], 0, 0, 0)
                        MM2(0, 1); MM2(1, 0); MM2(0, 0);
#undef MM2
                    }
                }
        }
    };

    if (XV16_NBUF == 2) {
        if (nk > 0) gstage(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) gstage(kt + 1, buf ^ 1);
            compute(buf);
            __syncthreads();
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            gstage(kt, 0);
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    }

    float bias_v[2] = {0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int n = n0 + wc * 64 + b * 32 + li;
            bias_v[b] = n < p.N ? p.bias[n] : 0.f;
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int n = n0 + wc * 64 + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M && n < p.N) p.C[(long)m * p.ldc + n] = acc[a][b][r] * p.out_scale + bias_v[b];
            }
        }
}

extern "C" int xvx_gemm16_nt(void* stream, const void* A, long lda, long a_plane, int a_rps, int a_pitch, const void* Bt, long ldb,
                             long b_plane, float* C, long ldc, int M, int N, int K, const float* bias, int mode, float out_scale) {
    XV_REQUIRE(mode == 2 || mode == 3, "gemm16_nt: mode must be 2 or 3");
    XV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && K % 8 == 0, "gemm16_nt: lda/ldb/K must be multiples of 8");
    if (ensure_zero16()) return 1;
    NT16Args p;
    p.A = (const u16*)A; p.lda = lda; p.a_plane = a_plane; p.a_rps = a_rps; p.a_pitch = a_pitch;
    p.Bt = (const u16*)Bt; p.ldb = ldb; p.b_plane = b_plane;
    p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.tiles_m = xv_cdiv(M, 128); p.tiles_n = xv_cdiv(N, 128);
    p.bias = bias; p.out_scale = out_scale; p.zero = g_zero16;
    dim3 grid(p.tiles_m * p.tiles_n);
    if (mode == 3) hipLaunchKernelGGL(xv_gemm16_nt_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(xv_gemm16_nt_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
    XV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// TN GEMM on planes (weight gradients): P[split][m][n] = sum_r A[rowmap_a(r)][m] * B[rowmap_b(r)][n]
// LDS image per plane: [BR reduction rows][128 columns] 16-bit, exactly as in HBM (filled by LDS-DMA).  The MFMA
// wants 8 consecutive r per lane at a fixed column, i.e. a column read: ds_read_b64_tr_b16 delivers a 4-row x
// 16-column block transposed (lane t of a 16-lane group gets column t of the 4 rows), two of them make one
// operand.  Rows are 256 B = one full bank row apart, so 32-byte column blocks are XOR-swizzled by the row
// (block ^= 2*(row&3)) - applied on the DMA source address and on the read address.
// ---------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct TN16Args {
    const u16* A; long lda; long a_plane; int a_pitch;
    const u16* B; long ldb; long b_plane; int b_pitch;
    int rps; float inv_rps;
    float* P;
    int M, N, R, r_chunk;
    int tiles_m, tiles_n;
    float out_scale;
    const u16* zero;
};

template <int PLANES>
__global__ __launch_bounds__(256, 2) void xv_gemm16_tn_kernel(TN16Args p) {
    constexpr int BR = 32;                               // reduction rows per K-step
    constexpr int PLANE_HALFS = BR * 128;
    constexpr int BUF_HALFS = 2 * PLANES * PLANE_HALFS;
    __shared__ __attribute__((aligned(16))) u16 smem[2 * BUF_HALFS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int v = xcd_swizzle16(blockIdx.x, gridDim.x);
    const int tiles = p.tiles_m * p.tiles_n;
    const int split = v / tiles, t = v - split * tiles;
    const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int r_begin = split * p.r_chunk;
    const int r_end = min(p.R, r_begin + p.r_chunk);
    const int nk = (r_end - r_begin + BR - 1) / BR;

    // DMA: one wave-instruction = 4 image rows x 256 B; lane -> row l>>4, 16-byte chunk position l&15
    const int drow = lane >> 4, dpos = lane & 15;
    // source 16-byte chunk for that position: 32-byte block index (pos>>1) ^ 2*(row&3); rows 4*(..)+drow => row&3 == drow
    const int schunk = ((((dpos >> 1) ^ (2 * drow)) << 1) | (dpos & 1));
    const int scol = schunk * 8;                          // 16-bit elements
    const bool a_cv = (m0 + scol) < p.M, b_cv = (n0 + scol) < p.N;
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto gstage = [&](int kt, int buf) {
        u16* base = smem + buf * BUF_HALFS;
#pragma unroll
        for (int i = 0; i < BR / 16; ++i) {              // 8 DMA row-groups per plane image, 2 per wave
            const int rg = (BR / 16) * uwave + i;         // row group: image rows 4*rg .. 4*rg+3
            int r = r_begin + kt * BR + 4 * ((BR / 16) * wave + i) + drow;
            bool rv = r < r_end;
            int seg = (int)((float)r * p.inv_rps);
            int tt = r - seg * p.rps;
            seg += (tt >= p.rps) - (tt < 0);
            tt = r - seg * p.rps;
            const long ao = ((long)seg * p.a_pitch + tt) * p.lda + m0 + scol;
            const long bo = ((long)seg * p.b_pitch + tt) * p.ldb + n0 + scol;
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
                const u16* pa = (rv && a_cv) ? p.A + pl * p.a_plane + ao : p.zero;
                const u16* pb = (rv && b_cv) ? p.B + pl * p.b_plane + bo : p.zero;
                __builtin_amdgcn_global_load_lds((gptr_t)pa, (lptr_t)(base + pl * PLANE_HALFS + 4 * rg * 128), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)pb, (lptr_t)(base + (PLANES + pl) * PLANE_HALFS + 4 * rg * 128), 16, 0, 0);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transposed-read lane geometry: 16-lane group g = lane>>4: column half (g&1), k half (g>>1) == lh
    const int tq = (lane & 15) >> 2, tp = lane & 3;       // row q and 4-column piece p inside the 4x16 block
    const int ghalf = (lane >> 4) & 1;
    auto tr_addr = [&](int colbase /* multiple of 32 */, int row /* image row of the block's first row */) {
        // block of 16 columns = 32 B: index within the 128-column row
        const int blk = (colbase >> 4) + ghalf;
        const int r = row + tq;
        const int sblk = blk ^ (2 * (r & 3));
        return r * 128 + sblk * 16 + tp * 4;
    };
    auto compute = [&](int buf) {
        const u16* base = smem + buf * BUF_HALFS;
#pragma unroll
        for (int kb = 0; kb < BR / 16; ++kb) {
            s16x8 af[PLANES][2], bf[PLANES][2];
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int row0 = kb * 16 + 8 * lh;
                    const u16* ia = base + pl * PLANE_HALFS;
                    const u16* ib = base + (PLANES + pl) * PLANE_HALFS;
                    s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ia + tr_addr(wr * 64 + s * 32, row0)));
                    s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ia + tr_addr(wr * 64 + s * 32, row0 + 4)));
                    s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ib + tr_addr(wc * 64 + s * 32, row0)));
                    s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ib + tr_addr(wc * 64 + s * 32, row0 + 4)));
                    af[pl][s] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    bf[pl][s] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (PLANES == 3) {
#define MM3(i, j) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i][a]), __builtin_bit_cast(bf16x8, bf[j][b]), acc[a][b], 0, 0, 0)
                        MM3(0, 2); MM3(2, 0); MM3(1, 1); MM3(0, 1); MM3(1, 0); MM3(0, 0);
#undef MM3
                    } else {
#define MM2(i, j) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][a]), __builtin_bit_cast(f16x8, bf[j][b]), acc[a][b], 0, 0, 0)
                        MM2(0, 1); MM2(1, 0); MM2(0, 0);
#undef MM2
                    }
                }
        }
    };

    if (nk > 0) gstage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gstage(kt + 1, buf ^ 1);
        compute(buf);
        __syncthreads();
    }

    float* P = p.P + (long)split * p.M * p.N;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int n = n0 + wc * 64 + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M && n < p.N) P[(long)m * p.N + n] = acc[a][b][r] * p.out_scale;
            }
        }
}

extern "C" int xvx_gemm16_tn(void* stream, const void* A, long lda, long a_plane, int a_pitch, const void* B, long ldb, long b_plane,
                             int b_pitch, int rps, float* P, int M, int N, int R, int splits, int mode, float out_scale) {
    XV_REQUIRE(mode == 2 || mode == 3, "gemm16_tn: mode must be 2 or 3");
    XV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && M % 8 == 0 && N % 8 == 0, "gemm16_tn: lda/ldb/M/N must be multiples of 8");
    XV_REQUIRE(R < (1 << 24) && rps > 0 && splits >= 1, "gemm16_tn: bad reduction shape");
    if (ensure_zero16()) return 1;
    TN16Args p;
    p.A = (const u16*)A; p.lda = lda; p.a_plane = a_plane; p.a_pitch = a_pitch;
    p.B = (const u16*)B; p.ldb = ldb; p.b_plane = b_plane; p.b_pitch = b_pitch;
    p.rps = rps; p.inv_rps = 1.0f / (float)rps;
    p.P = P; p.M = M; p.N = N; p.R = R;
    p.tiles_m = xv_cdiv(M, 128); p.tiles_n = xv_cdiv(N, 128);
    int ksteps = xv_cdiv(R, 32);
    p.r_chunk = xv_cdiv(ksteps, splits) * 32;
    int nsplit = xv_cdiv(R, p.r_chunk);
    XV_REQUIRE(nsplit == splits, "gemm16_tn: splits %d does not divide the reduction evenly (got %d)", splits, nsplit);
    p.out_scale = out_scale; p.zero = g_zero16;
    dim3 grid(p.tiles_m * p.tiles_n * splits);
    if (mode == 3) hipLaunchKernelGGL(xv_gemm16_tn_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(xv_gemm16_tn_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
    XV_LAUNCH_CHECK();
    return 0;
}
