// Split-precision ("f16x3") MFMA GEMMs: fp32 tensors carried as TWO planes of fp16 pieces so the big
// contractions of the TDNN run on the 16-bit matrix cores (16x the fp32-input MFMA rate) at fp32-class
// accuracy.
//
//   x * s = h + l + eps,   h = fp16(x*s),  l = fp16(x*s - h),   |eps| <= 2^-22 |x*s|
//   a.b  ~=  (ha.hb + ha.lb + la.hb) / (sa*sb)      three fp16 MFMAs per block, fp32 accumulate: v_mfma_f32_16x16x32_f16 in the
//                                                   context-window and weight-gradient kernels, v_mfma_f32_32x32x16_f16 in the generic one
//
// s is a power of two per tensor (exact to apply and undo) chosen from the tensor's max |x| so that
// max |x*s| lies in [2^12, 2^13): 3 bits below the fp16 maximum, and the low piece of every element that
// matters stays a normal fp16.  The max is produced by whoever writes the tensor (exact output range of
// BN+ReLU from the GEMM epilogue's column min/max; a bound for the BN backward; a reduction for inputs and
// weights) and travels as the uint bits of a float in device memory - no host round trip.
// Measured on MI355X (tests/test_gpu_ops_f16x3.py, tools/bench_kernel.py gemm16): max error vs float64 5e-7..8e-7 of the output scale (fp32-input MFMA:
// 1.7e-7), 2.6-3.0x the speed of the fp32-input MFMA kernels of xv_gemm.hip.
//
// Plane layout: [2][rows][ld] 16-bit, channel axis contiguous, ld a multiple of 8 (16-byte chunks), zero padded.
// The spliced (context-window) row map of xv_gemm.hip applies unchanged to both kernels.
#include "xv_common.h"
#include "xv_epilogue.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

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

__device__ __forceinline__ void split_f16(float x, float s, u16& h, u16& l) {
    float xs = x * s;
    _Float16 hh = (_Float16)xs;
    _Float16 ll = (_Float16)(xs - (float)hh);
    h = __builtin_bit_cast(u16, hh);
    l = __builtin_bit_cast(u16, ll);
}

// ---------------------------------------------------------------------------------------------
// producers of planes
// ---------------------------------------------------------------------------------------------
// One atomicMax per WORKGROUP, and only when it can raise the value (same-address atomics serialise in L2:
// thousands of them cost more than the read of the tensor).
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, size_t count, unsigned* __restrict__ amax) {
    __shared__ float red[4];
    float m = 0.f;
    const size_t nq = count / 4, stride = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (((uintptr_t)x & 15) == 0) {
        for (size_t i = i0; i < nq; i += stride) {
            float4 v = ((const float4*)x)[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (size_t i = nq * 4 + i0; i < count; i += stride) m = fmaxf(m, fabsf(x[i]));
    } else {
        for (size_t i = i0; i < count; i += stride) m = fmaxf(m, fabsf(x[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        unsigned bits = __float_as_uint(m);
        if (bits > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, bits);
    }
}

extern "C" int xv_amax(void* stream, const float* x, size_t count, uint32_t* amax_accum) {
    XV_REQUIRE(x && amax_accum && count > 0, "amax: bad arguments");
    long blocks = (long)((count / 4 + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(amax_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, count, (unsigned*)amax_accum);
    XV_LAUNCH_CHECK();
    return 0;
}

__device__ __forceinline__ void store_split8(const float (&v)[8], float s, u16* __restrict__ hi, u16* __restrict__ lo) {
    u16 h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_f16(v[j], s, h[j], l[j]);
    *(uint4*)hi = *(const uint4*)h;
    *(uint4*)lo = *(const uint4*)l;
}

// dst planes [2][rows][ldd] <- src [rows][lds] (columns >= c zero).  8 elements (one 16-byte chunk per plane) per thread;
// VEC: c and lds are multiples of 4 and src is 16-byte aligned, so the 8 inputs are two float4 loads.
template <bool VEC>
__global__ void split_planes_kernel(const float* __restrict__ src, long rows, int c, long lds, u16* __restrict__ dst, long ldd,
                                    long plane_stride, const unsigned* __restrict__ amax) {
    const float s = xv_pow2_scale(*amax);
    const unsigned cq = (unsigned)(ldd / 8), total = (unsigned)rows * cq;      // < 2^31 (checked by the wrapper)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const long r = i / cq;
        int col = (int)(i - (unsigned)r * cq) * 8;
        float v[8];
        if (VEC) {
            float4 a = col < c ? *(const float4*)(src + r * lds + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 b = col + 4 < c ? *(const float4*)(src + r * lds + col + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (col + j) < c ? src[r * lds + col + j] : 0.f;
        }
        store_split8(v, s, dst + r * ldd + col, dst + plane_stride + r * ldd + col);
    }
}

extern "C" int xv_split_planes(void* stream, const float* src, int rows, int c, int lds, void* planes, int ldp, size_t plane_stride,
                               const uint32_t* amax) {
    XV_REQUIRE(src && planes && amax && rows > 0 && c > 0 && lds >= c, "split_planes: bad arguments");
    XV_REQUIRE(ldp % 8 == 0 && ldp >= c && ((uintptr_t)planes % 16) == 0 && plane_stride % 8 == 0, "split_planes: ldp must be a multiple of 8 and >= c");
    long total = (long)rows * (ldp / 8);
    XV_REQUIRE(total < (1L << 31), "split_planes: tensor too large for 32-bit indexing");
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    const bool vec = c % 4 == 0 && lds % 4 == 0 && ((uintptr_t)src % 16) == 0;
    hipLaunchKernelGGL(vec ? split_planes_kernel<true> : split_planes_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                       (long)rows, c, (long)lds, (u16*)planes, (long)ldp, (long)plane_stride, (const unsigned*)amax);
    XV_LAUNCH_CHECK();
    return 0;
}

// planes <- relu?(z*scale + shift)   (BN + ReLU output written directly as the next layer's operand planes)
// n and ldz are multiples of 4 (checked by the wrapper): two float4 loads of z per thread.
__global__ void bn_apply_split_kernel(const float* __restrict__ z, long rows, int n, long ldz, const float* __restrict__ scale,
                                      const float* __restrict__ shift, int relu, const unsigned* __restrict__ amax,
                                      u16* __restrict__ dst, long ldd, long plane_stride, const float* __restrict__ slope) {
    const float s = xv_pow2_scale(*amax);
    const unsigned cq = (unsigned)(ldd / 8), total = (unsigned)rows * cq;      // < 2^31 (checked by the wrapper)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const long r = i / cq;
        int col = (int)(i - (unsigned)r * cq) * 8;
        float v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = col + 4 * q;
            float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < n) {
                float4 zz = *(const float4*)(z + r * ldz + c), sc = *(const float4*)(scale + c), sh = *(const float4*)(shift + c);
                y.x = zz.x * sc.x + sh.x; y.y = zz.y * sc.y + sh.y; y.z = zz.z * sc.z + sh.z; y.w = zz.w * sc.w + sh.w;
                if (relu && slope) {       // prelu / leaky ReLU (act context, xv_common.h)
                    const float4 sl = *(const float4*)(slope + c);
                    y.x = y.x > 0.f ? y.x : y.x * sl.x; y.y = y.y > 0.f ? y.y : y.y * sl.y;
                    y.z = y.z > 0.f ? y.z : y.z * sl.z; y.w = y.w > 0.f ? y.w : y.w * sl.w;
                } else if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
            }
            v[4 * q] = y.x; v[4 * q + 1] = y.y; v[4 * q + 2] = y.z; v[4 * q + 3] = y.w;
        }
        store_split8(v, s, dst + r * ldd + col, dst + plane_stride + r * ldd + col);
    }
}

extern "C" int xv_bn_apply_split(void* stream, const float* z, int rows, int n, int ldz, const float* scale, const float* shift, int relu,
                                 const uint32_t* amax, void* planes, int ldp, size_t plane_stride) {
    XV_REQUIRE(rows > 0 && n > 0 && ldz >= n && ldp % 8 == 0 && ldp >= n && plane_stride % 8 == 0, "bn_apply_split: bad shape");
    XV_REQUIRE(n % 4 == 0 && ldz % 4 == 0 && ((uintptr_t)z % 16) == 0, "bn_apply_split: n and ldz must be multiples of 4 (n=%d ldz=%d)", n, ldz);
    long total = (long)rows * (ldp / 8);
    XV_REQUIRE(total < (1L << 31), "bn_apply_split: tensor too large for 32-bit indexing");
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_apply_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, (long)rows, n, (long)ldz, scale, shift,
                       relu, (const unsigned*)amax, (u16*)planes, (long)ldp, (long)plane_stride, relu ? xv_act_context().slope : nullptr);
    XV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// NT: C[m][n] = (sum_k A[rowmap(m)][k] * Bt[n][k]) / (sA*sB) + bias[n]
// 128x128 tile, 4 waves (2x2) x 64x64 = 2x2 v_mfma_f32_32x32x16_f16 accumulators, K-step 32, two operands x two
// planes x [128][32] fp16 LDS images (64 KB double-buffered, 2 workgroups per CU) filled by LDS-DMA; 16-byte
// chunks XOR-swizzled by (row>>2)&3 on the DMA source and the ds_read_b128 address (conflict-free).
// ---------------------------------------------------------------------------------------------
struct NT16Args {
    const u16* A; long lda; long a_plane; int a_rps; int a_pitch;
    const u16* Bt; long ldb; long b_plane;
    float* C; long ldc;
    int M, N, K;
    int tiles_m, tiles_n;
    const float* bias;
    float* part;
    const unsigned* a_amax; const unsigned* b_amax;
    const u16* zero;
    XvBwdStats bwd;         // EPI == 2 only
};

#ifndef XV16_BK
#define XV16_BK 32
#endif
#ifndef XV16_WGS
#define XV16_WGS 2
#endif
#ifndef XV16_WAVES
#define XV16_WAVES 4     // 4: 2x2 waves of 64x64;  8: 2x4 waves of 64x32 (4 waves per SIMD at 2 workgroups per CU)
#endif
#define WN16 (XV16_WAVES / 2)          // waves along N
#define NB16 (4 / WN16)                // 32-column accumulator blocks per wave
#define BK16 XV16_BK
#define SWZ16(row) (BK16 == 64 ? (((row) >> 1) & 7) : BK16 == 32 ? (((row) >> 2) & 3) : 0)
#define CQ16 (BK16 / 8)
#define PLANE_HALFS (128 * BK16)
#define BUF_HALFS (4 * PLANE_HALFS)

// the workgroup barrier of the staging pipelines: xv_dma16's loads are inline assembly the compiler does not count
__device__ __forceinline__ void xv16_sync() {
    xv_dma_wait_all();
    __syncthreads();
}

template <int EPI>      // 0: plain, 1: + forward BN statistics of the tile, 2: + BN backward reductions (data gradient)
__global__ __launch_bounds__(64 * XV16_WAVES, XV16_WGS) void xv_gemm16_nt_kernel(NT16Args p) {
    constexpr int RPI = 64 / CQ16;                 // tile rows per LDS-DMA wave-instruction (16)
    constexpr int IPW = 128 / RPI / XV16_WAVES;    // DMA instructions per wave per plane per operand
    __shared__ __attribute__((aligned(16))) u16 smem[2 * BUF_HALFS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int wr = wave / WN16, wc = wave % WN16;
    const int li = lane & 31, lh = lane >> 5;
    const int t = xcd_swizzle16(blockIdx.x, gridDim.x);
    const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int nk = (p.K + BK16 - 1) / BK16;

    const int lrow = lane / CQ16, lpos = lane % CQ16;
    // per-lane byte offsets from the plane bases, swizzled chunk folded in (xv_dma16: scalar base + 32-bit offset, xv_common.h; both planes
    // share them).  Rows outside the matrix read row 0 - their products are never stored or counted; k beyond K must read zeros and
    // takes the zero page through the builtin's 64-bit form (a ragged last K-step only).
    unsigned aoff[IPW], boff[IPW];
    int ksrc[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int row = RPI * (IPW * wave + i) + lrow;
        ksrc[i] = ((lpos ^ SWZ16(row)) << 3);
        int m = m0 + row;
        int mm = m < p.M ? m : 0;
        int seg = mm / p.a_rps, tt = mm - seg * p.a_rps;
        aoff[i] = (unsigned)((((long)seg * p.a_pitch + tt) * p.lda + ksrc[i]) * 2);
        int n = n0 + row;
        boff[i] = (unsigned)(((long)(n < p.N ? n : 0) * p.ldb + ksrc[i]) * 2);
    }
    const unsigned lds0 = xv_lds_addr(smem + RPI * IPW * uwave * BK16);
    auto gstage = [&](int kt, int buf) {
        const int k0 = kt * BK16;
        if (k0 + BK16 <= p.K) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const float* abase = (const float*)(p.A + pl * p.a_plane + k0);
                const float* bbase = (const float*)(p.Bt + pl * p.b_plane + k0);
#pragma unroll
                for (int i = 0; i < IPW; ++i) {
                    xv_dma16(abase, aoff[i], lds0 + (buf * BUF_HALFS + pl * PLANE_HALFS + RPI * i * BK16) * 2);
                    xv_dma16(bbase, boff[i], lds0 + (buf * BUF_HALFS + (2 + pl) * PLANE_HALFS + RPI * i * BK16) * 2);
                }
            }
            return;
        }
        u16* base = smem + buf * BUF_HALFS + RPI * IPW * uwave * BK16;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const bool kv = k0 + ksrc[i] < p.K;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const u16* pa = kv ? (const u16*)((const char*)(p.A + pl * p.a_plane + k0) + aoff[i]) : p.zero;
                const u16* pb = kv ? (const u16*)((const char*)(p.Bt + pl * p.b_plane + k0) + boff[i]) : p.zero;
                xv_dma16_ptr(pa, base + pl * PLANE_HALFS + RPI * i * BK16);
                xv_dma16_ptr(pb, base + (2 + pl) * PLANE_HALFS + RPI * i * BK16);
            }
        }
    };

    f32x16 acc[2][NB16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB16; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int fsw = SWZ16(li);
    const int a_row = (wr * 64 + li) * BK16, b_row = (wc * 32 * NB16 + li) * BK16;
    if (nk > 0) gstage(0, 0);
    xv16_sync();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gstage(kt + 1, buf ^ 1);
        const u16* base = smem + buf * BUF_HALFS;
#pragma unroll
        for (int kb = 0; kb < BK16 / 16; ++kb) {
            const int pos = (((2 * kb + lh) ^ fsw) << 3);
            f32x4 af[2][2], bf[2][NB16];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                af[pl][0] = *(const f32x4*)(base + pl * PLANE_HALFS + a_row + pos);
                af[pl][1] = *(const f32x4*)(base + pl * PLANE_HALFS + a_row + 32 * BK16 + pos);
#pragma unroll
                for (int nb = 0; nb < NB16; ++nb) bf[pl][nb] = *(const f32x4*)(base + (2 + pl) * PLANE_HALFS + b_row + nb * 32 * BK16 + pos);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NB16; ++b) {
#define MM(i, j) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][a]), __builtin_bit_cast(f16x8, bf[j][b]), acc[a][b], 0, 0, 0)
                    MM(0, 1); MM(1, 0); MM(0, 0);          // cross terms first, then the leading product
#undef MM
                }
        }
        xv16_sync();
    }

    const float out_scale = 1.0f / (xv_pow2_scale(p.a_amax ? *p.a_amax : 0u) * xv_pow2_scale(p.b_amax ? *p.b_amax : 0u));
    float bias_v[NB16];
#pragma unroll
    for (int b = 0; b < NB16; ++b) {
        int n = n0 + (wc * NB16 + b) * 32 + li;
        bias_v[b] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)      // scale + bias first, stores afterwards (see the 16x16 form above)
#pragma unroll
        for (int b = 0; b < NB16; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = acc[a][b][r] * out_scale + bias_v[b];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB16; ++b) asm volatile("" : "+v"(acc[a][b]));      // keeps hipcc from sinking the pass back into the store blocks
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB16; ++b) {
            int n = n0 + (wc * NB16 + b) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M && n < p.N) p.C[(long)m * p.ldc + n] = acc[a][b][r];
            }
        }
    if (EPI == 1) xv_tile_stats_epilogue(acc, (float*)smem, tid, wr, wc, li, lh, m0, n0, p.M, p.N, tile_m, p.tiles_m, p.part);
    if (EPI == 2) xv_tile_bwd_stats_epilogue(acc, (float*)smem, tid, wr, wc, li, lh, m0, n0, p.M, p.N, tile_m, p.bwd);
}

// ---------------------------------------------------------------------------------------------
// Context-window ("conv") form of the NT kernel: K = taps x C with A row (m, tap j) = x row xrow(m) + j.
// The generic kernel re-reads every x row once per tap from L2 (its A image for tap j+1 is the image for tap j
// shifted by one row) - and the L2 request rate is what bounds it (PMC: 92 % L2 hits, TCP stalled on pending
// requests 45 % of the time, MFMA busy 40 %).  Here one LDS image of the x rows xrow(m0) .. xrow(m0+127)+taps-1
// for a 32-channel chunk serves all taps: the K loop runs channel chunk outer, tap inner, the MFMA A fragments
// of tap j are read at LDS row offset j, and only the weight tile (B) is staged per K-step.  The next chunk's
// x rows are fetched in `taps` slices, one per K-step, so every step issues about the same number of LDS-DMAs
// (5 per wave instead of 8 for taps = 5).  Rows of one tile may straddle segments: aoff(q) = xrow(m0+q) - xrow(m0)
// grows by (pitch - rps) at every crossing; the launcher falls back to the generic kernel when the rows of a
// tile do not fit the (BM + 32)-row image.
// LDS (128-row tile): A 2 x 2 planes x 160 rows x 64 B = 40 KB, B 2 x 2 planes x 128 x 64 B = 32 KB: two workgroups per CU.
// ---------------------------------------------------------------------------------------------
#define CONV_BPLANE (128 * 32)
#define CONV_BBUF (2 * CONV_BPLANE)

struct NT16ConvArgs {
    NT16Args g;                // tiles_m counts BM-row tiles (BM = 64 * WR)
    int taps, chunks;          // K = taps * C, chunks = C / 32
    long a_rows;               // rows of the A planes (reads beyond are zero)
};

// WR = row-waves per workgroup: 2 -> 128-row tile, 4 waves, 72 KB LDS, two workgroups per CU;
//                               4 -> 256-row tile, 8 waves, 104 KB LDS, one workgroup per CU: the weight tile staged per
//                                    K-step serves twice the rows (staging is the largest non-MFMA cost of this kernel).
template <int WR> struct ConvGeom {
    static constexpr int BM = 64 * WR;
    static constexpr int NW = 2 * WR;                 // waves
    static constexpr int AR = BM + 32;                // x rows per A image (tile + halo)
    static constexpr int APLANE = AR * 32;            // halfs per A plane image
    static constexpr int ABUF = 2 * APLANE;
    static constexpr int NG = AR / 16;                // 16-row DMA groups per plane
    static constexpr int BG = 8 / NW;                 // weight-tile DMA groups per wave
    static constexpr int LDS_HALFS = 2 * ABUF + 2 * CONV_BBUF;
};

// The context-window kernel contracts on v_mfma_f32_16x16x32_f16 (one instruction spans the whole 32-channel K-step; 4 x 4 accumulator
// blocks of 16 x 16 per wave), not v_mfma_f32_32x32x16_f16.  Operands: lane l reads row base + (l & 15), 16-byte chunk
// l >> 4 of a 64-byte row; chunk ^= 2 * ((row >> 2) & 1) puts the 16 lanes of every ds_read_b128 lane group on 16 distinct 4-bank columns
// for ANY base row (exhaustive check over the four lane groups and all 16 alignments; a context-window read shifts the rows by the tap) -
// the same involution is applied to the LDS-DMA source chunk.
// [measured] tdnn2 / tdnn3 at S1: forward 236 -> 218 us / 309 -> 300 us, data gradient 242 -> 234 us / 286 -> 270 us (+3 ... +8 %).
#define CONV_SWZ(row) ((((row) >> 2) & 1) << 1)

template <int EPI, int WR>
__global__ __launch_bounds__(128 * WR, 2) void xv_gemm16_nt_conv_kernel(NT16ConvArgs q) {
    typedef ConvGeom<WR> G;
    const NT16Args& p = q.g;
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* const sA = smem;
    u16* const sB = smem + 2 * G::ABUF;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int t = xcd_swizzle16(blockIdx.x, gridDim.x);
    const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
    const int m0 = tile_m * G::BM, n0 = tile_n * 128;
    const int taps = q.taps, nc = q.chunks;
    const long C = p.lda;

    auto xrow = [&](int m) {
        int mm = min(m, p.M - 1);
        int seg = mm / p.a_rps, tt = mm - seg * p.a_rps;
        return (long)seg * p.a_pitch + tt;
    };
    const long xr0 = xrow(m0);
    // LDS row of output row (wr*64 + a*32 + li) at tap 0
    int arow[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) arow[a] = (int)(xrow(m0 + wr * 64 + a * 32 + li) - xr0);

    // DMA lane geometry: one wave-instruction = 16 image rows x 64 B; lane -> row l>>2, 16-byte chunk l&3 (source chunk swizzled)
    const int drow = lane >> 2, dchunk = lane & 3;
    // B: rows 16*(BG*wave + i) + drow of the weight tile.  Byte offsets from the plane bases (xv_dma16: scalar base + 32-bit lane offset,
    // both planes share them); rows outside the operands read row 0 - their products are never stored or counted.
    unsigned boff[G::BG];
#pragma unroll
    for (int i = 0; i < G::BG; ++i) {
        const int row = 16 * (G::BG * wave + i) + drow;
        const int n = n0 + row;
        boff[i] = (unsigned)(((long)(n < p.N ? n : 0) * p.ldb + ((dchunk ^ CONV_SWZ(row)) << 3)) * 2);
    }
    const unsigned lds_b = xv_lds_addr(sB + 16 * G::BG * uwave * 32), lds_a = xv_lds_addr(sA);
    auto stage_b = [&](int cc, int j, int buf) {
        const long k0 = (long)j * C + cc * 32;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const float* bbase = (const float*)(p.Bt + pl * p.b_plane + k0);
#pragma unroll
            for (int i = 0; i < G::BG; ++i) xv_dma16(bbase, boff[i], lds_b + (buf * CONV_BBUF + pl * CONV_BPLANE + 16 * i * 32) * 2);
        }
    };
    // A: NG row groups x 2 planes wave-instructions per chunk, id = plane*NG + group; slice `part` of `nparts`
    // hands ids part*NW + wave + s*NW*nparts to this wave.
    const int asrc = (dchunk ^ CONV_SWZ(drow)) << 3;      // (CONV_SWZ looks at bit 2 of the row: 16 * group does not touch it)
    auto stage_a = [&](int cc, int buf, int part, int nparts) {
        for (int id = part * G::NW + uwave; id < 2 * G::NG; id += G::NW * nparts) {
            const int pl = id >= G::NG ? 1 : 0, grp = id - G::NG * pl;
            const long xr = xr0 + 16 * grp + drow;
            const unsigned aoff = (unsigned)(((xr < q.a_rows ? xr : 0) * C + asrc) * 2);
            xv_dma16((const float*)(p.A + pl * p.a_plane + cc * 32), aoff, lds_a + (buf * G::ABUF + pl * G::APLANE + 16 * grp * 32) * 2);
        }
    };

    // (EPI == 2, the BN-backward epilogue, exists for the 32x32x16 accumulator layout only: the launcher refuses it in this build)
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lc = lane & 15, lg = lane >> 4;                 // fragment row / 8-channel chunk of this lane
    int arow16[4], b_off[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        arow16[qd] = (int)(xrow(m0 + wr * 64 + qd * 16 + lc) - xr0);
        const int rb = wc * 64 + qd * 16 + lc;
        b_off[qd] = rb * 32 + ((lg ^ CONV_SWZ(rb)) << 3);
    }
    stage_a(0, 0, 0, 1);
    stage_b(0, 0, 0);
    xv16_sync();
    int st = 0;
    for (int cc = 0; cc < nc; ++cc) {
        const u16* abase = sA + (cc & 1) * G::ABUF;
        for (int j = 0; j < taps; ++j, ++st) {
            if (j + 1 < taps) stage_b(cc, j + 1, (st + 1) & 1);
            else if (cc + 1 < nc) stage_b(cc + 1, 0, (st + 1) & 1);
            if (cc + 1 < nc) stage_a(cc + 1, (cc + 1) & 1, j, taps);
            const u16* bbase = sB + (st & 1) * CONV_BBUF;
            f32x4 af[2][4], bf[2][4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int ra = arow16[qd] + j;
                const int ao = ra * 32 + ((lg ^ CONV_SWZ(ra)) << 3);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    af[pl][qd] = *(const f32x4*)(abase + pl * G::APLANE + ao);
                    bf[pl][qd] = *(const f32x4*)(bbase + pl * CONV_BPLANE + b_off[qd]);
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#define MM(i, jj) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[i][a]), __builtin_bit_cast(f16x8, bf[jj][b]), acc[a][b], 0, 0, 0)
                    MM(0, 1); MM(1, 0); MM(0, 0);
#undef MM
                }
            xv16_sync();
        }
    }
    const float out_scale = 1.0f / (xv_pow2_scale(p.a_amax ? *p.a_amax : 0u) * xv_pow2_scale(p.b_amax ? *p.b_amax : 0u));
    // scale + bias in a pass of their own, THEN the stores: with both in one predicated block per element hipcc put the
    // s_waitcnt vmcnt(0) of the amax / bias loads in front of every store, and stores count on vmcnt - each store waited for
    // the previous one to retire
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int n = n0 + wc * 64 + b * 16 + lc;
        const float bias_v = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[a][b][jj] = acc[a][b][jj] * out_scale + bias_v;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(acc[a][b]));      // keeps hipcc from sinking the pass back into the store blocks
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int n = n0 + wc * 64 + b * 16 + lc;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = m0 + wr * 64 + a * 16 + lg * 4 + jj;
                if (m < p.M && n < p.N) p.C[(long)m * p.ldc + n] = acc[a][b][jj];
            }
    }
    {
        const int half = wr >> 1;
        const int tiles128 = (p.M + 127) / 128;
        float* red = (float*)smem + half * 1024;
        if (EPI == 1)
            xv_tile_stats_epilogue16(acc, red, tid & 255, wr & 1, wc, lane, m0 + 128 * half, n0, p.M, p.N, (WR / 2) * tile_m + half, tiles128, p.part);
    }
    (void)li; (void)lh;
}

#ifndef XV16_CONV
#define XV16_CONV 1
#endif

// Does the context-window kernel apply?  K = taps*lda with whole 32-channel chunks, and the x rows of any BM-row tile
// (one extra (pitch - rps) per segment crossing, plus the taps) fit the A image of BM + 32 rows.
static bool conv_form_applies(const XvGemm16NT& g, int bm, int* taps_out) {
    if (!XV16_CONV || XV16_WAVES != 4 || g.lda % 32 != 0 || g.K % g.lda != 0) return false;
    const int taps = (int)(g.K / g.lda);
    if (taps < 2 || g.a_pitch < g.a_rps) return false;
    const long crossings = (bm - 1) / g.a_rps + 1;
    const long span = (bm - 1) + (long)(g.a_pitch - g.a_rps) * crossings + (taps - 1);
    if (span >= bm + 32) return false;
    *taps_out = taps;
    return true;
}

// [measured] 256-row tiles (XV_CONV_WR=4) are 5-7 % slower at S1 (tdnn2/3 forward 270 vs 291 TF): 384 one-per-CU
// workgroups on 256 CUs leave half the chip idle for the second round, which costs more than the halved weight-tile
// staging saves.  128-row tiles stay the default.
#ifndef XV16_CONV_WR
#define XV16_CONV_WR 2
#endif

template <int WR>
static int launch_conv(hipStream_t s, const XvGemm16NT& g, NT16Args p, int taps, bool bwd) {
    typedef ConvGeom<WR> G;
    NT16ConvArgs q;
    p.tiles_m = xv_cdiv(g.M, G::BM);
    q.g = p; q.taps = taps; q.chunks = (int)(g.lda / 32);
    q.a_rows = (long)xv_cdiv(g.M, g.a_rps) * g.a_pitch;
    const size_t lds = (size_t)G::LDS_HALFS * sizeof(u16);
    static bool attr_done = false;
    if (!attr_done) {
        XV_CHECK_HIP(hipFuncSetAttribute((const void*)xv_gemm16_nt_conv_kernel<0, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        XV_CHECK_HIP(hipFuncSetAttribute((const void*)xv_gemm16_nt_conv_kernel<1, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        XV_CHECK_HIP(hipFuncSetAttribute((const void*)xv_gemm16_nt_conv_kernel<2, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    dim3 grid(p.tiles_m * p.tiles_n), block(128 * WR);
    if (g.bn_part) hipLaunchKernelGGL((xv_gemm16_nt_conv_kernel<1, WR>), grid, block, lds, s, q);
    else if (bwd) hipLaunchKernelGGL((xv_gemm16_nt_conv_kernel<2, WR>), grid, block, lds, s, q);
    else hipLaunchKernelGGL((xv_gemm16_nt_conv_kernel<0, WR>), grid, block, lds, s, q);
    XV_LAUNCH_CHECK();
    return 0;
}

int xv_launch_gemm16_nt(hipStream_t s, const XvGemm16NT& g) {
    XV_REQUIRE(g.lda % 8 == 0 && g.ldb % 8 == 0, "gemm16_nt: lda/ldb must be multiples of 8 (lda=%ld ldb=%ld)", g.lda, g.ldb);
    XV_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.Bt % 16) == 0 && g.a_plane % 8 == 0 && g.b_plane % 8 == 0,
               "gemm16_nt: planes must be 16-byte aligned");
    XV_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.a_rps > 0, "gemm16_nt: empty problem");
    // xv_dma16 addresses every row of a plane as a 32-bit byte offset from the plane's base
    XV_REQUIRE(((long)xv_cdiv(g.M, g.a_rps) * g.a_pitch + 1) * g.lda * 2 < (1L << 32) && ((long)g.N + 1) * g.ldb * 2 < (1L << 32),
               "gemm16_nt: a plane spans 4 GB or more (M=%d a_pitch=%d lda=%ld N=%d ldb=%ld): split the batch", g.M, g.a_pitch, g.lda, g.N, g.ldb);
    if (ensure_zero16()) return 1;
    NT16Args p;
    p.A = (const u16*)g.A; p.lda = g.lda; p.a_plane = g.a_plane; p.a_rps = g.a_rps; p.a_pitch = g.a_pitch;
    p.Bt = (const u16*)g.Bt; p.ldb = g.ldb; p.b_plane = g.b_plane;
    p.C = g.C; p.ldc = g.ldc; p.M = g.M; p.N = g.N; p.K = (int)xv_align(g.K, 8);
    p.tiles_m = xv_cdiv(g.M, 128); p.tiles_n = xv_cdiv(g.N, 128);
    p.bias = g.bias; p.part = g.bn_part; p.a_amax = g.a_amax; p.b_amax = g.b_amax; p.zero = g_zero16;
    const bool bwd = g.bwd_part != nullptr;
    XV_REQUIRE(!(bwd && g.bn_part), "gemm16_nt: one epilogue at a time");
    XV_REQUIRE(!bwd || (g.bwd_z && g.bwd_scale && g.bwd_shift && g.bwd_mean && g.bwd_invstd && XV16_WAVES == 4),
               "gemm16_nt: incomplete BN-backward epilogue arguments");
    p.bwd = XvBwdStats{g.bwd_z, g.bwd_scale, g.bwd_shift, g.bwd_mean, g.bwd_invstd, g.bwd_part};
    dim3 grid(p.tiles_m * p.tiles_n);
    XvProfScope prof(s, g.bn_part ? 3 : 4, 2.0 * g.M * g.N * g.K);
    int taps = 0;
    // XV_CONV_WR=4 (experiments, tests) forces 256-row tiles wherever they apply; as a build default they would be
    // limited to problems with at least one tile per CU
    const XvEnv* env = xv_env();
    if (!env) return 2;
    const bool wr4 = env->conv_wr ? env->conv_wr == 4 : (XV16_CONV_WR == 4 && g.M >= 256 * 256);
    // the BN-backward epilogue (an off-by-default experiment) is written for the 32x32x16 accumulator layout: with the 16x16x32
    // build of the context-window kernel such a launch takes the generic kernel
    const bool conv_ok = !bwd;
    if (conv_ok && wr4 && conv_form_applies(g, 256, &taps)) return launch_conv<4>(s, g, p, taps, bwd);
    if (conv_ok && conv_form_applies(g, 128, &taps)) return launch_conv<2>(s, g, p, taps, bwd);
    if (g.bn_part) hipLaunchKernelGGL(xv_gemm16_nt_kernel<1>, grid, dim3(64 * XV16_WAVES), 0, s, p);
    else if (bwd) hipLaunchKernelGGL(xv_gemm16_nt_kernel<2>, grid, dim3(64 * XV16_WAVES), 0, s, p);
    else hipLaunchKernelGGL(xv_gemm16_nt_kernel<0>, grid, dim3(64 * XV16_WAVES), 0, s, p);
    XV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// TN (weight gradients): P[split][m][n] = (sum_r A[amap(r)][m] * B[bmap(r)][n]) / (sA*sB)
// LDS image per plane: [32 reduction rows][128 columns] fp16, exactly as in HBM (LDS-DMA, no transpose).  The MFMA
// wants 8 consecutive r per lane at a fixed column - a column read: ds_read_b64_tr_b16 delivers a 4-row x 16-column
// block transposed (lane t of a 16-lane group gets column t of the 4 rows); two of them form one operand.
// Rows are 256 B = one full bank row apart, so the 32-byte column blocks are XOR-swizzled by the row
// (block ^= 2*(row&3)) on the DMA source address and on the read address: conflict-free.
// ---------------------------------------------------------------------------------------------

struct TN16Args {
    const u16* A; long lda; long a_plane; int a_pitch;
    const u16* B; long ldb; long b_plane; int b_pitch;
    int rps; float inv_rps;
    float* P;
    int M, N, R, r_chunk;
    int tiles_m, tiles_n;
    const unsigned* a_amax; const unsigned* b_amax;
    const u16* zero;
};

__global__ __launch_bounds__(256, 2) void xv_gemm16_tn_kernel(TN16Args p) {
    constexpr int BR = 32;
    constexpr int PH = BR * 128;                  // halfs per plane image
    constexpr int BH = 4 * PH;                    // per buffer: A planes, B planes
    __shared__ __attribute__((aligned(16))) u16 smem[2 * BH];
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
    // 16x16x32: the four 16-lane groups of a transposed read sit on rows 8g + {0..3} (+4) of one 32-byte column block; the block
    // position is XOR-ed with 2*(row&3) ^ ((row>>3)&1) so that the 8 (group, row) pairs of each half-wave take 8 distinct blocks
    const int scol = (((((dpos >> 1) ^ (2 * drow) ^ (wave & 1)) << 1) | (dpos & 1))) * 8;    // rows 8*wave + 4*i + drow: (row>>3)&1 = wave&1
    const bool a_cv = (m0 + scol) < p.M, b_cv = (n0 + scol) < p.N;
    auto gstage_ragged = [&](int kt, int buf) {      // a stage with rows at or beyond r_end: those read the zero page (they are summed)
        u16* base = smem + buf * BH;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rg = 2 * uwave + i;                     // image rows 4*rg .. 4*rg+3
            int r = r_begin + kt * BR + 4 * (2 * wave + i) + drow;
            bool rv = r < r_end;
            int seg = (int)((float)r * p.inv_rps);            // r / rps without an integer divide (r < 2^24)
            int tt = r - seg * p.rps;
            seg += (tt >= p.rps) - (tt < 0);
            tt = r - seg * p.rps;
            const long ao = ((long)seg * p.a_pitch + tt) * p.lda + m0 + scol;
            const long bo = ((long)seg * p.b_pitch + tt) * p.ldb + n0 + scol;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const u16* pa = (rv && a_cv) ? p.A + pl * p.a_plane + ao : p.zero;
                const u16* pb = (rv && b_cv) ? p.B + pl * p.b_plane + bo : p.zero;
                xv_dma16_ptr(pa, base + pl * PH + 4 * rg * 128);
                xv_dma16_ptr(pb, base + (2 + pl) * PH + 4 * rg * 128);
            }
        }
    };
    // Full stages: scalar plane bases + 32-bit lane offsets that advance by BR rows per stage (xv_dma16; the fp32 weight-gradient kernel's
    // scheme, xv_gemm.hip): a row of the spliced view is (segment, frame), frame += BR, and on crossing a segment's last frame the offset
    // skips the rows between two segments.  Columns outside the matrix read column 0: their products are never stored.
    const bool steady = p.rps >= BR;
    int tt_i[2];
    unsigned aoff[2], boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = min(r_begin + 4 * (2 * wave + i) + drow, p.R - 1);
        const int seg = r / p.rps;
        tt_i[i] = r - seg * p.rps;
        aoff[i] = (unsigned)((((long)seg * p.a_pitch + tt_i[i]) * p.lda + (a_cv ? m0 + scol : 0)) * 2);
        boff[i] = (unsigned)((((long)seg * p.b_pitch + tt_i[i]) * p.ldb + (b_cv ? n0 + scol : 0)) * 2);
    }
    const unsigned a_step = (unsigned)(BR * p.lda * 2), b_step = (unsigned)(BR * p.ldb * 2);
    const unsigned a_skip = (unsigned)((long)(p.a_pitch - p.rps) * p.lda * 2), b_skip = (unsigned)((long)(p.b_pitch - p.rps) * p.ldb * 2);
    const unsigned lds0 = xv_lds_addr(smem + 4 * 2 * uwave * 128);
    auto gstage = [&](int kt, int buf) {
        if (!steady || r_begin + (kt + 1) * BR > r_end) {
            gstage_ragged(kt, buf);
            return;
        }
        // (kt counts up by one per call, so the offsets are at stage kt here)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                xv_dma16((const float*)(p.A + pl * p.a_plane), aoff[i], lds0 + (buf * BH + pl * PH + 4 * i * 128) * 2);
                xv_dma16((const float*)(p.B + pl * p.b_plane), boff[i], lds0 + (buf * BH + (2 + pl) * PH + 4 * i * 128) * 2);
            }
            tt_i[i] += BR;
            const bool wrap = tt_i[i] >= p.rps;
            tt_i[i] -= wrap ? p.rps : 0;
            aoff[i] += a_step + (wrap ? a_skip : 0u);
            boff[i] += b_step + (wrap ? b_skip : 0u);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // transposed-read lane geometry: 16-lane group lg = lane>>4 = k group (rows 8*lg .. 8*lg+7 of the stage), the 16 lanes of a group
    // cover 4 rows x 16 columns (lane t ends up with column t of the 4 rows)
    const int lc = lane & 15, lg = lane >> 4;
    const int tq = lc >> 2, tp = lc & 3;
    auto tr_off = [&](int cb /* 16-column block 0..7 */, int rbase /* 0 | 4 */) {
        const int r = 8 * lg + rbase + tq;
        return r * 128 + ((cb ^ (2 * (r & 3)) ^ ((r >> 3) & 1)) << 4) + tp * 4;
    };
    typedef __attribute__((address_space(3))) s16x4* ltr_t;
    if (nk > 0) gstage(0, 0);
    xv16_sync();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gstage(kt + 1, buf ^ 1);
        const u16* base = smem + buf * BH;
        s16x8 af[2][4], bf[2][4];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const u16* ia = base + pl * PH;
                const u16* ib = base + (2 + pl) * PH;
                s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(ia + tr_off(4 * wr + qd, 0)));
                s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(ia + tr_off(4 * wr + qd, 4)));
                s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(ib + tr_off(4 * wc + qd, 0)));
                s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(ib + tr_off(4 * wc + qd, 4)));
                af[pl][qd] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                bf[pl][qd] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#define MM(i, j) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[i][a]), __builtin_bit_cast(f16x8, bf[j][b]), acc[a][b], 0, 0, 0)
                MM(0, 1); MM(1, 0); MM(0, 0);
#undef MM
            }
        xv16_sync();
    }
    const float out_scale = 1.0f / (xv_pow2_scale(p.a_amax ? *p.a_amax : 0u) * xv_pow2_scale(p.b_amax ? *p.b_amax : 0u));
    float* P = p.P + (long)split * p.M * p.N;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int n = n0 + wc * 64 + b * 16 + lc;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = m0 + wr * 64 + a * 16 + lg * 4 + jj;
                if (m < p.M && n < p.N) P[(long)m * p.N + n] = acc[a][b][jj] * out_scale;
            }
        }
    (void)li; (void)lh;
}

int xv_tn16_splits(int M, int N, int R) {
    int tiles = xv_cdiv(M, 128) * xv_cdiv(N, 128);
    int ksteps = xv_cdiv(R, 32);
    int splits = 512 / tiles;                      // 2 resident workgroups per CU (64 KB LDS each): one co-resident round
    if (splits > ksteps / 2) splits = ksteps / 2;
    if (splits < 1) splits = 1;
    int chunk = xv_cdiv(ksteps, splits) * 32;
    return xv_cdiv(R, chunk);
}

int xv_launch_gemm16_tn(hipStream_t s, const XvGemm16TN& g) {
    XV_REQUIRE(g.lda % 8 == 0 && g.ldb % 8 == 0 && g.M % 8 == 0 && g.N % 8 == 0, "gemm16_tn: lda/ldb/M/N must be multiples of 8 (M=%d N=%d)", g.M, g.N);
    XV_REQUIRE(g.R > 0 && g.R < (1 << 24) && g.rps > 0 && g.splits >= 1, "gemm16_tn: bad reduction shape");
    {   // xv_dma16 addresses every row of a plane as a 32-bit byte offset from the plane's base
        const long segs = xv_cdiv(g.R, g.rps);
        XV_REQUIRE((segs * g.a_pitch + 1) * g.lda * 2 < (1L << 32) && (segs * g.b_pitch + 1) * g.ldb * 2 < (1L << 32),
                   "gemm16_tn: a plane spans 4 GB or more (%ld segments, lda=%ld ldb=%ld): split the batch", segs, g.lda, g.ldb);
    }
    if (ensure_zero16()) return 1;
    TN16Args p;
    p.A = (const u16*)g.A; p.lda = g.lda; p.a_plane = g.a_plane; p.a_pitch = g.a_pitch;
    p.B = (const u16*)g.B; p.ldb = g.ldb; p.b_plane = g.b_plane; p.b_pitch = g.b_pitch;
    p.rps = g.rps;
    // gap-free rows (every dense layer: one-frame "segments") are one segment of R rows, so the K-steps take the cheap form (xv_launch_gemm_tn)
    if (g.a_pitch == g.rps && g.b_pitch == g.rps) { p.rps = g.R; p.a_pitch = g.R; p.b_pitch = g.R; }
    p.inv_rps = 1.0f / (float)p.rps;
    p.P = g.P; p.M = g.M; p.N = g.N; p.R = g.R;
    p.tiles_m = xv_cdiv(g.M, 128); p.tiles_n = xv_cdiv(g.N, 128);
    int ksteps = xv_cdiv(g.R, 32);
    p.r_chunk = xv_cdiv(ksteps, g.splits) * 32;
    XV_REQUIRE(xv_cdiv(g.R, p.r_chunk) == g.splits, "gemm16_tn: splits must come from xv_tn16_splits");
    p.a_amax = g.a_amax; p.b_amax = g.b_amax; p.zero = g_zero16;
    dim3 grid(p.tiles_m * p.tiles_n * g.splits);
    XvProfScope prof(s, 5, 2.0 * g.M * g.N * g.R);
    hipLaunchKernelGGL(xv_gemm16_tn_kernel, grid, dim3(256), 0, s, p);
    XV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Public op-level wrappers (include/xvector_hip.h, "split precision")
// ---------------------------------------------------------------------------------------------
extern "C" int xv_affine_forward_f16x3(void* stream, const void* x_planes, size_t x_plane_stride, const uint32_t* x_amax, int segs, int t_in,
                                       int c_ld, int k, const void* wt_planes, size_t wt_plane_stride, const uint32_t* wt_amax,
                                       const float* bias, float* z, int o, int ldz, float* bn_part) {
    XV_REQUIRE(segs > 0 && k >= 1 && t_in >= k && c_ld > 0 && o > 0 && ldz >= o, "affine_forward_f16x3: bad shape (t_in=%d k=%d)", t_in, k);
    XvGemm16NT g = {};
    g.A = x_planes; g.lda = c_ld; g.a_plane = (long)x_plane_stride; g.a_rps = t_in - k + 1; g.a_pitch = t_in;
    g.Bt = wt_planes; g.ldb = (long)k * c_ld; g.b_plane = (long)wt_plane_stride;
    g.C = z; g.ldc = ldz; g.M = segs * (t_in - k + 1); g.N = o; g.K = k * c_ld;
    g.bias = bias; g.bn_part = bn_part; g.a_amax = x_amax; g.b_amax = wt_amax;
    return xv_launch_gemm16_nt((hipStream_t)stream, g);
}

extern "C" int xv_affine_dgrad_f16x3(void* stream, const void* dz_planes, size_t dz_plane_stride, const uint32_t* dz_amax, int segs, int t_out,
                                     int o_ld, int k, const void* wf_planes, size_t wf_plane_stride, const uint32_t* wf_amax, float* dx,
                                     int c) {
    XV_REQUIRE(segs > 0 && k >= 1 && t_out >= 1 && o_ld > 0 && c > 0, "affine_dgrad_f16x3: bad shape");
    XvGemm16NT g = {};
    g.A = dz_planes; g.lda = o_ld; g.a_plane = (long)dz_plane_stride; g.a_rps = t_out + k - 1; g.a_pitch = t_out + 2 * (k - 1);
    g.Bt = wf_planes; g.ldb = (long)k * o_ld; g.b_plane = (long)wf_plane_stride;
    g.C = dx; g.ldc = c; g.M = segs * (t_out + k - 1); g.N = c; g.K = k * o_ld;
    g.a_amax = dz_amax; g.b_amax = wf_amax;
    return xv_launch_gemm16_nt((hipStream_t)stream, g);
}

// xv_affine_dgrad_f16x3 whose epilogue also produces the per-tile partials of the BN backward of the layer that owns dx
// (dx = d a of a BN+ReLU layer with pre-BN output z_below [rows][c]): part [ceil(rows/128)][3][c] = sum dd | sum dd*xhat | max |dd|.
extern "C" int xv_affine_dgrad_bnstats_f16x3(void* stream, const void* dz_planes, size_t dz_plane_stride, const uint32_t* dz_amax, int segs,
                                             int t_out, int o_ld, int k, const void* wf_planes, size_t wf_plane_stride,
                                             const uint32_t* wf_amax, float* dx, int c, const float* z_below, const float* scale,
                                             const float* shift, const float* mean, const float* invstd, float* part) {
    XV_REQUIRE(segs > 0 && k >= 1 && t_out >= 1 && o_ld > 0 && c > 0, "affine_dgrad_bnstats_f16x3: bad shape");
    XV_REQUIRE(z_below && scale && shift && mean && invstd && part, "affine_dgrad_bnstats_f16x3: null BN argument");
    XvGemm16NT g = {};
    g.A = dz_planes; g.lda = o_ld; g.a_plane = (long)dz_plane_stride; g.a_rps = t_out + k - 1; g.a_pitch = t_out + 2 * (k - 1);
    g.Bt = wf_planes; g.ldb = (long)k * o_ld; g.b_plane = (long)wf_plane_stride;
    g.C = dx; g.ldc = c; g.M = segs * (t_out + k - 1); g.N = c; g.K = k * o_ld;
    g.a_amax = dz_amax; g.b_amax = wf_amax;
    g.bwd_z = z_below; g.bwd_scale = scale; g.bwd_shift = shift; g.bwd_mean = mean; g.bwd_invstd = invstd; g.bwd_part = part;
    return xv_launch_gemm16_nt((hipStream_t)stream, g);
}

extern "C" int xv_affine_wgrad_f16x3(void* stream, const void* x_planes, size_t x_plane_stride, const uint32_t* x_amax, int segs, int t_in,
                                     int c_ld, int k, int c, const void* dz_planes, size_t dz_plane_stride, const uint32_t* dz_amax,
                                     int dz_seg_pitch, int dz_row0, int o_ld, int o, const float* kernel, float l2_scale, float* dkernel,
                                     void* ws, size_t ws_bytes) {
    XV_REQUIRE(segs > 0 && k >= 1 && t_in >= k && c_ld >= c && o_ld >= o && o_ld % 8 == 0 && c_ld % 8 == 0,
               "affine_wgrad_f16x3: bad shape (c_ld=%d and o_ld=%d must be multiples of 8)", c_ld, o_ld);
    const int t_out = t_in - k + 1;
    XvGemm16TN g = {};
    g.A = x_planes; g.lda = c_ld; g.a_plane = (long)x_plane_stride; g.a_pitch = t_in;
    g.B = (const u16*)dz_planes + (long)dz_row0 * o_ld; g.ldb = o_ld; g.b_plane = (long)dz_plane_stride; g.b_pitch = dz_seg_pitch;
    g.rps = t_out;
    g.M = k * c_ld; g.N = o_ld; g.R = segs * t_out;
    g.splits = xv_tn16_splits(g.M, g.N, g.R);
    XV_REQUIRE((size_t)g.splits * g.M * g.N * sizeof(float) <= ws_bytes, "affine_wgrad_f16x3: workspace too small (%zu needed)",
               (size_t)g.splits * g.M * g.N * sizeof(float));
    g.P = (float*)ws;
    g.a_amax = x_amax; g.b_amax = dz_amax;
    int rc = xv_launch_gemm16_tn((hipStream_t)stream, g);
    if (rc) return rc;
    return xv_launch_wgrad_reduce((hipStream_t)stream, g.P, g.splits, k, c, c_ld, o_ld, o, l2_scale != 0.f ? kernel : nullptr, o, l2_scale,
                                  dkernel, o);
}
