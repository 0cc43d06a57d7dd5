// HBM-bound kernels of the TDNN hot path: layout prep, BatchNorm (statistics finalise, apply,
// backward), ReLU, statistics pooling, l2_scaling, bias gradients, optimisers.  gfx950 only.
// All tensors fp32 row-major with the channel axis contiguous, so threads always run along
// channels (16 B per lane where the pitch allows) and reductions over rows are per-thread
// serial + wave shuffle / LDS combine.  No atomics on any path that feeds a gradient.
#include <stdarg.h>

#include <algorithm>

#include "xv_common.h"
#include "xv_epilogue.h"

// ------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void xv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
// ---- environment switches (xv_common.h XvEnv) -----------------------------------------------------------------------------------
extern char** environ;
namespace {
struct XvEnvState { XvEnv env; bool bad; char why[256]; };
// Read once, on first use; a C++11 function-local static is initialised under a lock, so concurrent first calls (loader threads, two
// engines created from two host threads) see one parse.
XvEnvState xv_env_parse() {
    XvEnvState st;
    st.bad = false;
    st.why[0] = 0;
    XvEnv& env = st.env;
    // every XV_* name this package reads anywhere (native library, Python host, tools/, tests/).  Any other XV_* variable is reported ONCE on
    // stderr and otherwise ignored: it may be a typo or a switch of an earlier round (worth a line), but it may equally belong to another
    // program in the user's environment, which must not stop a training run.  A known switch with a value it does not understand still fails.
    static const char* known[] = {"XV_SEGMENT_FUSED", "XV_NT_SCHED", "XV_CONV_WR", "XV_PRECISION", "XV_LOADER", "XV_LOADER_PIN", "XV_SHARE_GPU",
                                  "XV_LIB", "XV_TUNE_TIMES", "XV_DATA_SCALE", "XV_B", "XV_PROBE_OPS", "XV_PROBE_ONLY", "XV_PROBE_PERIODS",
                                  "XV_DIAG_M", "XV_DIAG_N", "XV_DIAG_K", "XV_DIAG_REPS", "XV_DIAG_B", "XV_PROBE_EXTRA", "XV_DZ_SLOTS"};
    for (char** e = environ; e && *e; ++e) {
        if (strncmp(*e, "XV_", 3) != 0) continue;
        const char* eq = strchr(*e, '=');
        const size_t len = eq ? (size_t)(eq - *e) : strlen(*e);
        bool ok = false;
        for (const char* k : known) ok = ok || (strlen(k) == len && strncmp(k, *e, len) == 0);
        if (!ok) fprintf(stderr, "libxvector_hip: ignoring unknown environment switch %.*s (INTEGRATION.md section 6 lists the supported ones)\n", (int)len, *e);
    }
    auto fail = [&](const char* fmt, const char* name, const char* v) {
        if (!st.bad) { snprintf(st.why, sizeof st.why, fmt, name, v); st.bad = true; }
    };
    auto flag = [&](const char* name, int dflt, int* out) {
        const char* v = getenv(name);
        *out = dflt;
        if (!v || !*v) return;
        if (!strcmp(v, "0") || !strcmp(v, "1")) *out = v[0] - '0';
        else fail("%s=%s: expected 0 or 1", name, v);
    };
    flag("XV_SEGMENT_FUSED", 1, &env.segment_fused);
    env.nt_sched = 0;
    if (const char* v = getenv("XV_NT_SCHED")) {
        if (!strcmp(v, "dp")) env.nt_sched = 1;
        else if (!strcmp(v, "sk")) env.nt_sched = 2;
        else if (*v) fail("%s=%s: expected dp or sk", "XV_NT_SCHED", v);
    }
    env.dz_slots = 0;
    if (const char* v = getenv("XV_DZ_SLOTS")) {
        if (!strcmp(v, "2")) env.dz_slots = 2;
        else if (*v) fail("%s=%s: expected 2", "XV_DZ_SLOTS", v);
    }
    env.conv_wr = 0;
    if (const char* v = getenv("XV_CONV_WR")) {
        if (!strcmp(v, "4")) env.conv_wr = 4;
        else if (*v && strcmp(v, "2")) fail("%s=%s: expected 2 or 4", "XV_CONV_WR", v);
    }
    return st;
}
}  // namespace

const XvEnv* xv_env() {
    static const XvEnvState st = xv_env_parse();
    if (st.bad) { xv_set_error("%s", st.why); return nullptr; }
    return &st.env;
}

extern "C" const char* xv_last_error(void) { return g_err; }
extern "C" int xv_abi_version(void) { return XV_ABI_VERSION; }
extern "C" int xv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static inline int grid_for(long total, int block, int cap = 4096) {
    long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int xv_copy_2d(void* stream, float* dst, size_t ldd, const float* src, size_t lds, int rows, int cols) {
    XV_REQUIRE(dst && src && rows > 0 && cols > 0 && ldd >= (size_t)cols && lds >= (size_t)cols, "copy_2d: bad arguments");
    XV_CHECK_HIP(hipMemcpy2DAsync(dst, ldd * sizeof(float), src, lds * sizeof(float), (size_t)cols * sizeof(float), rows,
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// ------------------------------------------------------------------------------------
// layout prep
// ------------------------------------------------------------------------------------
__global__ void pad_channels_kernel(const float* __restrict__ src, long rows, int c_src, float* __restrict__ dst, int c_dst) {
    XV_EW_FILLER();
    long total = rows * c_dst;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / c_dst;
        int c = (int)(i - r * c_dst);
        dst[i] = c < c_src ? src[r * c_src + c] : 0.f;
    }
}

extern "C" int xv_pad_channels(void* stream, const float* src, int rows, int c_src, float* dst, int c_dst) {
    XV_REQUIRE(rows > 0 && c_src > 0 && c_dst >= c_src, "pad_channels: bad shape rows=%d c_src=%d c_dst=%d", rows, c_src, c_dst);
    long total = (long)rows * c_dst;
    hipLaunchKernelGGL(pad_channels_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, (long)rows, c_src, dst, c_dst);
    XV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------
// Kaldi 'CM ' compressed-matrix decode on the GPU (kaldi_io.py:768-867 / compressed-matrix.h) for batches the native loader delivers
// packed (include/xvector_io.h: per chunk [min f32][range f32][D x (p0, p25, p75, p100) u16][D x T u8, column after column], padded to
// `stride` bytes).  One workgroup per chunk: the column parameters once, then tiles of CMD_TT frames - bytes in along the frame axis
// (how they are stored), floats out along the feature axis (how [b][t][d] is stored), transposed through LDS.
// The arithmetic is the reference codec's, operation by operation in float with no contraction (fp contract off), so the result is
// bit-identical to the host decoder (xv_loader.cpp, -ffp-contract=off) and to the reference reader.
// ------------------------------------------------------------------------------------
#define CMD_MAX_D 128
#define CMD_TT 128
// Ragged form (batched extraction: whole utterances of different lengths): chunk i starts at byte offs[i], holds rows[i] frames (its bytes
// are [D][rows[i]]) and is written to out[i][0 .. rows[i]) of a [b][T][D] tensor whose remaining rows are zeroed.  hdr = bytes in front
// of the column headers: 8 (min, range: the native loader's packing) or 16 (min, range, rows, cols: the matrix as it sits in the archive).
__global__ __launch_bounds__(256) void cm_decode_kernel(const uint8_t* __restrict__ packed, long stride, int T, int D, float* __restrict__ out,
                                                        const long* __restrict__ offs, const int* __restrict__ rows, int hdr) {
#pragma clang fp contract(off)
    XV_EW_FILLER();
    __shared__ float prm[6][CMD_MAX_D];                  // p0, p25, p75, s_lo, s_mid, s_hi per column
    __shared__ uint8_t tile[CMD_MAX_D][CMD_TT + 4];
    const uint8_t* chunk = packed + (offs ? offs[blockIdx.x] : (long)blockIdx.x * stride);
    const int tid = threadIdx.x;
    const int Tout = T;
    if (rows) {
        T = min(T, rows[blockIdx.x]);
        float* pad = out + ((long)blockIdx.x * Tout + T) * D;
        for (long i = tid; i < (long)(Tout - T) * D; i += 256) pad[i] = 0.f;
    }
    float minv, range;
    memcpy(&minv, chunk, 4);
    memcpy(&range, chunk + 4, 4);
    // plain operators under "fp contract(off)": HIP's __fmul_rn / __fadd_rn are inline functions compiled with the default contraction,
    // and their multiply-adds get fused into v_fma_f32 after inlining (1 ulp off the codec)
    const float gs = range * 1.52590218966964e-05f;        // 1/65535
    if (tid < D) {
        unsigned short h[4];
        memcpy(h, chunk + hdr + 8 * tid, 8);
        const float p0 = minv + gs * (float)h[0], p25 = minv + gs * (float)h[1];
        const float p75 = minv + gs * (float)h[2], p100 = minv + gs * (float)h[3];
        prm[0][tid] = p0; prm[1][tid] = p25; prm[2][tid] = p75;
        prm[3][tid] = (p25 - p0) / 64.0f;
        prm[4][tid] = (p75 - p25) / 128.0f;
        prm[5][tid] = (p100 - p75) / 63.0f;
    }
    const uint8_t* bytes = chunk + hdr + 8 * (long)D;
    float* o = out + (long)blockIdx.x * Tout * D;
    for (int t0 = 0; t0 < T; t0 += CMD_TT) {
        const int tt_n = min(CMD_TT, T - t0);
        __syncthreads();
        for (int idx = tid; idx < D * CMD_TT; idx += 256) {
            const int d = idx / CMD_TT, tt = idx - d * CMD_TT;
            if (tt < tt_n) tile[d][tt] = bytes[(long)d * T + t0 + tt];
        }
        __syncthreads();
        for (int idx = tid; idx < tt_n * D; idx += 256) {
            const int tt = idx / D, d = idx - tt * D;
            const uint8_t b = tile[d][tt];
            const float v = (float)b;
            float y;
            if (b <= 64) y = prm[0][d] + prm[3][d] * v;
            else if (b <= 192) y = prm[1][d] + prm[4][d] * (v - 64.0f);
            else y = prm[2][d] + prm[5][d] * (v - 192.0f);
            o[(long)(t0 + tt) * D + d] = y;
        }
    }
}

extern "C" int xv_cm_decode(void* stream, const uint8_t* packed, int b, int t, int d, size_t chunk_stride, float* out) {
    XV_REQUIRE(packed && out && b > 0 && t > 0 && d > 0, "cm_decode: bad arguments");
    XV_REQUIRE(d <= CMD_MAX_D, "cm_decode: at most %d feature dimensions (got %d)", CMD_MAX_D, d);
    XV_REQUIRE(chunk_stride >= (size_t)8 + 8 * (size_t)d + (size_t)d * t, "cm_decode: chunk stride %zu is smaller than a chunk", chunk_stride);
    hipLaunchKernelGGL(cm_decode_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, packed, (long)chunk_stride, t, d, out, (const long*)nullptr,
                       (const int*)nullptr, 8);
    XV_LAUNCH_CHECK();
    return 0;
}

extern "C" int xv_cm_decode_ragged(void* stream, const uint8_t* packed, const int64_t* offsets, const int32_t* rows, int b, int t, int d, float* out) {
    XV_REQUIRE(packed && offsets && rows && out && b > 0 && t > 0 && d > 0, "cm_decode_ragged: bad arguments");
    XV_REQUIRE(d <= CMD_MAX_D, "cm_decode_ragged: at most %d feature dimensions (got %d)", CMD_MAX_D, d);
    static_assert(sizeof(long) == sizeof(int64_t), "offsets are passed as long");
    hipLaunchKernelGGL(cm_decode_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, packed, 0L, t, d, out, (const long*)offsets, (const int*)rows, 16);
    XV_LAUNCH_CHECK();
    return 0;
}

// wt[o][j*c_pad + c] = kernel[(j*C + c)*O + o], zero for c >= C.  32x32 LDS-tiled transpose:
// reads run along o (contiguous in kernel), writes run along the padded k axis (contiguous in wt).
__global__ void prep_weight_fwd_kernel(const float* __restrict__ w, int k, int C, int O, float* __restrict__ wt, int c_pad) {
    XV_EW_FILLER();
    __shared__ float tile[32][33];
    const int kp = k * c_pad;
    const int kk0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int kk = kk0 + r, o = o0 + tx;
        float v = 0.f;
        if (kk < kp && o < O) {
            int j = kk / c_pad, c = kk - j * c_pad;
            if (c < C) v = w[((long)j * C + c) * O + o];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int o = o0 + r, kk = kk0 + tx;
        if (o < O && kk < kp) wt[(long)o * kp + kk] = tile[tx][r];
    }
}

extern "C" int xv_prep_weight_fwd(void* stream, const float* kernel, int k, int c, int o, float* wt, int c_pad) {
    XV_REQUIRE(k > 0 && c > 0 && o > 0 && c_pad >= c, "prep_weight_fwd: bad shape");
    dim3 grid(xv_cdiv((long)k * c_pad, 32), xv_cdiv(o, 32));
    hipLaunchKernelGGL(prep_weight_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, kernel, k, c, o, wt, c_pad);
    XV_LAUNCH_CHECK();
    return 0;
}

__global__ void prep_weight_dgrad_kernel(const float* __restrict__ w, int k, int C, int O, float* __restrict__ wf) {
    XV_EW_FILLER();
    long total = (long)k * C * O;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int o = (int)(i % O);
        long jc = i / O;
        int c = (int)(jc % C), j = (int)(jc / C);
        wf[(long)c * k * O + (long)(k - 1 - j) * O + o] = w[i];
    }
}

extern "C" int xv_prep_weight_dgrad(void* stream, const float* kernel, int k, int c, int o, float* wf) {
    XV_REQUIRE(k > 0 && c > 0 && o > 0, "prep_weight_dgrad: bad shape");
    long total = (long)k * c * o;
    hipLaunchKernelGGL(prep_weight_dgrad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, kernel, k, c, o, wf);
    XV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------
// multi-job weight preparation (xv_common.h): one launch for every layout copy of every layer
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weight_prep_multi_kernel(XvPrepJobs J) {
    XV_EW_FILLER();
    __shared__ float tile[32][33];
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < J.n; ++i) if ((int)blockIdx.x >= J.j[i].tile0) ji = i;
    const XvPrepJob& q = J.j[ji];
    const int lt = blockIdx.x - q.tile0;
    const int txt = lt % q.tiles_x, tyt = lt / q.tiles_x;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    if (q.type == XV_PREP_PAD) {
        // rows r0 .. r0 + 31 of [O][C] -> [O][c_pad], columns c0 .. c0 + 31 (pad columns zero)
        const int r0 = tyt * 32, c = txt * 32 + tx;
        if (c < q.c_pad)
            for (int r = r0 + ty; r < min(r0 + 32, q.O); r += 8) ((float*)q.dst)[(long)r * q.c_pad + c] = c < q.C ? q.w[(long)r * q.C + c] : 0.f;
        return;
    }
    const bool planes = q.type >= XV_PREP_T16 && q.type != XV_PREP_PAD;
    const float sc = planes ? xv_pow2_scale(*q.amax) : 1.0f;
    if (q.type == XV_PREP_T32 || q.type == XV_PREP_T16) {
        // transpose through LDS: reads run along o (contiguous in w), writes along the padded k axis
        const int kp = q.k * q.c_pad;
        const int kk0 = txt * 32, o0 = tyt * 32;
        for (int r = ty; r < 32; r += 8) {
            int kk = kk0 + r, o = o0 + tx;
            float v = 0.f;
            if (kk < kp && o < q.O) {
                int j = kk / q.c_pad, c = kk - j * q.c_pad;
                if (c < q.C) v = q.w[((long)j * q.C + c) * q.O + o];
            }
            tile[r][tx] = v;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            int o = o0 + r, kk = kk0 + tx;
            if (o < q.O && kk < kp) {
                float v = tile[tx][r];
                if (!planes) ((float*)q.dst)[(long)o * kp + kk] = v;
                else {
                    float xs = v * sc;
                    _Float16 h = (_Float16)xs, l = (_Float16)(xs - (float)h);
                    unsigned short* d = (unsigned short*)q.dst;
                    d[(long)o * kp + kk] = __builtin_bit_cast(unsigned short, h);
                    d[q.plane + (long)o * kp + kk] = __builtin_bit_cast(unsigned short, l);
                }
            }
        }
    } else {
        // tap flip, no transpose: rows jc = j*C + c of w, columns o (pad columns o in [O, o_ld) are zero)
        const int jc0 = tyt * 32, o0 = txt * 32;
        const long ldd = (long)q.k * q.o_ld;
        for (int r = ty; r < 32; r += 8) {
            int jc = jc0 + r, o = o0 + tx;
            if (jc < q.k * q.C && o < q.o_ld) {
                int j = jc / q.C, c = jc - j * q.C;
                float v = o < q.O ? q.w[(long)jc * q.O + o] : 0.f;
                long di = (long)c * ldd + (long)(q.k - 1 - j) * q.o_ld + o;
                if (!planes) ((float*)q.dst)[di] = v;
                else {
                    float xs = v * sc;
                    _Float16 h = (_Float16)xs, l = (_Float16)(xs - (float)h);
                    unsigned short* d = (unsigned short*)q.dst;
                    d[di] = __builtin_bit_cast(unsigned short, h);
                    d[q.plane + di] = __builtin_bit_cast(unsigned short, l);
                }
            }
        }
    }
}

int xv_prep_add(XvPrepJobs& J, int type, const float* w, int k, int C, int O, int c_pad, int o_ld, void* dst, long plane,
                const unsigned* amax) {
    XV_REQUIRE(J.n < XV_PREP_MAX_JOBS, "weight_prep: too many jobs");
    XvPrepJob& q = J.j[J.n++];
    q.type = type; q.k = k; q.C = C; q.O = O; q.c_pad = c_pad; q.o_ld = o_ld; q.w = w; q.dst = dst; q.plane = plane; q.amax = amax;
    int tiles_y;
    if (type == XV_PREP_PAD) { q.tiles_x = xv_cdiv(c_pad, 32); tiles_y = xv_cdiv(O, 32); }
    else if (type == XV_PREP_T32 || type == XV_PREP_T16) { q.tiles_x = xv_cdiv((long)k * c_pad, 32); tiles_y = xv_cdiv(O, 32); }
    else { q.tiles_x = xv_cdiv(o_ld, 32); tiles_y = xv_cdiv((long)k * C, 32); }
    q.tile0 = J.total_tiles;
    J.total_tiles += q.tiles_x * tiles_y;
    return 0;
}

int xv_launch_weight_prep(hipStream_t s, const XvPrepJobs& J) {
    if (J.n == 0) return 0;
    hipLaunchKernelGGL(weight_prep_multi_kernel, dim3(J.total_tiles), dim3(256), 0, s, J);
    XV_LAUNCH_CHECK();
    return 0;
}

// max |x| of up to 8 tensors in one launch: blockIdx.y = tensor, one atomicMax per workgroup (slots zeroed by the caller)
__global__ __launch_bounds__(256) void amax_multi_kernel(XvAmaxJobs J) {
    XV_EW_FILLER();
    __shared__ float red[4];
    const float* x = J.x[blockIdx.y];
    const size_t count = J.count[blockIdx.y];
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        unsigned bits = __float_as_uint(m);
        unsigned* out = J.out[blockIdx.y];
        if (bits > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, bits);
    }
}

int xv_launch_amax_multi(hipStream_t s, const XvAmaxJobs& J) {
    if (J.n == 0) return 0;
    hipLaunchKernelGGL(amax_multi_kernel, dim3(256, J.n), dim3(256), 0, s, J);
    XV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------
// column sums / column statistics (row-chunk partials, then a fixed-order combine)
// ------------------------------------------------------------------------------------
#define CS_ROWS 128
// block = 256 threads = 64 columns x 4 row lanes; row lanes are combined in a fixed order
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ a, int rows, int n, long lda,
                                                             float* __restrict__ part) {
    XV_EW_PRIORITY();
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
    float s = 0.f;
    if (col < n) {
        int r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            float v0 = a[(long)r * lda + col], v1 = a[(long)(r + 4) * lda + col];
            float v2 = a[(long)(r + 8) * lda + col], v3 = a[(long)(r + 12) * lda + col];
            s += (v0 + v1) + (v2 + v3);
        }
        for (; r < r1; r += 4) s += a[(long)r * lda + col];
    }
    red[rl][cx] = s;
    __syncthreads();
    if (rl == 0 && col < n) part[(long)blockIdx.y * n + col] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}
// block = 256 threads = 32 columns x 8 chunk lanes
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int chunks, int n, float* __restrict__ out) {
    XV_EW_PRIORITY();
    __shared__ float red[8][32];
    const int cx = threadIdx.x & 31, cl = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cx;
    float s = 0.f;
    if (col < n)
        for (int c = cl; c < chunks; c += 8) s += part[(long)c * n + col];
    red[cl][cx] = s;
    __syncthreads();
    if (cl == 0 && col < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][cx];
        out[col] = t;
    }
}

extern "C" int xv_colsum(void* stream, const float* a, int rows, int n, int lda, float* out, void* ws, size_t ws_bytes) {
    XV_REQUIRE(rows > 0 && n > 0 && lda >= n, "colsum: bad shape");
    int chunks = xv_cdiv(rows, CS_ROWS);
    XV_REQUIRE((size_t)chunks * n * sizeof(float) <= ws_bytes, "colsum: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(xv_cdiv(n, 64), chunks), dim3(256), 0, s, a, rows, n, (long)lda, (float*)ws);
    XV_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3(xv_cdiv(n, 32)), dim3(256), 0, s, (const float*)ws, chunks, n, out);
    XV_LAUNCH_CHECK();
    return 0;
}

// bn_part layout: [2][tiles][n] with tiles = ceil(rows / XV_TILE_M): sum, then centred sum of squares.
// block = 256 threads = 32 columns x 8 row lanes over one 128-row tile; two passes (sum/min/max, then
// squares centred on the tile mean) with fixed-order combines through LDS.  Output layout: xv_epilogue.h.
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ z, int rows, int n, long ldz,
                                                        float* __restrict__ part, int tiles) {
    XV_EW_PRIORITY();
    __shared__ float red[8][32], rmin[8][32], rmax[8][32];
    __shared__ float s_mean[32];
    const int cx = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cx;
    const int tile = blockIdx.y;
    const int r0 = tile * XV_TILE_M, r1 = min(rows, r0 + XV_TILE_M);
    const long plane = (long)tiles * n;
    float s = 0.f, mn = INFINITY, mx = -INFINITY;
    if (col < n)
        for (int r = r0 + rl; r < r1; r += 8) {
            float v = z[(long)r * ldz + col];
            s += v; mn = fminf(mn, v); mx = fmaxf(mx, v);
        }
    red[rl][cx] = s; rmin[rl][cx] = mn; rmax[rl][cx] = mx;
    __syncthreads();
    if (rl == 0) {
        float t = 0.f, a = INFINITY, b = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) { t += red[k][cx]; a = fminf(a, rmin[k][cx]); b = fmaxf(b, rmax[k][cx]); }
        if (col < n) {
            part[(long)tile * n + col] = t;
            part[2 * plane + (long)tile * n + col] = a;
            part[3 * plane + (long)tile * n + col] = b;
        }
        s_mean[cx] = t / (float)(r1 - r0);
    }
    __syncthreads();
    const float mean = s_mean[cx];
    float q = 0.f;
    if (col < n)
        for (int r = r0 + rl; r < r1; r += 8) {
            float d = z[(long)r * ldz + col] - mean;
            q += d * d;
        }
    __syncthreads();
    red[rl][cx] = q;
    __syncthreads();
    if (rl == 0 && col < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][cx];
        part[plane + (long)tile * n + col] = t;
    }
}

// The same statistics for rows of whole float4s (n and ldz multiples of 4, 16-byte aligned): ONE pass over memory.  block = 256 threads =
// 32 column quads x 8 row lanes over one 128-row tile; a thread keeps its 16 rows x 4 columns in registers, so the squares centred on the
// tile mean come from registers instead of a second read, and a wave instruction covers two rows of 512 contiguous bytes (the scalar
// form above: 128 bytes per row, every element read twice - 0.24-0.30 of the HBM rate in round 2).  Same fixed-order combines.
__global__ __launch_bounds__(256) void col_stats4_kernel(const float* __restrict__ z, int rows, int n, long ldz, float* __restrict__ part,
                                                         int tiles) {
    XV_EW_PRIORITY();
    __shared__ f32x4 red[8][32], rmin[8][32], rmax[8][32];
    __shared__ f32x4 s_mean[32];
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = (blockIdx.x * 32 + cq) * 4;
    const int tile = blockIdx.y;
    const int r0 = tile * XV_TILE_M, r1 = min(rows, r0 + XV_TILE_M);
    const long plane = (long)tiles * n;
    const bool cv = col < n;
    f32x4 v[XV_TILE_M / 8];
    f32x4 s = {0, 0, 0, 0}, mn = {INFINITY, INFINITY, INFINITY, INFINITY}, mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    // unconditional loads (row / column clamped), all sixteen in flight: a predicated load compiles to a branch + s_waitcnt vmcnt(0)
    const float* __restrict__ zc = z + (cv ? col : 0);
#pragma unroll
    for (int i = 0; i < XV_TILE_M / 8; ++i) v[i] = *(const f32x4*)(zc + (long)min(r0 + rl + 8 * i, r1 - 1) * ldz);
#pragma unroll
    for (int i = 0; i < XV_TILE_M / 8; ++i) {
        const bool ok = cv && r0 + rl + 8 * i < r1;
        const f32x4 x = v[i];
        s += ok ? x : f32x4{0, 0, 0, 0};
        mn.x = ok ? fminf(mn.x, x.x) : mn.x; mn.y = ok ? fminf(mn.y, x.y) : mn.y; mn.z = ok ? fminf(mn.z, x.z) : mn.z; mn.w = ok ? fminf(mn.w, x.w) : mn.w;
        mx.x = ok ? fmaxf(mx.x, x.x) : mx.x; mx.y = ok ? fmaxf(mx.y, x.y) : mx.y; mx.z = ok ? fmaxf(mx.z, x.z) : mx.z; mx.w = ok ? fmaxf(mx.w, x.w) : mx.w;
    }
    red[rl][cq] = s; rmin[rl][cq] = mn; rmax[rl][cq] = mx;
    __syncthreads();
    if (rl == 0) {
        f32x4 t = {0, 0, 0, 0}, a = rmin[0][cq], b = rmax[0][cq];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            t += red[k][cq];
            const f32x4 u = rmin[k][cq], w = rmax[k][cq];
            a.x = fminf(a.x, u.x); a.y = fminf(a.y, u.y); a.z = fminf(a.z, u.z); a.w = fminf(a.w, u.w);
            b.x = fmaxf(b.x, w.x); b.y = fmaxf(b.y, w.y); b.z = fmaxf(b.z, w.z); b.w = fmaxf(b.w, w.w);
        }
        if (cv) {
            *(f32x4*)(part + (long)tile * n + col) = t;
            *(f32x4*)(part + 2 * plane + (long)tile * n + col) = a;
            *(f32x4*)(part + 3 * plane + (long)tile * n + col) = b;
        }
        s_mean[cq] = t / (float)(r1 - r0);
    }
    __syncthreads();
    const f32x4 mean = s_mean[cq];
    f32x4 q = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < XV_TILE_M / 8; ++i)
        if (cv && r0 + rl + 8 * i < r1) {
            const f32x4 d = v[i] - mean;
            q += d * d;
        }
    __syncthreads();
    red[rl][cq] = q;
    __syncthreads();
    if (rl == 0 && cv) {
        f32x4 t = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][cq];
        *(f32x4*)(part + plane + (long)tile * n + col) = t;
    }
}

extern "C" int xv_col_stats(void* stream, const float* z, int rows, int n, int ldz, float* bn_part) {
    XV_REQUIRE(rows > 0 && n > 0 && ldz >= n, "col_stats: bad shape");
    int tiles = xv_cdiv(rows, XV_TILE_M);
    if (n % 4 == 0 && ldz % 4 == 0 && ((uintptr_t)z % 16) == 0 && ((uintptr_t)bn_part % 16) == 0)
        hipLaunchKernelGGL(col_stats4_kernel, dim3(xv_cdiv(n, 128), tiles), dim3(256), 0, (hipStream_t)stream, z, rows, n, (long)ldz, bn_part, tiles);
    else
        hipLaunchKernelGGL(col_stats_kernel, dim3(xv_cdiv(n, 32), tiles), dim3(256), 0, (hipStream_t)stream, z, rows, n, (long)ldz, bn_part, tiles);
    XV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------
// Activation context (network_relu_type, tdnn.py:24-30 / common.py:27-42): the non-linearity behind a BatchNorm is
//   act(y) = y > 0 ? y : slope[c] * y     slope = NULL: ReLU | a constant 0.2 vector: tf.nn.leaky_relu | the layer's alpha: prelu
// (prelu(x) = relu(x) + alpha (x - |x|) / 2 is exactly that).  The engine sets the context around a layer's calls; every entry
// point with a `relu` flag reads it, so the C signatures stay as they are.  dalpha: where the backward entry points write
// d alpha[c] = sum_rows d act * min(y, 0) (prelu only).
// ------------------------------------------------------------------------------------
static thread_local XvActContext g_act = {nullptr, nullptr};
void xv_set_act_context(const float* slope, float* dalpha) { g_act.slope = slope; g_act.dalpha = dalpha; }
XvActContext xv_act_context() { return g_act; }
extern "C" int xv_set_activation(const float* slope, float* dalpha) {
    XV_REQUIRE(slope || !dalpha, "set_activation: a d alpha buffer needs a slope vector");
    xv_set_act_context(slope, dalpha);
    return 0;
}

__device__ __forceinline__ float act1(float y, float sl) { return y > 0.f ? y : y * sl; }
__device__ __forceinline__ f32x4 act4(f32x4 y, f32x4 sl) {
    f32x4 r;
    r.x = act1(y.x, sl.x); r.y = act1(y.y, sl.y); r.z = act1(y.z, sl.z); r.w = act1(y.w, sl.w);
    return r;
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
}
__device__ __forceinline__ f32x4 neg4(f32x4 y) {      // min(y, 0)
    f32x4 r;
    r.x = fminf(y.x, 0.f); r.y = fminf(y.y, 0.f); r.z = fminf(y.z, 0.f); r.w = fminf(y.w, 0.f);
    return r;
}

// ------------------------------------------------------------------------------------
// BatchNorm
// ------------------------------------------------------------------------------------
// block = 256 threads = 8 channels x 32 tile lanes (n/8 workgroups: the partials are few, the latency of a
// serial walk over them is what this kernel costs).  Each lane folds its tiles' (count, mean, M2) with Chan's
// pairwise formula in double, lanes are then folded in lane order (deterministic).
#define FIN_CH 8
#define FIN_LANES 32
#define FIN_BATCH 8
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int rows, int n, int tiles,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float momentum, int unbiased, float* __restrict__ mmean,
                                                          float* __restrict__ mvar, float* __restrict__ mean_o,
                                                          float* __restrict__ invstd_o, float* __restrict__ scale_o,
                                                          float* __restrict__ shift_o, float* __restrict__ zmin_o,
                                                          float* __restrict__ zmax_o, unsigned* __restrict__ amax_o, int relu,
                                                          const float* __restrict__ slope) {
    XV_EW_PRIORITY();
    __shared__ double s_cnt[FIN_LANES][FIN_CH], s_mean[FIN_LANES][FIN_CH], s_m2[FIN_LANES][FIN_CH];
    __shared__ float s_mn[FIN_LANES][FIN_CH], s_mx[FIN_LANES][FIN_CH];
    const int cx = threadIdx.x & (FIN_CH - 1), tl = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + cx;
    double cnt = 0.0, mean = 0.0, m2 = 0.0;
    float zmn = INFINITY, zmx = -INFINITY;
    if (c < n) {
        // the loads of FIN_BATCH tiles are issued together, then folded in tile order: the kernel is a chain of memory round trips
        // (7 per lane at S1 when every tile waited for its own four loads: 11.6 us per layer, five layers per step)
        for (int t0 = tl; t0 < tiles; t0 += FIN_LANES * FIN_BATCH) {
            float ps[FIN_BATCH], pq[FIN_BATCH], pmn[FIN_BATCH], pmx[FIN_BATCH];
#pragma unroll
            for (int u = 0; u < FIN_BATCH; ++u) {
                const int t = min(t0 + u * FIN_LANES, tiles - 1);
                ps[u] = part[(long)t * n + c];
                pq[u] = part[((long)tiles + t) * n + c];
                pmn[u] = part[(2L * tiles + t) * n + c];
                pmx[u] = part[(3L * tiles + t) * n + c];
            }
#pragma unroll
            for (int u = 0; u < FIN_BATCH; ++u) {
                const int t = t0 + u * FIN_LANES;
                if (t >= tiles) break;
                zmn = fminf(zmn, pmn[u]);
                zmx = fmaxf(zmx, pmx[u]);
                int tc = min(XV_TILE_M, rows - t * XV_TILE_M);
                double tm = (double)(ps[u] / (float)tc);   // the tile mean the producer centred on
                double tq = (double)pq[u];
                double nn = cnt + (double)tc, d = tm - mean;
                mean += d * ((double)tc / nn);
                m2 += tq + d * d * (cnt * (double)tc / nn);
                cnt = nn;
            }
        }
    }
    s_cnt[tl][cx] = cnt; s_mean[tl][cx] = mean; s_m2[tl][cx] = m2;
    s_mn[tl][cx] = zmn; s_mx[tl][cx] = zmx;
    __syncthreads();
    // fold the 32 lanes of a channel as a tree (lane l takes lane l + stride: a fixed order): the serial fold by lane 0 was 31 dependent
    // Chan merges with two fp64 divisions each - 4 of the kernel's 12 us
    for (int stride = FIN_LANES / 2; stride >= 1; stride >>= 1) {
        if (tl < stride) {
            const double cb = s_cnt[tl + stride][cx];
            if (cb > 0.0) {
                const double nn = cnt + cb, d = s_mean[tl + stride][cx] - mean;
                mean += d * (cb / nn);
                m2 += s_m2[tl + stride][cx] + d * d * (cnt * cb / nn);
                cnt = nn;
            }
            zmn = fminf(zmn, s_mn[tl + stride][cx]);
            zmx = fmaxf(zmx, s_mx[tl + stride][cx]);
            s_cnt[tl][cx] = cnt; s_mean[tl][cx] = mean; s_m2[tl][cx] = m2;
            s_mn[tl][cx] = zmn; s_mx[tl][cx] = zmx;
        }
        __syncthreads();
    }
    if (tl != 0 || c >= n) return;
    float var = (float)(m2 / (double)rows);
    float meanf = (float)mean;
    float invstd = 1.0f / sqrtf(var + eps);
    float sc = gamma[c] * invstd;
    mean_o[c] = meanf;
    invstd_o[c] = invstd;
    const float sh = beta[c] - meanf * sc;
    scale_o[c] = sc;
    shift_o[c] = sh;
    if (zmin_o) { zmin_o[c] = zmn; zmax_o[c] = zmx; }
    if (amax_o) {
        // exact range of y = z*sc + sh over the batch (affine => extremes at the ends), same fma as bn_apply
        float y0 = zmn * sc + sh, y1 = zmx * sc + sh;
        float am = relu ? fmaxf(0.f, fmaxf(y0, y1)) : fmaxf(fabsf(y0), fabsf(y1));
        if (relu && slope) am = fmaxf(fabsf(act1(y0, slope[c])), fabsf(act1(y1, slope[c])));      // piecewise linear through 0: extremes at the ends
        atomicMax(amax_o, __float_as_uint(am));          // max of non-negative floats == max of their bit patterns
    }
    if (mmean) {
        float v = (unbiased && rows > 1) ? var * ((float)rows / (float)(rows - 1)) : var;
        mmean[c] = mmean[c] * momentum + meanf * (1.0f - momentum);
        mvar[c] = mvar[c] * momentum + v * (1.0f - momentum);
    }
}

extern "C" int xv_bn_finalize(void* stream, const float* bn_part, int rows, int n, const float* gamma, const float* beta,
                              float eps, float momentum, int unbiased_moving, float* moving_mean, float* moving_var,
                              float* mean, float* invstd, float* scale, float* shift, float* zmin, float* zmax,
                              uint32_t* amax, int relu) {
    XV_REQUIRE(rows > 0 && n > 0, "bn_finalize: bad shape");
    int tiles = xv_cdiv(rows, XV_TILE_M);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(xv_cdiv(n, FIN_CH)), dim3(256), 0, (hipStream_t)stream, bn_part, rows, n, tiles,
                       gamma, beta, eps, momentum, unbiased_moving, moving_mean, moving_var, mean, invstd, scale, shift, zmin, zmax, amax, relu,
                       g_act.slope);
    XV_LAUNCH_CHECK();
    return 0;
}

// Range of z per channel from the GEMM epilogue's min/max partials and the exact output range of
// relu?(z*scale+shift) for given (e.g. inference) scale/shift: *amax |= its float bits.
__global__ void bn_output_range_kernel(const float* __restrict__ part, int rows, int n, int tiles, const float* __restrict__ scale,
                                       const float* __restrict__ shift, int relu, float* __restrict__ zmin_o,
                                       float* __restrict__ zmax_o, unsigned* __restrict__ amax_o, const float* __restrict__ slope) {
    XV_EW_PRIORITY();
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    float mn = INFINITY, mx = -INFINITY;
    for (int t = 0; t < tiles; ++t) {
        mn = fminf(mn, part[(2L * tiles + t) * n + c]);
        mx = fmaxf(mx, part[(3L * tiles + t) * n + c]);
    }
    if (zmin_o) { zmin_o[c] = mn; zmax_o[c] = mx; }
    float y0 = mn * scale[c] + shift[c], y1 = mx * scale[c] + shift[c];
    float am = relu ? fmaxf(0.f, fmaxf(y0, y1)) : fmaxf(fabsf(y0), fabsf(y1));
    if (relu && slope) am = fmaxf(fabsf(act1(y0, slope[c])), fabsf(act1(y1, slope[c])));
    atomicMax(amax_o, __float_as_uint(am));
}

extern "C" int xv_bn_output_range(void* stream, const float* bn_part, int rows, int n, const float* scale, const float* shift, int relu,
                                  float* zmin, float* zmax, uint32_t* amax) {
    XV_REQUIRE(bn_part && rows > 0 && n > 0 && scale && shift && amax, "bn_output_range: bad arguments");
    int tiles = xv_cdiv(rows, XV_TILE_M);
    hipLaunchKernelGGL(bn_output_range_kernel, dim3(xv_cdiv(n, 128)), dim3(128), 0, (hipStream_t)stream, bn_part, rows, n, tiles, scale, shift,
                       relu, zmin, zmax, (unsigned*)amax, g_act.slope);
    XV_LAUNCH_CHECK();
    return 0;
}

__global__ void bn_inference_scale_kernel(int n, const float* __restrict__ gamma, const float* __restrict__ beta,
                                          const float* __restrict__ mmean, const float* __restrict__ mvar, float eps,
                                          float* __restrict__ scale, float* __restrict__ shift) {
    XV_EW_PRIORITY();
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    float sc = gamma[c] * (1.0f / sqrtf(mvar[c] + eps));
    scale[c] = sc;
    shift[c] = beta[c] - mmean[c] * sc;
}

extern "C" int xv_bn_inference_scale(void* stream, int n, const float* gamma, const float* beta, const float* moving_mean,
                                     const float* moving_var, float eps, float* scale, float* shift) {
    XV_REQUIRE(n > 0, "bn_inference_scale: bad shape");
    hipLaunchKernelGGL(bn_inference_scale_kernel, dim3(xv_cdiv(n, 128)), dim3(128), 0, (hipStream_t)stream, n, gamma, beta,
                       moving_mean, moving_var, eps, scale, shift);
    XV_LAUNCH_CHECK();
    return 0;
}

// a = relu?(z*scale+shift).  Thread = one channel quad (16 B) x a strip of rows, block = 64 quads x 4 row lanes over BA_ROWS rows: scale,
// shift and slope are loaded once per thread and the strip's eight loads are in flight together (the element-per-thread grid-stride form
// divided by the row length and reloaded the three vectors for every 16 bytes: 34 us for 97.5 MB alone, r02_elementwise.json).
#define BA_ROWS 32
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, int rows, int nq, long ldz, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int relu, float* __restrict__ a, long lda,
                                                       const float* __restrict__ slope) {
    XV_EW_PRIORITY();
    const int q = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    if (q >= nq) return;
    const f32x4 sc = *(const f32x4*)(scale + 4 * q), sh = *(const f32x4*)(shift + 4 * q);
    f32x4 sl = {0, 0, 0, 0};
    if (slope) sl = *(const f32x4*)(slope + 4 * q);
    const int r0 = blockIdx.x * BA_ROWS + rl;
    f32x4 v[BA_ROWS / 4];
#pragma unroll
    for (int j = 0; j < BA_ROWS / 4; ++j) v[j] = *(const f32x4*)(z + (long)min(r0 + 4 * j, rows - 1) * ldz + 4 * q);
#pragma unroll
    for (int j = 0; j < BA_ROWS / 4; ++j) {
        f32x4 y = v[j] * sc + sh;
        if (relu) y = slope ? act4(y, sl) : relu4(y);
        if (r0 + 4 * j < rows) *(f32x4*)(a + (long)(r0 + 4 * j) * lda + 4 * q) = y;
    }
}

extern "C" int xv_bn_apply(void* stream, const float* z, int rows, int n, int ldz, const float* scale, const float* shift,
                           int relu, float* a, int lda) {
    XV_REQUIRE(rows > 0 && n > 0 && n % 4 == 0 && ldz % 4 == 0 && lda % 4 == 0, "bn_apply: n/ld must be multiples of 4 (n=%d)", n);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(xv_cdiv(rows, BA_ROWS), xv_cdiv(n / 4, 64)), dim3(256), 0, (hipStream_t)stream, z, rows, n / 4,
                       (long)ldz, scale, shift, relu, a, (long)lda, g_act.slope);
    XV_LAUNCH_CHECK();
    return 0;
}

// Upstream gradient of a layer whose output feeds statistics pooling directly (tdnn5): instead of reading a
// materialised d(activation), the BN backward evaluates the pooling backward (pooling.py:9-34) on the fly from the
// pooled statistics [b][mean | std] and their gradient:  da = dmean/T + dstd/(T*std) * (a - mean),  a = relu?(z*scale+shift).
struct PoolGrad { const float* out; const float* dout; int t; const float* w; const float* wpos; const float* amax; };   // w: per-frame attention weights or null (1/t)
// wpos [b][n] (optional): the share of each chunk's frame weights that sits on frames with an active ReLU, written by the pooling forward -
// with it the BatchNorm backward's two reductions have a closed form per (chunk, channel) and the pass over z is not needed (bn_bwd_pooled_stats_kernel).
// amax [b][n] (optional): each chunk's largest activation - bounds |d a| for the split-precision dz scale

// The pooled statistics of chunk b at one channel quad, in the form the per-element formula needs: they change only when
// the row loop crosses into the next chunk, so the kernels reload them there instead of once per element (4 vector loads,
// 4 divisions and an integer division by T per 16 bytes of z before).
struct PoolCoef { f32x4 mean, dm, q; };      // q = dstd / std (0 where the forward clamped the variance)
__device__ __forceinline__ PoolCoef pool_coef(const PoolGrad& pg, int b, int n, int col) {
    const float sd_eps = 1e-6f;      // sqrt(1e-12): the forward clamps the variance there (pooling.py:28-29)
    const float* o = pg.out + (long)b * 2 * n;
    const float* g = pg.dout + (long)b * 2 * n;
    PoolCoef pc;
    pc.mean = *(const f32x4*)(o + col);
    pc.dm = *(const f32x4*)(g + col);
    const f32x4 sd = *(const f32x4*)(o + n + col), ds = *(const f32x4*)(g + n + col);
    pc.q.x = sd.x <= sd_eps ? 0.f : ds.x / sd.x; pc.q.y = sd.y <= sd_eps ? 0.f : ds.y / sd.y;
    pc.q.z = sd.z <= sd_eps ? 0.f : ds.z / sd.z; pc.q.w = sd.w <= sd_eps ? 0.f : ds.w / sd.w;
    return pc;
}
// da = dmean * w + (dstd / std * w) * (a - mean), w = 1/T or the frame's attention weight (pooling.py:148-155)
__device__ __forceinline__ f32x4 pool_grad(const PoolCoef& pc, float invT, f32x4 a) {
    return pc.dm * invT + (pc.q * invT) * (a - pc.mean);
}
__device__ __forceinline__ float pool_frame_weight(const PoolGrad& pg, long row) { return pg.w ? pg.w[row] : 1.f / (float)pg.t; }

// masked upstream gradient of one channel quad: POOLED ? pooling backward on the fly : da, zeroed where the ReLU was off
// sl / hs: the activation's negative-side slope of this channel quad and whether there is one (act context); dneg, if given,
// receives d act * min(y, 0) - the summand of d alpha (prelu).
template <bool POOLED>
__device__ __forceinline__ f32x4 upstream_grad(const float* __restrict__ da, const PoolCoef& pc, float invT, long r, int n, int col,
                                               f32x4 zz, f32x4 sc, f32x4 sh, int relu, f32x4 sl = f32x4{0, 0, 0, 0}, bool hs = false,
                                               f32x4* dneg = nullptr) {
    f32x4 y = zz * sc + sh;
    f32x4 dd;
    if (POOLED) {
        f32x4 a = y;
        if (relu) a = hs ? act4(a, sl) : relu4(a);
        dd = pool_grad(pc, invT, a);
    } else {
        dd = *(const f32x4*)(da + r * n + col);
    }
    if (relu) {
        if (dneg) *dneg = dd * neg4(y);
        if (hs) {
            dd.x = y.x > 0.f ? dd.x : dd.x * sl.x; dd.y = y.y > 0.f ? dd.y : dd.y * sl.y;
            dd.z = y.z > 0.f ? dd.z : dd.z * sl.z; dd.w = y.w > 0.f ? dd.w : dd.w * sl.w;
        } else {
            dd.x = y.x > 0.f ? dd.x : 0.f; dd.y = y.y > 0.f ? dd.y : 0.f;
            dd.z = y.z > 0.f ? dd.z : 0.f; dd.w = y.w > 0.f ? dd.w : 0.f;
        }
    }
    return dd;
}

// Backward pass 1: per (64-row chunk, 256-column block) partial sums of dy and dy*xhat (upstream gradient d a in memory; the pooled
// form is bn_bwd_reduce_pooled_kernel below).
// block = 256 threads = 64 column-quads x 4 row lanes.
#define BB_ROWS 64
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ da, const float* __restrict__ z, int rows,
                                                            int n, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu,
                                                            float* __restrict__ part /* [chunks][nstat][n]: sum dy, sum dy*xhat, max |dy| (, sum d act*min(y,0)) */,
                                                            const float* __restrict__ slope, int nstat) {
    XV_EW_PRIORITY();
    __shared__ f32x4 red[4][4][64];
    const int qx = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + qx) * 4;
    const int r0 = blockIdx.y * BB_ROWS, r1 = min(rows, r0 + BB_ROWS);
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0}, s3 = {0, 0, 0, 0}, s4 = {0, 0, 0, 0};
    const bool hs = slope != nullptr, want4 = nstat == 4;
    if (col < n) {
        f32x4 mu = *(const f32x4*)(mean + col), is = *(const f32x4*)(invstd + col);
        f32x4 sc = *(const f32x4*)(scale + col), sh = *(const f32x4*)(shift + col);
        f32x4 sl = {0, 0, 0, 0};
        if (hs) sl = *(const f32x4*)(slope + col);
        const PoolCoef pc = {};
        // one row per trip: [measured] four rows' loads issued together make this form slower inside the step (42 -> 46 us; its 50 MB
        // tensors sit in the Infinity Cache)
        for (int r = r0 + rl; r < r1; r += 4) {
            const f32x4 zz = *(const f32x4*)(z + (long)r * n + col);
            // (the prelu term d act * min(y, 0) is formed here, from the unmasked gradient: handing upstream_grad a pointer for it put
            // the value in scratch memory - 32 bytes per lane, written and re-read once per row - in the kernel the data-gradient chain waits for)
            const f32x4 raw = *(const f32x4*)(da + (long)r * n + col);
            if (want4 && relu) s4 += raw * neg4(zz * sc + sh);
            f32x4 dd = upstream_grad<false>(da, pc, 0.f, (long)r, n, col, zz, sc, sh, relu, sl, hs);
            f32x4 xh = (zz - mu) * is;
            s1 += dd;
            s2 += dd * xh;
            s3.x = fmaxf(s3.x, fabsf(dd.x)); s3.y = fmaxf(s3.y, fabsf(dd.y));
            s3.z = fmaxf(s3.z, fabsf(dd.z)); s3.w = fmaxf(s3.w, fabsf(dd.w));
        }
    }
    red[0][rl][qx] = s1;
    red[1][rl][qx] = s2;
    red[2][rl][qx] = s3;
    red[3][rl][qx] = s4;
    __syncthreads();
    if (rl == 0 && col < n) {
        f32x4 t1 = (red[0][0][qx] + red[0][1][qx]) + (red[0][2][qx] + red[0][3][qx]);
        f32x4 t2 = (red[1][0][qx] + red[1][1][qx]) + (red[1][2][qx] + red[1][3][qx]);
        f32x4 t3;
        t3.x = fmaxf(fmaxf(red[2][0][qx].x, red[2][1][qx].x), fmaxf(red[2][2][qx].x, red[2][3][qx].x));
        t3.y = fmaxf(fmaxf(red[2][0][qx].y, red[2][1][qx].y), fmaxf(red[2][2][qx].y, red[2][3][qx].y));
        t3.z = fmaxf(fmaxf(red[2][0][qx].z, red[2][1][qx].z), fmaxf(red[2][2][qx].z, red[2][3][qx].z));
        t3.w = fmaxf(fmaxf(red[2][0][qx].w, red[2][1][qx].w), fmaxf(red[2][2][qx].w, red[2][3][qx].w));
        *(f32x4*)(part + ((long)blockIdx.y * nstat + 0) * n + col) = t1;
        *(f32x4*)(part + ((long)blockIdx.y * nstat + 1) * n + col) = t2;
        *(f32x4*)(part + ((long)blockIdx.y * nstat + 2) * n + col) = t3;
        if (want4) *(f32x4*)(part + ((long)blockIdx.y * nstat + 3) * n + col) = (red[3][0][qx] + red[3][1][qx]) + (red[3][2][qx] + red[3][3][qx]);
    }
}

// The POOLED reductions as a kernel of their own (prelu / lrelu / attention pooling: the cases without a closed form).  A workgroup stays
// inside ONE chunk of the batch (grid.y = chunk x row block), so the pooled statistics, their four divisions and 1/T are loaded once per
// thread, the row loop has no chunk-crossing branch and keeps eight 16-byte loads of z in flight per lane; rows beyond the block carry a
// frame weight of zero (every summand of theirs is 0) instead of a predicate.  [measured, round 2, 143 MB of z at S1] the generic kernel above
// needed 52.7 us (86 branches, a vmcnt(0) per row) where the pooling forward reads the same bytes in 27.5 us.
#define BBP_ROWS 64
#define BBP_FLIGHT 8
template <bool RELU, bool HS, bool ATT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pooled_kernel(PoolGrad pg, const float* __restrict__ z, int n, int nsub, int rows_per,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   float* __restrict__ part /* [chunks * nsub][nstat][n] */,
                                                                   const float* __restrict__ slope, int nstat) {
    XV_EW_PRIORITY();
    __shared__ f32x4 red[4][4][64];
    const int qx = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + qx) * 4;
    const int b = blockIdx.y / nsub, jb = blockIdx.y - b * nsub;
    const int r0 = b * pg.t + jb * rows_per, r1 = min((b + 1) * pg.t, r0 + rows_per);
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0}, s3 = {0, 0, 0, 0}, s4 = {0, 0, 0, 0};
    if (col < n && r0 < r1) {
        const f32x4 mu = *(const f32x4*)(mean + col), is = *(const f32x4*)(invstd + col);
        const f32x4 sc = *(const f32x4*)(scale + col), sh = *(const f32x4*)(shift + col);
        f32x4 sl = {0, 0, 0, 0};
        if (HS) sl = *(const f32x4*)(slope + col);
        const PoolCoef pc = pool_coef(pg, b, n, col);
        const float w_uniform = 1.f / (float)pg.t;
        const float* __restrict__ zc = z + col;
        for (int rb = r0 + rl; rb < r1; rb += 4 * BBP_FLIGHT) {
            f32x4 zq[BBP_FLIGHT];
            float wq[BBP_FLIGHT];
#pragma unroll
            for (int j = 0; j < BBP_FLIGHT; ++j) {
                const int r = min(rb + 4 * j, r1 - 1);
                zq[j] = *(const f32x4*)(zc + (long)r * n);
                wq[j] = ATT ? pg.w[r] : w_uniform;
            }
#pragma unroll
            for (int j = 0; j < BBP_FLIGHT; ++j) {
                const float w = rb + 4 * j < r1 ? wq[j] : 0.f;
                const f32x4 y = zq[j] * sc + sh;
                f32x4 a = y;
                if (RELU) a = HS ? act4(a, sl) : relu4(a);
                f32x4 dd = pool_grad(pc, w, a);
                if (RELU) {
                    if (HS) {
                        s4 += dd * neg4(y);
                        dd.x = y.x > 0.f ? dd.x : dd.x * sl.x; dd.y = y.y > 0.f ? dd.y : dd.y * sl.y;
                        dd.z = y.z > 0.f ? dd.z : dd.z * sl.z; dd.w = y.w > 0.f ? dd.w : dd.w * sl.w;
                    } else {
                        dd.x = y.x > 0.f ? dd.x : 0.f; dd.y = y.y > 0.f ? dd.y : 0.f;
                        dd.z = y.z > 0.f ? dd.z : 0.f; dd.w = y.w > 0.f ? dd.w : 0.f;
                    }
                }
                const f32x4 xh = (zq[j] - mu) * is;
                s1 += dd;
                s2 += dd * xh;
                s3.x = fmaxf(s3.x, fabsf(dd.x)); s3.y = fmaxf(s3.y, fabsf(dd.y));
                s3.z = fmaxf(s3.z, fabsf(dd.z)); s3.w = fmaxf(s3.w, fabsf(dd.w));
            }
        }
    }
    red[0][rl][qx] = s1;
    red[1][rl][qx] = s2;
    red[2][rl][qx] = s3;
    red[3][rl][qx] = s4;
    __syncthreads();
    if (rl == 0 && col < n) {
        f32x4 t3;
        t3.x = fmaxf(fmaxf(red[2][0][qx].x, red[2][1][qx].x), fmaxf(red[2][2][qx].x, red[2][3][qx].x));
        t3.y = fmaxf(fmaxf(red[2][0][qx].y, red[2][1][qx].y), fmaxf(red[2][2][qx].y, red[2][3][qx].y));
        t3.z = fmaxf(fmaxf(red[2][0][qx].z, red[2][1][qx].z), fmaxf(red[2][2][qx].z, red[2][3][qx].z));
        t3.w = fmaxf(fmaxf(red[2][0][qx].w, red[2][1][qx].w), fmaxf(red[2][2][qx].w, red[2][3][qx].w));
        float* o = part + (long)blockIdx.y * nstat * n + col;
        *(f32x4*)(o) = (red[0][0][qx] + red[0][1][qx]) + (red[0][2][qx] + red[0][3][qx]);
        *(f32x4*)(o + n) = (red[1][0][qx] + red[1][1][qx]) + (red[1][2][qx] + red[1][3][qx]);
        *(f32x4*)(o + 2 * n) = t3;
        if (nstat == 4) *(f32x4*)(o + 3 * n) = (red[3][0][qx] + red[3][1][qx]) + (red[3][2][qx] + red[3][3][qx]);
    }
}

// the POOLED reductions of (z, pooled statistics) into part [*chunks][nstat][n]; *chunks = the number of partials written
static void launch_bn_bwd_reduce_pooled(hipStream_t s, const PoolGrad& pg, const float* z, int rows, int n, const float* mean, const float* invstd,
                                        const float* scale, const float* shift, int relu, float* part, const float* slope, int nstat, int* chunks) {
    const int nb = rows / pg.t, nsub = xv_cdiv(pg.t, BBP_ROWS), rows_per = xv_cdiv(pg.t, nsub);
    *chunks = nb * nsub;
    const dim3 grid(xv_cdiv(n / 4, 64), nb * nsub), block(256);
#define XV_BBP(R, H, A) hipLaunchKernelGGL((bn_bwd_reduce_pooled_kernel<R, H, A>), grid, block, 0, s, pg, z, n, nsub, rows_per, mean, invstd, scale, shift, part, slope, nstat)
    const bool hs = relu && slope, att = pg.w != nullptr;
    if (!relu) { if (att) XV_BBP(false, false, true); else XV_BBP(false, false, false); }
    else if (hs) { if (att) XV_BBP(true, true, true); else XV_BBP(true, true, false); }
    else { if (att) XV_BBP(true, false, true); else XV_BBP(true, false, false); }
#undef XV_BBP
}
static int bn_bwd_pooled_chunks(int rows, int t) { return (rows / t) * xv_cdiv(t, BBP_ROWS); }

// block = 256 threads = 8 channels x 32 chunk lanes, fixed-order combine
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int chunks, int n, int rows,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef /* [2][n] */,
                                                              const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                              float* __restrict__ dbias, const float* __restrict__ mean,
                                                              const float* __restrict__ zmin, const float* __restrict__ zmax,
                                                              unsigned* __restrict__ dz_amax, int nstat, float* __restrict__ dalpha) {
    XV_EW_PRIORITY();
    __shared__ float r1[FIN_LANES][FIN_CH], r2[FIN_LANES][FIN_CH], r3[FIN_LANES][FIN_CH], r4[FIN_LANES][FIN_CH];
    const int cx = threadIdx.x & (FIN_CH - 1), cl = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + cx;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    if (c < n)
        for (int k0 = cl; k0 < chunks; k0 += FIN_LANES * FIN_BATCH) {      // FIN_BATCH chunks' loads in flight, summed in chunk order
            float p1[FIN_BATCH], p2[FIN_BATCH], p3[FIN_BATCH], p4[FIN_BATCH];
#pragma unroll
            for (int u = 0; u < FIN_BATCH; ++u) {
                const long k = min(k0 + u * FIN_LANES, chunks - 1);
                p1[u] = part[(k * nstat + 0) * n + c];
                p2[u] = part[(k * nstat + 1) * n + c];
                p3[u] = part[(k * nstat + 2) * n + c];
                p4[u] = nstat == 4 ? part[(k * nstat + 3) * n + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < FIN_BATCH; ++u) {
                if (k0 + u * FIN_LANES >= chunks) break;
                s1 += p1[u];
                s2 += p2[u];
                s3 = fmaxf(s3, p3[u]);
                s4 += p4[u];
            }
        }
    r1[cl][cx] = s1; r2[cl][cx] = s2; r3[cl][cx] = s3; r4[cl][cx] = s4;
    __syncthreads();
    if (cl != 0 || c >= n) return;
    s1 = 0.f; s2 = 0.f; s3 = 0.f; s4 = 0.f;
#pragma unroll
    for (int k = 0; k < FIN_LANES; ++k) { s1 += r1[k][cx]; s2 += r2[k][cx]; s3 = fmaxf(s3, r3[k][cx]); s4 += r4[k][cx]; }
    dbeta[c] = s1;
    dgamma[c] = s2;
    if (dalpha) dalpha[c] = s4;
    const float c1 = s1 / (float)rows;
    coef[c] = c1;
    coef[n + c] = s2 / (float)rows;
    if (dbias) dbias[c] = gamma[c] * invstd[c] * (s1 - c1 * (float)rows);   // == sum(dz) up to rounding: 0 + noise
    if (dz_amax) {
        // upper bound of |dz| = |gamma*invstd| * |dy - c1 - xhat*c2| over the batch (split-precision operand scale)
        float xh = fmaxf(fabsf(zmax[c] - mean[c]), fabsf(zmin[c] - mean[c])) * invstd[c];
        float bound = fabsf(gamma[c] * invstd[c]) * (s3 + fabsf(c1) + xh * fabsf(s2 / (float)rows)) * 1.0001f;
        atomicMax(dz_amax, __float_as_uint(bound));
    }
}

// Backward pass 2: dz = gamma*invstd*(dy - c1 - xhat*c2) into the segment-padded layout (fp32).
// Thread = one channel quad (16 B) x a strip of rows, block = 64 quads x 4 row lanes over BAF_ROWS padded rows: the seven
// per-channel parameter vectors are loaded once per thread and the pooled statistics once per chunk (the element-per-thread
// form reloaded both - and divided - for every 16 bytes of z: 115 us for tdnn5's 143 MB at S1).
#define BAF_ROWS 32
template <bool POOLED>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ da, PoolGrad pg, const float* __restrict__ z, int segs,
                                                           int t, int n, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ coef, int relu, int pad,
                                                           float* __restrict__ dz, const float* __restrict__ slope, int ldz /* rows of z and dz */) {
    XV_EW_PRIORITY();
    const int tp = t + 2 * pad;
    const int col = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int rl = threadIdx.x >> 6;
    if (col >= n) return;
    const int total_rows = segs * tp;
    const int r0 = blockIdx.y * BAF_ROWS, r1 = min(total_rows, r0 + BAF_ROWS);
    const f32x4 mu = *(const f32x4*)(mean + col), is = *(const f32x4*)(invstd + col);
    const f32x4 sc = *(const f32x4*)(scale + col), sh = *(const f32x4*)(shift + col);
    const f32x4 c1 = *(const f32x4*)(coef + col), c2 = *(const f32x4*)(coef + n + col);
    const f32x4 g_is = *(const f32x4*)(gamma + col) * is;
    const bool hs = slope != nullptr;
    f32x4 sl = {0, 0, 0, 0};
    if (hs) sl = *(const f32x4*)(slope + col);
    int seg = r0 / tp, u = r0 - seg * tp;          // padded row r0 -> (segment, frame + pad)
    u += rl;
    while (u >= tp) { u -= tp; ++seg; }
    PoolCoef pc = {};
    int b_end = 0;                                 // first row beyond the chunk whose statistics are in pc
    float invT = 0.f;
    for (int dr = r0 + rl; dr < r1; dr += 4) {
        f32x4 out = {0, 0, 0, 0};
        const int f = u - pad;
        if (f >= 0 && f < t) {
            const long r = (long)seg * t + f;
            if (POOLED) {
                if (r >= b_end) {
                    const int pb = (int)r / pg.t;
                    b_end = (pb + 1) * pg.t;
                    pc = pool_coef(pg, pb, n, col);
                }
                invT = pool_frame_weight(pg, r);
            }
            const f32x4 zz = *(const f32x4*)(z + r * ldz + col);
            const f32x4 dd = upstream_grad<POOLED>(da, pc, invT, r, n, col, zz, sc, sh, relu, sl, hs);
            const f32x4 xh = (zz - mu) * is;
            out = g_is * (dd - c1 - xh * c2);
        }
        *(f32x4*)(dz + (long)dr * ldz + col) = out;
        u += 4;
        while (u >= tp) { u -= tp; ++seg; }
    }
}

// The same pass for a layer without zero frames around its chunks (pad == 0: the dense layers, among them the last frame layer whose
// upstream gradient is the pooling backward - 286 MB at S1, on the critical chain between the loss and the first data-gradient GEMM).
// The loop above loads, computes and stores one row per trip behind two branches, so every trip waits for its own load; here a thread's
// BAF_ROWS / 4 rows are loaded together (addresses clamped instead of predicated) and, POOLED, the statistics of the at most two chunks
// a strip touches are loaded up front and selected per row.  [measured, round 3, r03_elementwise.json] generic form, 1 500 channels:
// 0.62 of 8 TB/s alone, 0.49 in the step (72.6 us for 285.7 MB).
template <bool POOLED>
__global__ __launch_bounds__(256) void bn_bwd_apply_dense_kernel(const float* __restrict__ da, PoolGrad pg, const float* __restrict__ z, int rows,
                                                                 int n, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ coef, int relu,
                                                                 float* __restrict__ dz, const float* __restrict__ slope, int ldz /* rows of z and dz */) {
    XV_EW_PRIORITY();
    constexpr int NR = BAF_ROWS / 4;
    const int col = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int rl = threadIdx.x >> 6;
    if (col >= n) return;
    const int r0 = blockIdx.y * BAF_ROWS + rl;
    f32x4 zz[NR], dd[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) zz[j] = *(const f32x4*)(z + (long)min(r0 + 4 * j, rows - 1) * ldz + col);
    if (!POOLED) {
#pragma unroll
        for (int j = 0; j < NR; ++j) dd[j] = *(const f32x4*)(da + (long)min(r0 + 4 * j, rows - 1) * n + col);
    }
    const f32x4 mu = *(const f32x4*)(mean + col), is = *(const f32x4*)(invstd + col);
    const f32x4 sc = *(const f32x4*)(scale + col), sh = *(const f32x4*)(shift + col);
    const f32x4 c1 = *(const f32x4*)(coef + col), c2 = *(const f32x4*)(coef + n + col);
    const f32x4 g_is = *(const f32x4*)(gamma + col) * is;
    const bool hs = slope != nullptr;
    f32x4 sl = {0, 0, 0, 0};
    if (hs) sl = *(const f32x4*)(slope + col);
    PoolCoef pc0 = {}, pc1 = {};
    int b_end = 0;                                  // first row of the strip's second chunk
    float w[NR];
    if (POOLED) {
        const int nb = rows / pg.t;
        const int b0 = min(blockIdx.y * BAF_ROWS / pg.t, nb - 1);
        b_end = (b0 + 1) * pg.t;
        pc0 = pool_coef(pg, b0, n, col);
        pc1 = pool_coef(pg, min(b0 + 1, nb - 1), n, col);
#pragma unroll
        for (int j = 0; j < NR; ++j) w[j] = pg.w ? pg.w[min(r0 + 4 * j, rows - 1)] : 1.f / (float)pg.t;
        // (a strip of BAF_ROWS rows crosses at most one chunk boundary when pg.t >= BAF_ROWS; shorter chunks take the generic kernel)
    }
    if (POOLED && !hs) {
        // Plain ReLU (or none): with a = y where the unit is on, the pooling backward and the BatchNorm backward are both affine in z,
        //   dz = on ? w (A z + B) + (C z + D) : C z + D,   A = g q sc, B = g (dm + q (sh - mean_p)), C = -g is c2, D = g (is c2 mu - c1), g = gamma is
        // - four fused multiply-adds, a compare and a select per element instead of the ~16 operations of the general form below: at 1 500
        // channels x 23 808 rows that form kept the vector ALUs busy for about half of the pass's memory time (0.60 of 8 TB/s alone against
        // 0.73 for the forward pass over the same bytes, r04_elementwise.json).  [measured, same box] 66.9 -> 65.8 us in the step: the pass is not
        // ALU-bound after all; kept for the shorter code path.
        const f32x4 gc2 = g_is * is * c2;
        const f32x4 C = -gc2, D = gc2 * mu - g_is * c1;
        const f32x4 A0 = g_is * (pc0.q * sc), B0 = g_is * (pc0.dm + pc0.q * (sh - pc0.mean));
        const f32x4 A1 = g_is * (pc1.q * sc), B1 = g_is * (pc1.dm + pc1.q * (sh - pc1.mean));
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int r = r0 + 4 * j;
            const bool second = r >= b_end;
            const f32x4 A = second ? A1 : A0, B = second ? B1 : B0;
            const f32x4 y = zz[j] * sc + sh;
            const f32x4 off = C * zz[j] + D;
            f32x4 on = w[j] * (A * zz[j] + B) + off;
            if (relu) {
                on.x = y.x > 0.f ? on.x : off.x; on.y = y.y > 0.f ? on.y : off.y;
                on.z = y.z > 0.f ? on.z : off.z; on.w = y.w > 0.f ? on.w : off.w;
            }
            if (r < rows) *(f32x4*)(dz + (long)r * ldz + col) = on;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = r0 + 4 * j;
        f32x4 d;
        if (POOLED) {
            const bool second = r >= b_end;
            PoolCoef pc;
            pc.mean = second ? pc1.mean : pc0.mean; pc.dm = second ? pc1.dm : pc0.dm; pc.q = second ? pc1.q : pc0.q;
            d = upstream_grad<true>(nullptr, pc, w[j], (long)r, n, col, zz[j], sc, sh, relu, sl, hs);
        } else {
            const f32x4 y = zz[j] * sc + sh;
            d = dd[j];
            if (relu) {
                if (hs) {
                    d.x = y.x > 0.f ? d.x : d.x * sl.x; d.y = y.y > 0.f ? d.y : d.y * sl.y;
                    d.z = y.z > 0.f ? d.z : d.z * sl.z; d.w = y.w > 0.f ? d.w : d.w * sl.w;
                } else {
                    d.x = y.x > 0.f ? d.x : 0.f; d.y = y.y > 0.f ? d.y : 0.f;
                    d.z = y.z > 0.f ? d.z : 0.f; d.w = y.w > 0.f ? d.w : 0.f;
                }
            }
        }
        const f32x4 xh = (zz[j] - mu) * is;
        if (r < rows) *(f32x4*)(dz + (long)r * ldz + col) = g_is * (d - c1 - xh * c2);
    }
}

// Same as bn_bwd_apply_kernel but dz is written as two fp16 planes [2][segs*(t+2pad)][ldd] scaled by the power of two
// derived from *amax (xv_gemm16.hip); pad rows / columns are zero.
// Thread = one 8-channel chunk (16 B per plane) x a strip of rows: the 7 per-channel parameter vectors are loaded
// once per thread, not once per element (they were 3/4 of the load instructions of the element-per-thread form).
// block = 64 chunks x 4 row lanes, BAS_ROWS padded rows per block.
#define BAS_ROWS 32
template <bool POOLED>
__global__ __launch_bounds__(256) void bn_bwd_apply_split_kernel(const float* __restrict__ da, PoolGrad pg, const float* __restrict__ z,
                                                                 int segs, int t, int n, const float* __restrict__ gamma,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ coef, int relu, int pad,
                                                                 const unsigned* __restrict__ amax, unsigned short* __restrict__ dst,
                                                                 long ldd, long plane_stride, const float* __restrict__ slope) {
    XV_EW_PRIORITY();
    const float s = xv_pow2_scale(*amax);
    const int tp = t + 2 * pad;
    const int col = (blockIdx.x * 64 + (threadIdx.x & 63)) * 8;
    const int rl = threadIdx.x >> 6;
    if (col >= ldd) return;
    const int total_rows = segs * tp;
    const int r0 = blockIdx.y * BAS_ROWS, r1 = min(total_rows, r0 + BAS_ROWS);
    f32x4 g_is[2], mu[2], is[2], sc[2], sh[2], c1[2], c2[2], sl[2];
    bool cv[2];
    const bool hs = slope != nullptr;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = col + 4 * q;
        cv[q] = c < n;
        const int cc = cv[q] ? c : 0;
        sl[q] = hs ? *(const f32x4*)(slope + cc) : f32x4{0, 0, 0, 0};
        mu[q] = *(const f32x4*)(mean + cc); is[q] = *(const f32x4*)(invstd + cc);
        sc[q] = *(const f32x4*)(scale + cc); sh[q] = *(const f32x4*)(shift + cc);
        c1[q] = *(const f32x4*)(coef + cc); c2[q] = *(const f32x4*)(coef + n + cc);
        g_is[q] = *(const f32x4*)(gamma + cc) * is[q];
    }
    int seg = r0 / tp, u = r0 - seg * tp;          // padded row r0 -> (segment, frame + pad)
    u += rl;
    while (u >= tp) { u -= tp; ++seg; }
    PoolCoef pc[2] = {};
    int b_end = 0;                                 // first row beyond the chunk whose statistics are in pc
    float invT = 0.f;
    for (int dr = r0 + rl; dr < r1; dr += 4) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int f = u - pad;
        if (f >= 0 && f < t) {
            const long r = (long)seg * t + f;
            if (POOLED) {
                if (r >= b_end) {
                    const int pb = (int)r / pg.t;
                    b_end = (pb + 1) * pg.t;
#pragma unroll
                    for (int q = 0; q < 2; ++q) pc[q] = pool_coef(pg, pb, n, cv[q] ? col + 4 * q : 0);
                }
                invT = pool_frame_weight(pg, r);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!cv[q]) continue;
                const int c = col + 4 * q;
                f32x4 zz = *(const f32x4*)(z + r * n + c);
                f32x4 dd = upstream_grad<POOLED>(da, pc[q], invT, r, n, c, zz, sc[q], sh[q], relu, sl[q], hs);
                f32x4 xh = (zz - mu[q]) * is[q];
                f32x4 o = g_is[q] * (dd - c1[q] - xh * c2[q]);
                v[4 * q] = o.x; v[4 * q + 1] = o.y; v[4 * q + 2] = o.z; v[4 * q + 3] = o.w;
            }
        }
        unsigned short h[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float xs = v[j] * s;
            _Float16 hh = (_Float16)xs;
            _Float16 ll = (_Float16)(xs - (float)hh);
            h[j] = __builtin_bit_cast(unsigned short, hh);
            l[j] = __builtin_bit_cast(unsigned short, ll);
        }
        *(uint4*)(dst + (long)dr * ldd + col) = *(const uint4*)h;
        *(uint4*)(dst + plane_stride + (long)dr * ldd + col) = *(const uint4*)l;
        u += 4;
        while (u >= tp) { u -= tp; ++seg; }
    }
}

// BatchNorm backward reductions of a layer that feeds statistics pooling, WITHOUT a pass over z (plain ReLU or no activation).
// With a = act(y), y = gamma*xhat + beta, frame weights omega (1/T or the attention weights, sum 1) and the pooled mean / variance
// (mu, var) of chunk b, the upstream gradient on an active frame is  dd = omega*(dm + q*(a - mu)),  q = dstd/std, and xhat = (a - beta)/gamma
// there; off frames contribute nothing.  Summed over the frames of the chunk, with W+ = the weight on active frames (pooling forward):
//   sum dd        = dm*W+ + q*mu*(1 - W+)
//   sum dd*xhat   = (dm*(mu - beta*W+) + q*(var - beta*mu*(1 - W+))) / gamma
// (sum_on omega*a = mu and sum_on omega*a^2 = var + mu^2 because a = 0 off).  One workgroup = 4 channel quads x 64 chunk lanes (the
// kernel is a handful of dependent memory round trips: 16 quads x 16 lanes, 8 chunks per lane, took 20 us), chunks summed in a fixed
// order; also does bn_bwd_finalize_kernel's job.  gamma == 0 (xhat not recoverable from a) yields inf / nan - loudly.
#define PS_QUADS 4
#define PS_LANES 64
__global__ __launch_bounds__(256) void bn_bwd_pooled_stats_kernel(PoolGrad pg, int segs, int n, int rows, const float* __restrict__ gamma,
                                                                  const float* __restrict__ shift, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, const float* __restrict__ scale,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ coef, float* __restrict__ dbias,
                                                                  const float* __restrict__ zmin, const float* __restrict__ zmax,
                                                                  unsigned* __restrict__ dz_amax) {
    XV_EW_PRIORITY();
    __shared__ f32x4 r1[PS_LANES][PS_QUADS], r2[PS_LANES][PS_QUADS], r3[PS_LANES][PS_QUADS];
    const int cq = threadIdx.x & (PS_QUADS - 1), bl = threadIdx.x / PS_QUADS;
    const int col = (blockIdx.x * PS_QUADS + cq) * 4;
    const bool cv = col < n;
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0}, s3 = {0, 0, 0, 0};
    const float invT = 1.f / (float)pg.t;
    if (cv) {
        // beta = shift + mean*scale (shift = beta - mean*scale, bn_finalize)
        const f32x4 bt = *(const f32x4*)(shift + col) + *(const f32x4*)(mean + col) * *(const f32x4*)(scale + col);
        for (int b = bl; b < segs; b += PS_LANES) {
            const PoolCoef pc = pool_coef(pg, b, n, col);
            const f32x4 sd = *(const f32x4*)(pg.out + (long)b * 2 * n + n + col);
            const f32x4 wp = *(const f32x4*)(pg.wpos + (long)b * n + col);
            const f32x4 off = f32x4{1.f, 1.f, 1.f, 1.f} - wp;
            s1 += pc.dm * wp + pc.q * pc.mean * off;
            s2 += pc.dm * (pc.mean - bt * wp) + pc.q * (sd * sd - bt * pc.mean * off);
            if (dz_amax) {
                // |d a| over the chunk (unit frame weights): d a = (dm + q*(a - mu)) / T is linear in a, a in [0, amax] -> the ends
                const f32x4 am = *(const f32x4*)(pg.amax + (long)b * n + col);
                const f32x4 e0 = pc.dm - pc.q * pc.mean, e1 = pc.dm + pc.q * (am - pc.mean);
                s3.x = fmaxf(s3.x, fmaxf(fabsf(e0.x), fabsf(e1.x)) * invT); s3.y = fmaxf(s3.y, fmaxf(fabsf(e0.y), fabsf(e1.y)) * invT);
                s3.z = fmaxf(s3.z, fmaxf(fabsf(e0.z), fabsf(e1.z)) * invT); s3.w = fmaxf(s3.w, fmaxf(fabsf(e0.w), fabsf(e1.w)) * invT);
            }
        }
    }
    r1[bl][cq] = s1; r2[bl][cq] = s2; r3[bl][cq] = s3;
    __syncthreads();
    if (bl != 0 || !cv) return;
    s1 = r1[0][cq]; s2 = r2[0][cq];
    for (int k = 1; k < PS_LANES; ++k) {
        s1 += r1[k][cq]; s2 += r2[k][cq];
        s3.x = fmaxf(s3.x, r3[k][cq].x); s3.y = fmaxf(s3.y, r3[k][cq].y); s3.z = fmaxf(s3.z, r3[k][cq].z); s3.w = fmaxf(s3.w, r3[k][cq].w);
    }
    const f32x4 g = *(const f32x4*)(gamma + col), is = *(const f32x4*)(invstd + col);
    s2.x /= g.x; s2.y /= g.y; s2.z /= g.z; s2.w /= g.w;
    const float inv_rows = 1.0f / (float)rows;
    const f32x4 c1 = s1 * inv_rows;
    *(f32x4*)(dbeta + col) = s1;
    *(f32x4*)(dgamma + col) = s2;
    *(f32x4*)(coef + col) = c1;
    *(f32x4*)(coef + n + col) = s2 * inv_rows;
    if (dbias) *(f32x4*)(dbias + col) = g * is * (s1 - c1 * (float)rows);
    if (dz_amax) {
        // upper bound of |dz| = |gamma*invstd| * |d a - c1 - xhat*c2| over the batch (bn_bwd_finalize_kernel's, with the analytic |d a| bound)
        const f32x4 mu = *(const f32x4*)(mean + col), zn = *(const f32x4*)(zmin + col), zx = *(const f32x4*)(zmax + col);
        float bound = 0.f;
        const float s3v[4] = {s3.x, s3.y, s3.z, s3.w}, s1v[4] = {c1.x, c1.y, c1.z, c1.w}, s2v[4] = {s2.x, s2.y, s2.z, s2.w};
        const float gv[4] = {g.x, g.y, g.z, g.w}, iv[4] = {is.x, is.y, is.z, is.w}, muv[4] = {mu.x, mu.y, mu.z, mu.w};
        const float znv[4] = {zn.x, zn.y, zn.z, zn.w}, zxv[4] = {zx.x, zx.y, zx.z, zx.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = fmaxf(fabsf(zxv[j] - muv[j]), fabsf(znv[j] - muv[j])) * iv[j];
            bound = fmaxf(bound, fabsf(gv[j] * iv[j]) * (s3v[j] + fabsf(s1v[j]) + xh * fabsf(s2v[j] * inv_rows)) * 1.0001f);
        }
        atomicMax(dz_amax, __float_as_uint(bound));
    }
}

static int bn_relu_backward_impl(hipStream_t s, const float* da, PoolGrad pg, const float* z, int segs, int t, int n, const float* gamma,
                                 const float* mean, const float* invstd, const float* scale, const float* shift, int relu, int pad,
                                 float* dz_pad, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes, int ldz = 0) {
    XV_REQUIRE(segs > 0 && t > 0 && n > 0 && n % 4 == 0 && pad >= 0, "bn_relu_backward: bad shape (n=%d must be a multiple of 4)", n);
    // ldz: leading dimension of z and dz when their rows are padded (the pooled layer: rows on the 128-byte grid); closed-form pooled path only
    if (ldz == 0) ldz = n;
    XV_REQUIRE(ldz >= n && ldz % 4 == 0 && (ldz == n || (pg.out && pg.wpos && !(relu && g_act.slope))), "bn_relu_backward: a row pitch is only supported on the closed-form pooled path");
    XV_REQUIRE((long)segs * (t + 2 * pad) * (n / 4) < (1L << 31), "bn_relu_backward: tensor too large for 32-bit indexing");
    const int rows = segs * t;
    const bool pooled = pg.out != nullptr;
    XV_REQUIRE(!pooled || (pg.t > 0 && rows % pg.t == 0), "bn_relu_backward: %d rows are not whole chunks of %d pooled frames", rows, pg.t);
    int chunks = pooled ? bn_bwd_pooled_chunks(rows, pg.t) : xv_cdiv(rows, BB_ROWS);
    const XvActContext act = g_act;
    const int nstat = (relu && act.slope && act.dalpha) ? 4 : 3;      // prelu: one more reduction, sum d act * min(y, 0)
    size_t need = ((size_t)chunks * nstat * n + 2 * n) * sizeof(float);
    XV_REQUIRE(need <= ws_bytes, "bn_relu_backward: workspace too small (%zu > %zu)", need, ws_bytes);
    float* part = (float*)ws;
    float* coef = part + (size_t)chunks * nstat * n;
    if (pooled && pg.wpos && !(relu && act.slope)) {
        // closed form from the pooled statistics: no pass over z, no finalize launch
        hipLaunchKernelGGL(bn_bwd_pooled_stats_kernel, dim3(xv_cdiv(n / 4, PS_QUADS)), dim3(256), 0, s, pg, rows / pg.t, n, rows, gamma, shift, mean,
                           invstd, scale, dgamma, dbeta, coef, dbias, (const float*)nullptr, (const float*)nullptr, (unsigned*)nullptr);
        XV_LAUNCH_CHECK();
    } else {
        if (pooled)
            launch_bn_bwd_reduce_pooled(s, pg, z, rows, n, mean, invstd, scale, shift, relu, part, relu ? act.slope : nullptr, nstat, &chunks);
        else
            hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(xv_cdiv(n / 4, 64), chunks), dim3(256), 0, s,
                               da, z, rows, n, mean, invstd, scale, shift, relu, part, relu ? act.slope : nullptr, nstat);
        XV_LAUNCH_CHECK();
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(xv_cdiv(n, FIN_CH)), dim3(256), 0, s, (const float*)part, chunks, n, rows,
                           dgamma, dbeta, coef, gamma, invstd, dbias, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           (unsigned*)nullptr, nstat, nstat == 4 ? act.dalpha : (float*)nullptr);
        XV_LAUNCH_CHECK();
    }
    dim3 agrid(xv_cdiv(n / 4, 64), xv_cdiv(segs * (t + 2 * pad), BAF_ROWS));
    // [measured, round 6, profiles/r06_pooled_kernels.txt] two other forms of the pooled pass were built and dropped: a workgroup per (chunk, 256
    // channels) with the parameters set up once and two batches of eight loads in flight (78 us for statistics + apply alone, as this strip form),
    // and whole-row workgroups that stream consecutive bytes as torch's flat element-wise kernel does (82 us)
    if (pad == 0 && (!pooled || pg.t >= BAF_ROWS))
        hipLaunchKernelGGL(pooled ? bn_bwd_apply_dense_kernel<true> : bn_bwd_apply_dense_kernel<false>, agrid, dim3(256), 0, s, da, pg, z, rows, n, gamma,
                           mean, invstd, scale, shift, (const float*)coef, relu, dz_pad, relu ? act.slope : nullptr, ldz);
    else
        hipLaunchKernelGGL(pooled ? bn_bwd_apply_kernel<true> : bn_bwd_apply_kernel<false>, agrid, dim3(256), 0, s, da,
                           pg, z, segs, t, n, gamma, mean, invstd, scale, shift, (const float*)coef, relu, pad, dz_pad, relu ? act.slope : nullptr, ldz);
    XV_LAUNCH_CHECK();
    return 0;
}

static int bn_relu_backward_split_impl(hipStream_t s, const float* da, PoolGrad pg, const float* z, int segs, int t, int n,
                                       const float* gamma, const float* mean, const float* invstd, const float* scale, const float* shift,
                                       const float* zmin, const float* zmax, int relu, int pad, void* dz_planes, int ldp,
                                       size_t plane_stride, uint32_t* dz_amax, float* dgamma, float* dbeta, float* dbias, void* ws,
                                       size_t ws_bytes, const float* ext_part = nullptr, int ext_chunks = 0, bool zero_amax = true) {
    XV_REQUIRE(segs > 0 && t > 0 && n > 0 && n % 4 == 0 && pad >= 0, "bn_relu_backward_split: bad shape (n=%d must be a multiple of 4)", n);
    XV_REQUIRE(ldp % 8 == 0 && ldp >= n && plane_stride % 8 == 0 && zmin && zmax && dz_amax, "bn_relu_backward_split: bad plane arguments");
    XV_REQUIRE((long)segs * (t + 2 * pad) * (ldp / 8) < (1L << 31), "bn_relu_backward_split: tensor too large for 32-bit indexing");
    const int rows = segs * t;
    // the reduction partials either come from the data-gradient GEMM's epilogue (ext_part, one chunk per 128-row tile) or
    // are computed here from (da, z)
    const bool pooled = pg.out != nullptr;
    XV_REQUIRE(!pooled || (pg.t > 0 && rows % pg.t == 0), "bn_relu_backward_split: %d rows are not whole chunks of %d pooled frames", rows, pg.t);
    int chunks = ext_part ? ext_chunks : pooled ? bn_bwd_pooled_chunks(rows, pg.t) : xv_cdiv(rows, BB_ROWS);
    const XvActContext act = g_act;
    XV_REQUIRE(!(ext_part && relu && act.slope), "bn_relu_backward_split: GEMM-epilogue partials only exist for a plain ReLU");
    const int nstat = (relu && act.slope && act.dalpha) ? 4 : 3;
    size_t need = ((size_t)(ext_part ? 0 : chunks) * nstat * n + 2 * n) * sizeof(float);
    XV_REQUIRE(need <= ws_bytes, "bn_relu_backward_split: workspace too small (%zu > %zu)", need, ws_bytes);
    float* part = ext_part ? const_cast<float*>(ext_part) : (float*)ws;
    float* coef = ext_part ? (float*)ws : part + (size_t)chunks * nstat * n;
    if (zero_amax) XV_CHECK_HIP(hipMemsetAsync(dz_amax, 0, sizeof(uint32_t), s));
    if (pooled && pg.wpos && pg.amax && !pg.w && !ext_part && !(relu && act.slope)) {
        // statistics pooling, plain ReLU: reductions, finalize and the |dz| bound in closed form from the pooled statistics (no pass over z)
        hipLaunchKernelGGL(bn_bwd_pooled_stats_kernel, dim3(xv_cdiv(n / 4, PS_QUADS)), dim3(256), 0, s, pg, rows / pg.t, n, rows, gamma, shift, mean,
                           invstd, scale, dgamma, dbeta, coef, dbias, zmin, zmax, (unsigned*)dz_amax);
        XV_LAUNCH_CHECK();
    } else {
        if (!ext_part) {
            if (pooled)
                launch_bn_bwd_reduce_pooled(s, pg, z, rows, n, mean, invstd, scale, shift, relu, part, relu ? act.slope : nullptr, nstat, &chunks);
            else
                hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(xv_cdiv(n / 4, 64), chunks), dim3(256), 0, s,
                                   da, z, rows, n, mean, invstd, scale, shift, relu, part, relu ? act.slope : nullptr, nstat);
            XV_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(xv_cdiv(n, FIN_CH)), dim3(256), 0, s, (const float*)part, chunks, n, rows,
                           dgamma, dbeta, coef, gamma, invstd, dbias, mean, zmin, zmax, (unsigned*)dz_amax, nstat,
                           nstat == 4 ? act.dalpha : (float*)nullptr);
        XV_LAUNCH_CHECK();
    }
    dim3 agrid(xv_cdiv(ldp / 8, 64), xv_cdiv(segs * (t + 2 * pad), BAS_ROWS));
    hipLaunchKernelGGL(pooled ? bn_bwd_apply_split_kernel<true> : bn_bwd_apply_split_kernel<false>, agrid,
                       dim3(256), 0, s, da, pg, z, segs, t, n, gamma, mean, invstd, scale, shift, (const float*)coef, relu, pad,
                       (const unsigned*)dz_amax, (unsigned short*)dz_planes, (long)ldp, (long)plane_stride, relu ? act.slope : nullptr);
    XV_LAUNCH_CHECK();
    return 0;
}

extern "C" int xv_bn_relu_backward(void* stream, const float* da, const float* z, int segs, int t, int n, const float* gamma,
                                   const float* mean, const float* invstd, const float* scale, const float* shift, int relu,
                                   int pad, float* dz_pad, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes) {
    XV_REQUIRE(da, "bn_relu_backward: null upstream gradient");
    PoolGrad pg = {nullptr, nullptr, 1, nullptr, nullptr, nullptr};
    return bn_relu_backward_impl((hipStream_t)stream, da, pg, z, segs, t, n, gamma, mean, invstd, scale, shift, relu, pad, dz_pad, dgamma,
                                 dbeta, dbias, ws, ws_bytes);
}

extern "C" int xv_bn_relu_backward_split(void* stream, const float* da, const float* z, int segs, int t, int n, const float* gamma,
                                         const float* mean, const float* invstd, const float* scale, const float* shift,
                                         const float* zmin, const float* zmax, int relu, int pad, void* dz_planes, int ldp,
                                         size_t plane_stride, uint32_t* dz_amax, float* dgamma, float* dbeta, float* dbias, void* ws,
                                         size_t ws_bytes) {
    XV_REQUIRE(da, "bn_relu_backward_split: null upstream gradient");
    PoolGrad pg = {nullptr, nullptr, 1, nullptr, nullptr, nullptr};
    return bn_relu_backward_split_impl((hipStream_t)stream, da, pg, z, segs, t, n, gamma, mean, invstd, scale, shift, zmin, zmax, relu, pad,
                                       dz_planes, ldp, plane_stride, dz_amax, dgamma, dbeta, dbias, ws, ws_bytes);
}

// engine-internal form of the three public variants (xv_common.h)
int xv_bn_relu_backward_split_ex(hipStream_t s, const XvBnBwdSplit& x, const float* z, int segs, int t, int n, const float* gamma,
                                 const float* mean, const float* invstd, const float* scale, const float* shift, const float* zmin,
                                 const float* zmax, int relu, int pad, void* dz_planes, int ldp, size_t plane_stride, uint32_t* dz_amax,
                                 float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes) {
    PoolGrad pg = {x.pool_out, x.dpool, x.pool_out ? x.pool_t : 1, x.weights, x.pool_out ? x.wpos : nullptr, x.pool_out ? x.pamax : nullptr};
    return bn_relu_backward_split_impl(s, x.da, pg, z, segs, t, n, gamma, mean, invstd, scale, shift, zmin, zmax, relu, pad, dz_planes, ldp,
                                       plane_stride, dz_amax, dgamma, dbeta, dbias, ws, ws_bytes, x.ext_part, x.ext_chunks, x.zero_amax);
}

// xv_bn_relu_backward_split with the reduction partials already produced by xv_affine_dgrad_bnstats_f16x3 (the pass over
// (da, z) that computes them is skipped): part [chunks][3][n], chunks = ceil(rows / 128), ReLU layers, pad as usual.
extern "C" int xv_bn_relu_backward_split_from_part(void* stream, const float* part, int chunks, const float* da, const float* z, int segs,
                                                   int t, int n, const float* gamma, const float* mean, const float* invstd,
                                                   const float* scale, const float* shift, const float* zmin, const float* zmax, int pad,
                                                   void* dz_planes, int ldp, size_t plane_stride, uint32_t* dz_amax, float* dgamma,
                                                   float* dbeta, float* dbias, void* ws, size_t ws_bytes) {
    XV_REQUIRE(da && part && chunks == xv_cdiv(segs * t, XV_TILE_M), "bn_relu_backward_split_from_part: one chunk per 128-row tile expected");
    PoolGrad pg = {nullptr, nullptr, 1, nullptr, nullptr, nullptr};
    return bn_relu_backward_split_impl((hipStream_t)stream, da, pg, z, segs, t, n, gamma, mean, invstd, scale, shift, zmin, zmax, 1, pad,
                                       dz_planes, ldp, plane_stride, dz_amax, dgamma, dbeta, dbias, ws, ws_bytes, part, chunks);
}

extern "C" int xv_bn_relu_backward_pooled(void* stream, const float* pool_out, const float* dpool, const float* weights, int b, int t,
                                          const float* z, int n,
                                          const float* gamma, const float* mean, const float* invstd, const float* scale,
                                          const float* shift, int relu, float* dz, float* dgamma, float* dbeta, float* dbias, void* ws,
                                          size_t ws_bytes) {
    XV_REQUIRE(pool_out && dpool && b > 0 && t > 0, "bn_relu_backward_pooled: bad arguments");
    PoolGrad pg = {pool_out, dpool, t, weights, nullptr, nullptr};
    return bn_relu_backward_impl((hipStream_t)stream, nullptr, pg, z, b * t, 1, n, gamma, mean, invstd, scale, shift, relu, 0, dz, dgamma,
                                 dbeta, dbias, ws, ws_bytes);
}

extern "C" int xv_bn_relu_backward_pooled_aux(void* stream, const float* pool_out, const float* dpool, const float* weights, const float* wpos,
                                              int b, int t, const float* z, int n, const float* gamma, const float* mean, const float* invstd,
                                              const float* scale, const float* shift, int relu, float* dz, float* dgamma, float* dbeta,
                                              float* dbias, void* ws, size_t ws_bytes) {
    XV_REQUIRE(wpos, "bn_relu_backward_pooled_aux: wpos is required");
    return xv_bn_relu_backward_pooled_ex((hipStream_t)stream, pool_out, dpool, weights, wpos, b, t, z, n, gamma, mean, invstd, scale, shift, relu,
                                         dz, dgamma, dbeta, dbias, ws, ws_bytes, 0);
}

// Engine form: wpos from xv_stat_pool_forward_bn_ex (see PoolGrad) - the reductions then need no pass over z
int xv_bn_relu_backward_pooled_ex(hipStream_t s, const float* pool_out, const float* dpool, const float* weights, const float* wpos, int b, int t,
                                  const float* z, int n, const float* gamma, const float* mean, const float* invstd, const float* scale,
                                  const float* shift, int relu, float* dz, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes, int ldz) {
    XV_REQUIRE(pool_out && dpool && b > 0 && t > 0, "bn_relu_backward_pooled: bad arguments");
    PoolGrad pg = {pool_out, dpool, t, weights, wpos, nullptr};
    return bn_relu_backward_impl(s, nullptr, pg, z, b * t, 1, n, gamma, mean, invstd, scale, shift, relu, 0, dz, dgamma, dbeta, dbias, ws, ws_bytes, ldz);
}

extern "C" int xv_bn_relu_backward_pooled_split(void* stream, const float* pool_out, const float* dpool, const float* weights, int b, int t,
                                                const float* z, int n,
                                                const float* gamma, const float* mean, const float* invstd, const float* scale,
                                                const float* shift, const float* zmin, const float* zmax, int relu, void* dz_planes,
                                                int ldp, size_t plane_stride, uint32_t* dz_amax, float* dgamma, float* dbeta,
                                                float* dbias, void* ws, size_t ws_bytes) {
    XV_REQUIRE(pool_out && dpool && b > 0 && t > 0, "bn_relu_backward_pooled_split: bad arguments");
    PoolGrad pg = {pool_out, dpool, t, weights, nullptr, nullptr};
    return bn_relu_backward_split_impl((hipStream_t)stream, nullptr, pg, z, b * t, 1, n, gamma, mean, invstd, scale, shift, zmin, zmax, relu,
                                       0, dz_planes, ldp, plane_stride, dz_amax, dgamma, dbeta, dbias, ws, ws_bytes);
}

// ------------------------------------------------------------------------------------
// small-row BatchNorm, one launch (xv_common.h).  block = 256 threads = 16 channels x 16 row lanes; fixed-order combines.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float small_reduce16(float v, float (*red)[16], int rl, int cx) {
    __syncthreads();
    red[rl][cx] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    return t;
}

__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const float* __restrict__ z, int rows, int n, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float momentum, int unbiased,
                                                           float* __restrict__ mmean, float* __restrict__ mvar, float* __restrict__ mean_o,
                                                           float* __restrict__ invstd_o, float* __restrict__ scale_o,
                                                           float* __restrict__ shift_o, int relu, float* __restrict__ a,
                                                           const float* __restrict__ slope) {
    XV_EW_PRIORITY();
    __shared__ float red[16][16];
    const int cx = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    const bool cv = c < n;
    float s = 0.f;
    if (cv) for (int r = rl; r < rows; r += 16) s += z[(long)r * n + c];
    const float mean = small_reduce16(s, red, rl, cx) / (float)rows;
    float q = 0.f;
    if (cv) for (int r = rl; r < rows; r += 16) { float d = z[(long)r * n + c] - mean; q += d * d; }
    const float var = small_reduce16(q, red, rl, cx) / (float)rows;      // biased, two-pass (tf.nn.moments)
    if (!cv) return;
    const float invstd = 1.0f / sqrtf(var + eps);
    const float sc = gamma[c] * invstd, sh = beta[c] - mean * sc;
    if (rl == 0) {
        mean_o[c] = mean; invstd_o[c] = invstd; scale_o[c] = sc; shift_o[c] = sh;
        if (mmean) {
            float v = (unbiased && rows > 1) ? var * ((float)rows / (float)(rows - 1)) : var;
            mmean[c] = mmean[c] * momentum + mean * (1.0f - momentum);
            mvar[c] = mvar[c] * momentum + v * (1.0f - momentum);
        }
    }
    if (a)
        for (int r = rl; r < rows; r += 16) {
            float y = z[(long)r * n + c] * sc + sh;
            a[(long)r * n + c] = relu ? (slope ? act1(y, slope[c]) : fmaxf(y, 0.f)) : y;
        }
}

int xv_bn_small_forward(hipStream_t s, const float* z, int rows, int n, const float* gamma, const float* beta, float eps, float momentum,
                        int unbiased_moving, float* moving_mean, float* moving_var, float* mean, float* invstd, float* scale, float* shift,
                        int relu, float* a) {
    XV_REQUIRE(rows > 0 && rows <= XV_BN_SMALL_MAX_ROWS && n > 0, "bn_small_forward: bad shape (rows=%d)", rows);
    hipLaunchKernelGGL(bn_small_fwd_kernel, dim3(xv_cdiv(n, 16)), dim3(256), 0, s, z, rows, n, gamma, beta, eps, momentum, unbiased_moving,
                       moving_mean, moving_var, mean, invstd, scale, shift, relu, a, relu ? g_act.slope : nullptr);
    XV_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const float* __restrict__ da, const float* __restrict__ z, int rows, int n,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu, float* __restrict__ dz,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ dbias, const float* __restrict__ slope,
                                                           float* __restrict__ dalpha) {
    XV_EW_PRIORITY();
    __shared__ float red[16][16];
    const int cx = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    const bool cv = c < n;
    float mu = 0.f, is = 0.f, sc = 0.f, sh = 0.f, sl = 0.f;
    if (cv) { mu = mean[c]; is = invstd[c]; sc = scale[c]; sh = shift[c]; sl = slope ? slope[c] : 0.f; }
    float s1 = 0.f, s2 = 0.f, s4 = 0.f;
    if (cv)
        for (int r = rl; r < rows; r += 16) {
            float zz = z[(long)r * n + c], dd = da[(long)r * n + c];
            const float y = zz * sc + sh;
            if (relu) { s4 += dd * fminf(y, 0.f); if (!(y > 0.f)) dd *= sl; }
            s1 += dd;
            s2 += dd * ((zz - mu) * is);
        }
    s1 = small_reduce16(s1, red, rl, cx);
    s2 = small_reduce16(s2, red, rl, cx);
    if (dalpha) s4 = small_reduce16(s4, red, rl, cx);      // (uniform: every thread of the block takes the same path)
    if (!cv) return;
    const float c1 = s1 / (float)rows, c2 = s2 / (float)rows;
    const float g = gamma[c] * is;
    if (rl == 0) {
        dbeta[c] = s1; dgamma[c] = s2;
        if (dbias) dbias[c] = g * (s1 - c1 * (float)rows);
        if (dalpha) dalpha[c] = s4;
    }
    for (int r = rl; r < rows; r += 16) {
        float zz = z[(long)r * n + c], dd = da[(long)r * n + c];
        if (relu && !(zz * sc + sh > 0.f)) dd *= sl;
        dz[(long)r * n + c] = g * (dd - c1 - ((zz - mu) * is) * c2);
    }
}

int xv_bn_small_backward(hipStream_t s, const float* da, const float* z, int rows, int n, const float* gamma, const float* mean,
                         const float* invstd, const float* scale, const float* shift, int relu, float* dz, float* dgamma, float* dbeta,
                         float* dbias) {
    XV_REQUIRE(rows > 0 && rows <= XV_BN_SMALL_MAX_ROWS && n > 0, "bn_small_backward: bad shape (rows=%d)", rows);
    hipLaunchKernelGGL(bn_small_bwd_kernel, dim3(xv_cdiv(n, 16)), dim3(256), 0, s, da, z, rows, n, gamma, mean, invstd, scale, shift, relu,
                       dz, dgamma, dbeta, dbias, relu ? g_act.slope : nullptr, (relu && g_act.slope) ? g_act.dalpha : nullptr);
    XV_LAUNCH_CHECK();
    return 0;
}

__global__ void relu_bwd_kernel(const float* __restrict__ da, const float* __restrict__ a, size_t count, float* __restrict__ dz) {
    XV_EW_PRIORITY();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        dz[i] = a[i] > 0.f ? da[i] : 0.f;
}
// The same with a slope (act context): forward a = act(z) and backward dz = da * act'(z), d alpha[c] = sum_r da * min(z, 0); rows <= XV_BN_SMALL_MAX_ROWS
__global__ __launch_bounds__(256) void act_small_kernel(const float* __restrict__ da, const float* __restrict__ z, int rows, int n,
                                                        const float* __restrict__ slope, float* __restrict__ out,
                                                        float* __restrict__ dalpha) {
    XV_EW_PRIORITY();
    __shared__ float red[16][16];
    const int cx = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    const bool cv = c < n;
    const float sl = cv ? slope[c] : 0.f;
    float s4 = 0.f;
    if (cv)
        for (int r = rl; r < rows; r += 16) {
            const float zz = z[(long)r * n + c];
            if (da) {
                const float dd = da[(long)r * n + c];
                s4 += dd * fminf(zz, 0.f);
                out[(long)r * n + c] = zz > 0.f ? dd : dd * sl;
            } else {
                out[(long)r * n + c] = act1(zz, sl);
            }
        }
    if (dalpha) {
        s4 = small_reduce16(s4, red, rl, cx);
        if (cv && rl == 0) dalpha[c] = s4;
    }
}
int xv_act_small(hipStream_t s, const float* da, const float* z, int rows, int n, float* out) {
    XV_REQUIRE(g_act.slope && rows > 0 && rows <= XV_BN_SMALL_MAX_ROWS && n > 0, "act_small: needs an activation slope and a segment-level tensor");
    hipLaunchKernelGGL(act_small_kernel, dim3(xv_cdiv(n, 16)), dim3(256), 0, s, da, z, rows, n, g_act.slope, out, da ? g_act.dalpha : (float*)nullptr);
    XV_LAUNCH_CHECK();
    return 0;
}

// y[r][c] = x > 0 ? x : alpha[c] * x  (common.py:27-42 prelu = relu(x) + alpha (x - |x|) / 2; a constant alpha = leaky ReLU)
__global__ void prelu_fwd_kernel(const float* __restrict__ x, size_t count, int n, const float* __restrict__ alpha, float* __restrict__ y) {
    XV_EW_PRIORITY();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) y[i] = act1(x[i], alpha[i % n]);
}
extern "C" int xv_prelu_forward(void* stream, const float* x, int rows, int n, const float* alpha, float* y) {
    XV_REQUIRE(x && alpha && y && rows > 0 && n > 0, "prelu_forward: bad arguments");
    const size_t count = (size_t)rows * n;
    hipLaunchKernelGGL(prelu_fwd_kernel, dim3(grid_for((long)count, 256)), dim3(256), 0, (hipStream_t)stream, x, count, n, alpha, y);
    XV_LAUNCH_CHECK();
    return 0;
}

extern "C" int xv_relu_backward(void* stream, const float* da, const float* a, size_t count, float* dz) {
    XV_REQUIRE(count > 0, "relu_backward: empty");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for((long)count, 256)), dim3(256), 0, (hipStream_t)stream, da, a, count, dz);
    XV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------
// statistics pooling (pooling.py:9-34)
// wave = 64 channel-quads of one frame, block = 4 waves = 4 frame lanes.  Each lane runs Welford over its frames for 4 channels (two
// interleaved chains); chains and waves are merged with Chan's formula (through LDS across waves).
// ------------------------------------------------------------------------------------
struct Wf4 { f32x4 mean, m2; float n; };
__device__ __forceinline__ void wf_merge(f32x4& mean, f32x4& m2, float& n, const f32x4& mean_b, const f32x4& m2_b, float n_b) {
    float nn = n + n_b;
    if (nn > 0.f) {
        f32x4 d = mean_b - mean;
        float w = n_b / nn;
        mean = mean + d * w;
        m2 = m2 + m2_b + d * d * (n * w);
    }
    n = nn;
}

// BN: the pooled tensor is relu?(x*scale + shift) evaluated on the fly (x = the pre-BN output of the last frame layer), so the
// activation is never written to memory.
// A wave reads ONE contiguous KiB per instruction (64 channel quads of one frame), the 4 waves are 4 frame lanes, every lane keeps 8
// loads in flight and runs TWO Welford chains (alternate frames of its lane), merged at the end.  [measured, S1: 143 MB] 33 us warm
// (Infinity Cache) / 54 us cold with event overhead - what a plain streaming read of the tensor takes (own amax kernel 32 / 53 us,
// torch.sum 35 / 60 us); the first form (32 quads x 8 frame lanes, 4 loads, one chain) took 41 / 61 us.
template <bool BN>
__global__ __launch_bounds__(256) void stat_pool_fwd_kernel(const float* __restrict__ x, int T, int C, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu, const float* __restrict__ wts,
                                                            float* __restrict__ out, const float* __restrict__ slope,
                                                            float* __restrict__ wpos, float* __restrict__ amax_o,
                                                            const int* __restrict__ flen, int shrink, int ld /* floats per row of x (>= C) */) {
    XV_EW_PRIORITY();
    __shared__ f32x4 s_mean[4][64], s_m2[4][64], s_wp[4][64], s_mx[4][64];
    __shared__ float s_n[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * 4;
    const int b = blockIdx.y;
    const bool cv = col < C;
    const float* xp = x + (long)b * T * ld + (cv ? col : 0);
    // flen (batched extraction): chunk b holds flen[b] - shrink valid frames, the rest of its T rows is padding that is not pooled
    const int Tstride = T;
    if (flen) T = max(1, min(T, flen[b] - shrink));
    f32x4 sc = {1, 1, 1, 1}, sh = {0, 0, 0, 0}, sl = {0, 0, 0, 0};
    const bool hs = BN && slope != nullptr;
    if (BN && cv) { sc = *(const f32x4*)(scale + col); sh = *(const f32x4*)(shift + col); if (hs) sl = *(const f32x4*)(slope + col); }
    auto act = [&](f32x4 v) {
        if (BN) {
            v = v * sc + sh;
            if (relu) v = hs ? act4(v, sl) : relu4(v);
        }
        return v;
    };
    const float* wp = wts ? wts + (long)b * Tstride : nullptr;
    f32x4 mean0 = {0, 0, 0, 0}, m20 = {0, 0, 0, 0}, mean1 = {0, 0, 0, 0}, m21 = {0, 0, 0, 0};
    f32x4 wpp = {0, 0, 0, 0};       // sum of the frame weights where the activation is on (all frames without a ReLU)
    const bool cnt_all = !(BN && relu);
    f32x4 amx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    auto on = [&](f32x4 a, float w) {
        wpp.x += (cnt_all || a.x > 0.f) ? w : 0.f; wpp.y += (cnt_all || a.y > 0.f) ? w : 0.f;
        wpp.z += (cnt_all || a.z > 0.f) ? w : 0.f; wpp.w += (cnt_all || a.w > 0.f) ? w : 0.f;
        if (amax_o) { amx.x = fmaxf(amx.x, a.x); amx.y = fmaxf(amx.y, a.y); amx.z = fmaxf(amx.z, a.z); amx.w = fmaxf(amx.w, a.w); }      // (split precision only)
    };
    float n0 = 0.f, n1 = 0.f;
    // frame weights: 1 (statistics pooling; n counts frames) or the attention weights of this chunk (n sums them); weighted incremental
    // mean / M2 (West), identical to Welford for unit weights.  Unit weights: 1 / n through v_rcp_f32 (1 ulp, exact for n = 1) - the
    // IEEE division sequence sat on the serial mean -> M2 chain of every frame and made this pass VALU-latency-bound.  Attention weights
    // keep the exact quotient: there w / n must be exactly 1 on a lane's first frame, or a constant chunk no longer has a zero variance
    // (reference test_utils.py / pooling.py:160-162 clamp).
#define XV_POOL_STEP2(mean, m2, n, v, w) { n += (w); const f32x4 d_ = (v) - mean; if (n > 0.f) mean += d_ * (wp ? (w) / n : __builtin_amdgcn_rcpf(n)); m2 += d_ * ((v) - mean) * (w); }
    // [measured, round 6] two batches of four loads in flight (the next batch issued before the current one is folded): 26.4 us alone against
    // 26.5 for this form; two batches of eight need 204 VGPRs (a workgroup less per CU)
    int t = wave;
    for (; t + 28 < T; t += 32) {
        f32x4 v[8];
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(xp + (long)(t + 4 * u) * ld);
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = wp ? wp[t + 4 * u] : 1.f;
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            const f32x4 a0 = act(v[u]), a1 = act(v[u + 1]);
            XV_POOL_STEP2(mean0, m20, n0, a0, w[u])
            XV_POOL_STEP2(mean1, m21, n1, a1, w[u + 1])
            if (wpos) { on(a0, w[u]); on(a1, w[u + 1]); }
        }
    }
    for (; t < T; t += 4) {
        const f32x4 a0 = act(*(const f32x4*)(xp + (long)t * ld));
        const float w0 = wp ? wp[t] : 1.f;
        XV_POOL_STEP2(mean0, m20, n0, a0, w0)
        if (wpos) on(a0, w0);
    }
#undef XV_POOL_STEP2
    wf_merge(mean0, m20, n0, mean1, m21, n1);
    s_mean[wave][lane] = mean0;
    s_m2[wave][lane] = m20;
    s_wp[wave][lane] = wpp;
    s_mx[wave][lane] = amx;
    if (lane == 0) s_n[wave] = n0;
    __syncthreads();
    if (wave == 0 && cv) {
        f32x4 mean = s_mean[0][lane], m2 = s_m2[0][lane];
        float n = s_n[0];
        for (int w = 1; w < 4; ++w) wf_merge(mean, m2, n, s_mean[w][lane], s_m2[w][lane], s_n[w]);
        if (wpos) *(f32x4*)(wpos + (long)b * C + col) = ((s_wp[0][lane] + s_wp[1][lane]) + (s_wp[2][lane] + s_wp[3][lane])) * (1.f / n);
        if (amax_o) {
            f32x4 mx = s_mx[0][lane];
            for (int w = 1; w < 4; ++w) {
                mx.x = fmaxf(mx.x, s_mx[w][lane].x); mx.y = fmaxf(mx.y, s_mx[w][lane].y);
                mx.z = fmaxf(mx.z, s_mx[w][lane].z); mx.w = fmaxf(mx.w, s_mx[w][lane].w);
            }
            *(f32x4*)(amax_o + (long)b * C + col) = mx;
        }
        f32x4 var = m2 * (1.f / n);
        const float eps = 1e-12f;
        f32x4 sd;
        sd.x = sqrtf(var.x <= eps ? eps : var.x); sd.y = sqrtf(var.y <= eps ? eps : var.y);
        sd.z = sqrtf(var.z <= eps ? eps : var.z); sd.w = sqrtf(var.w <= eps ? eps : var.w);
        *(f32x4*)(out + (long)b * 2 * C + col) = mean;
        *(f32x4*)(out + (long)b * 2 * C + C + col) = sd;
    }
}

extern "C" int xv_stat_pool_forward(void* stream, const float* x, int b, int t, int c, float* out) {
    XV_REQUIRE(b > 0 && t > 0 && c > 0 && c % 4 == 0, "stat_pool_forward: bad shape (c=%d must be a multiple of 4)", c);
    hipLaunchKernelGGL(stat_pool_fwd_kernel<false>, dim3(xv_cdiv(c / 4, 64), b), dim3(256), 0, (hipStream_t)stream, x, t, c,
                       (const float*)nullptr, (const float*)nullptr, 0, (const float*)nullptr, out, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                       (const int*)nullptr, 0, c);
    XV_LAUNCH_CHECK();
    return 0;
}

// wpos, amax (optional, [b][c]): see PoolGrad
int xv_stat_pool_forward_bn_ex(hipStream_t s, const float* z, int b, int t, int c, const float* scale, const float* shift, int relu,
                               const float* weights, float* out, float* wpos, float* amax, const int32_t* frames, int shrink, int ldz) {
    XV_REQUIRE(b > 0 && t > 0 && c > 0 && c % 4 == 0 && scale && shift, "stat_pool_forward_bn: bad shape (c=%d must be a multiple of 4)", c);
    if (ldz == 0) ldz = c;
    XV_REQUIRE(ldz >= c && ldz % 4 == 0, "stat_pool_forward_bn: bad row pitch %d", ldz);
    hipLaunchKernelGGL(stat_pool_fwd_kernel<true>, dim3(xv_cdiv(c / 4, 64), b), dim3(256), 0, s, z, t, c, scale, shift,
                       relu, weights, out, relu ? g_act.slope : nullptr, wpos, wpos ? amax : nullptr, (const int*)frames, shrink, ldz);
    XV_LAUNCH_CHECK();
    return 0;
}

extern "C" int xv_stat_pool_forward_bn_aux(void* stream, const float* z, int b, int t, int c, const float* scale, const float* shift, int relu,
                                           const float* weights, float* out, float* wpos, float* amax) {
    XV_REQUIRE(wpos, "stat_pool_forward_bn_aux: wpos is required");
    return xv_stat_pool_forward_bn_ex((hipStream_t)stream, z, b, t, c, scale, shift, relu, weights, out, wpos, amax);
}

extern "C" int xv_stat_pool_forward_bn(void* stream, const float* z, int b, int t, int c, const float* scale, const float* shift, int relu,
                                       const float* weights, float* out) {
    return xv_stat_pool_forward_bn_ex((hipStream_t)stream, z, b, t, c, scale, shift, relu, weights, out, nullptr, nullptr);
}

__global__ void stat_pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ out, const float* __restrict__ dout,
                                     int T, int cq, float* __restrict__ dx, long total) {
    XV_EW_PRIORITY();
    const int C = cq * 4;
    const float invT = 1.f / (float)T;
    const float sd_eps = sqrtf(1e-12f);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long row = i / cq;
        int col = (int)(i - row * cq) * 4;
        int b = (int)(row / T);
        const float* o = out + (long)b * 2 * C;
        const float* g = dout + (long)b * 2 * C;
        f32x4 mean = *(const f32x4*)(o + col), sd = *(const f32x4*)(o + C + col);
        f32x4 dm = *(const f32x4*)(g + col), ds = *(const f32x4*)(g + C + col);
        f32x4 v = *(const f32x4*)(x + row * C + col);
        f32x4 k;   // dstd * (1/std) / T, zero where the variance was clamped (pooling.py:28-29)
        k.x = sd.x <= sd_eps ? 0.f : ds.x / sd.x * invT; k.y = sd.y <= sd_eps ? 0.f : ds.y / sd.y * invT;
        k.z = sd.z <= sd_eps ? 0.f : ds.z / sd.z * invT; k.w = sd.w <= sd_eps ? 0.f : ds.w / sd.w * invT;
        *(f32x4*)(dx + row * C + col) = dm * invT + k * (v - mean);
    }
}

extern "C" int xv_stat_pool_backward(void* stream, const float* x, const float* out, const float* dout, int b, int t, int c, float* dx) {
    XV_REQUIRE(b > 0 && t > 0 && c > 0 && c % 4 == 0, "stat_pool_backward: bad shape");
    long total = (long)b * t * (c / 4);
    hipLaunchKernelGGL(stat_pool_bwd_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, out, dout, t,
                       c / 4, dx, total);
    XV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------
// l2_scaling (common.py:45-58): one wave per row
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void l2_scaling_fwd_kernel(const float* __restrict__ x, int rows, int n, float factor, float* __restrict__ y) {
    XV_EW_PRIORITY();
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long)row * n;
    float ss = 0.f;
    for (int c = lane; c < n; c += 64) ss += xr[c] * xr[c];
    ss = wave_sum(ss);
    float inv = rsqrtf(fmaxf(ss, 1e-12f)) * factor;
    for (int c = lane; c < n; c += 64) y[(long)row * n + c] = xr[c] * inv;
}
__global__ void l2_scaling_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int rows, int n, float factor,
                                      float* __restrict__ dx) {
    XV_EW_PRIORITY();
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long)row * n;
    const float* gr = dy + (long)row * n;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < n; c += 64) { ss += xr[c] * xr[c]; dot += xr[c] * gr[c]; }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    float inv = rsqrtf(fmaxf(ss, 1e-12f)) * factor;
    float k = ss >= 1e-12f ? inv / ss * dot : 0.f;
    for (int c = lane; c < n; c += 64) dx[(long)row * n + c] = gr[c] * inv - xr[c] * k;
}
extern "C" int xv_l2_scaling_forward(void* stream, const float* x, int rows, int n, float factor, float* y) {
    XV_REQUIRE(rows > 0 && n > 0, "l2_scaling_forward: bad shape");
    hipLaunchKernelGGL(l2_scaling_fwd_kernel, dim3(xv_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, n, factor, y);
    XV_LAUNCH_CHECK();
    return 0;
}
extern "C" int xv_l2_scaling_backward(void* stream, const float* x, const float* dy, int rows, int n, float factor, float* dx) {
    XV_REQUIRE(rows > 0 && n > 0, "l2_scaling_backward: bad shape");
    hipLaunchKernelGGL(l2_scaling_bwd_kernel, dim3(xv_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, dy, rows, n, factor, dx);
    XV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------
// scalar reductions (reporting / clip-by-global-norm only) and optimisers
// ------------------------------------------------------------------------------------
__global__ void sumsq_kernel(const float* __restrict__ w, size_t count, float scale, float* __restrict__ out) {
    XV_EW_PRIORITY();
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) s += w[i] * w[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, scale * ((red[0] + red[1]) + (red[2] + red[3])));
}
extern "C" int xv_l2_reg_loss(void* stream, const float* w, size_t count, float scale, float* out_accum) {
    XV_REQUIRE(count > 0, "l2_reg_loss: empty");
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for((long)count, 256, 512)), dim3(256), 0, (hipStream_t)stream, w, count, 0.5f * scale, out_accum);
    XV_LAUNCH_CHECK();
    return 0;
}
extern "C" int xv_sumsq(void* stream, const float* g, size_t count, float* out_accum) {
    XV_REQUIRE(count > 0, "sumsq: empty");
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for((long)count, 256, 512)), dim3(256), 0, (hipStream_t)stream, g, count, 1.0f, out_accum);
    XV_LAUNCH_CHECK();
    return 0;
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, size_t count, float lr, float gs) {
    XV_EW_PRIORITY();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        p[i] = p[i] - lr * (g[i] * gs);
}
__global__ void momentum_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ acc, size_t count, float lr,
                                float mom, int nesterov, float gs) {
    XV_EW_PRIORITY();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        float gi = g[i] * gs;
        float a = mom * acc[i] + gi;
        acc[i] = a;
        p[i] = nesterov ? p[i] - lr * (gi + mom * a) : p[i] - lr * a;
    }
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t count, float lr_t, float b1, float b2, float eps, float gs) {
    XV_EW_PRIORITY();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        float gi = g[i] * gs;
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}
extern "C" int xv_sgd_update(void* stream, float* p, const float* g, size_t count, float lr, float grad_scale) {
    XV_REQUIRE(count > 0, "sgd_update: empty");
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for((long)count, 256, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, count, lr, grad_scale);
    XV_LAUNCH_CHECK();
    return 0;
}
extern "C" int xv_momentum_update(void* stream, float* p, const float* g, float* acc, size_t count, float lr, float momentum,
                                  int nesterov, float grad_scale) {
    XV_REQUIRE(count > 0, "momentum_update: empty");
    hipLaunchKernelGGL(momentum_kernel, dim3(grid_for((long)count, 256, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, acc, count, lr,
                       momentum, nesterov, grad_scale);
    XV_LAUNCH_CHECK();
    return 0;
}
extern "C" int xv_adam_update(void* stream, float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1,
                              float beta2, float eps, int t, float grad_scale) {
    XV_REQUIRE(count > 0 && t >= 1, "adam_update: bad arguments");
    double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for((long)count, 256, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, count,
                       (float)lr_t, beta1, beta2, eps, grad_scale);
    XV_LAUNCH_CHECK();
    return 0;
}
