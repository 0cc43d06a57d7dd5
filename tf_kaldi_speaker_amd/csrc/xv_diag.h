// Diagnostics hooks of the fp32 GEMM kernels (xv_gemm.hip).  In the product build every hook is empty: no stamp executes, no buffer
// exists.  -DXV_DIAG (tools/variant.sh unit builds such a library next to the product, tools/gemm_probe.cpp and tools/step_clock.py read it)
// compiles per-workgroup s_memtime / s_memrealtime stamps and the hardware placement of every workgroup of the NT kernels in:
//   [wg][8] = {entry, loop start, loop end, exit, HW_ID | XCC_ID << 32, realtime at entry, realtime at exit, cycles wave 0 spent in the
//   per-K-step wait + barrier (XV_DIAG >= 2)}
// kept in two halves that alternate per stamped launch.  XV_DIAG_M / _N / _K in the environment restrict the stamps to launches of that
// problem size (e.g. tdnn2's forward pass inside a full training step); the stamp values go to a buffer nothing else reads.
// This is the only place the GEMM translation unit branches on a build flag.
#pragma once
#include "xv_common.h"

#ifdef XV_DIAG
#define XV_DBG_STAMP_WGS 4096
__device__ unsigned long long xv_dbg_stamps[2 * XV_DBG_STAMP_WGS * 8];
static int g_stamp_half = 0;
static int xv_diag_read(void* dst, size_t bytes, int half) {
    XV_REQUIRE(bytes <= sizeof(unsigned long long) * XV_DBG_STAMP_WGS * 8, "debug_read_stamps: at most %d workgroups", XV_DBG_STAMP_WGS);
    XV_CHECK_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(xv_dbg_stamps), bytes, sizeof(unsigned long long) * XV_DBG_STAMP_WGS * 8 * half));
    return 0;
}
extern "C" int xv_debug_read_stamps(void* dst, size_t bytes) { return xv_diag_read(dst, bytes, g_stamp_half ^ 1); }        // the last stamped launch
extern "C" int xv_debug_read_stamps_prev(void* dst, size_t bytes) { return xv_diag_read(dst, bytes, g_stamp_half); }    // the one before it
// host side: the half this launch stamps into, or -1 (not stamped)
static int xv_diag_half(int M, int N, int K) {
    static const int fm = getenv("XV_DIAG_M") ? atoi(getenv("XV_DIAG_M")) : 0, fn = getenv("XV_DIAG_N") ? atoi(getenv("XV_DIAG_N")) : 0,
                     fk = getenv("XV_DIAG_K") ? atoi(getenv("XV_DIAG_K")) : 0;
    if ((fm && fm != M) || (fn && fn != N) || (fk && fk != K)) return -1;
    const int h = g_stamp_half;
    g_stamp_half ^= 1;
    return h;
}
#define XV_DIAG_SLOT(half, s) xv_dbg_stamps[((half) * XV_DBG_STAMP_WGS + blockIdx.x) * 8 + (s)]
#define XV_DIAG_ON(half) ((half) >= 0 && threadIdx.x == 0 && blockIdx.x < XV_DBG_STAMP_WGS)
#define XV_STAMP(half, s) do { if (XV_DIAG_ON(half)) XV_DIAG_SLOT(half, s) = __builtin_amdgcn_s_memtime(); } while (0)
#define XV_STAMP_ENTRY(half) do { XV_STAMP(half, 0); if (XV_DIAG_ON(half)) { \
        XV_DIAG_SLOT(half, 4) = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32); \
        XV_DIAG_SLOT(half, 5) = __builtin_amdgcn_s_memrealtime(); XV_DIAG_SLOT(half, 1) = 0; } } while (0)
#define XV_STAMP_EXIT(half, stall) do { __builtin_amdgcn_s_waitcnt(0); XV_STAMP(half, 3); if (XV_DIAG_ON(half)) { \
        XV_DIAG_SLOT(half, 6) = __builtin_amdgcn_s_memrealtime(); XV_DIAG_SLOT(half, 7) = (stall); } } while (0)
#define XV_STAMP_ONCE(half, s, flag) do { if (flag) { XV_STAMP(half, s); flag = false; } } while (0)
// the per-K-step barrier, timed (XV_DIAG >= 2) or plain
#define XV_DIAG_BARRIER(stall) do { if (XV_DIAG >= 2) { const unsigned long long tw_ = __builtin_amdgcn_s_memtime(); __syncthreads(); \
        stall += __builtin_amdgcn_s_memtime() - tw_; } else __syncthreads(); } while (0)
#else
static inline int xv_diag_half(int, int, int) { return -1; }
#define XV_STAMP(half, s) ((void)0)
#define XV_STAMP_ENTRY(half) ((void)0)
#define XV_STAMP_EXIT(half, stall) ((void)0)
#define XV_STAMP_ONCE(half, s, flag) ((void)0)
#define XV_DIAG_BARRIER(stall) __syncthreads()
#endif
