// fp32 MFMA GEMM kernels for the TDNN hot path (gfx950 / CDNA4 only).
//
// Two shapes cover every contraction of tdnn.py / loss.py:
//
//  NT  C[m][n] = sum_k A[rowmap(m)][k] * Bt[n][k]
//      forward conv/dense (A = spliced activation view, Bt = kernel^T), data gradients
//      (A = zero-padded dz view, Bt = tap-flipped kernel), logits.
//  TN  P[m][n] = sum_r A[amap(r)][m] * B[bmap(r)][n]
//      weight gradients (reduction over the ~25k (chunk,frame) rows), split over r.
//
// The "spliced view" makes the context window free: row (b,t) of a layer with context k is
// the contiguous span x[b][t..t+k-1][:], so rowmap(m) = (m / rps) * pitch + (m % rps) with
// leading dimension C < K.  Rows overlap in memory; the 128-row A tile of one workgroup
// therefore re-reads the same (128+k-1) x C window k times from L2, never from HBM.
//
// Both kernels: 128x128 output tile, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile
// as 2x2 v_mfma_f32_32x32x2_f32 accumulators (64 acc VGPRs), K-step XV_TILE_K = 16, operand tiles
// staged by LDS-DMA (global_load_lds_dwordx4, scalar base + 32-bit lane offsets: xv_dma16 in xv_common.h - the address
// form decides what a stage costs the MFMA pipe) into a double-buffered 32 KB LDS image, one barrier per K-step,
// XV_WGS_PER_CU = 4 workgroups per CU (xv_common.h).
//
// MFMA operand maps (cdna_hip_programming.md section 3): lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; D register r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), column l&31.
// The contraction is order-insensitive in k as long as A and B agree, so the NT kernel lets
// lane-half h own 4 CONSECUTIVE k of every 8 (one ds_read_b128 feeds 4 MFMAs).
#include "xv_common.h"
#include <type_traits>
#include "xv_epilogue.h"
#include "xv_diag.h"
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#define BM XV_TILE_M
#define BN XV_TILE_N
#define BK XV_TILE_K
// NT LDS image: [128 rows][BK floats], no padding, 16-byte chunk c of row r stored at chunk
// c ^ f(r) with f(r) = (r >> NT_SW_SHIFT) & (NT_KQ-1).  ds_read_b128 (banks = 16 slots of 16 B,
// 16-lane groups reading 16 different rows at one chunk index) sees 16 distinct slots, and
// ds_write_b128 (8-lane groups = 2 rows x 4 chunks at BK=16) sees 8 distinct slots: the padded
// image of the first build had 2-way write conflicts at BK=16 (SQ_LDS_BANK_CONFLICT, profiles/).
#define NT_PITCH BK
#define NT_SW_SHIFT (BK == 16 ? 2 : 1)
#define NT_SWZ(row) ((((row) >> NT_SW_SHIFT) & (BK / 4 - 1)))
static_assert(BK == 16 || BK == 32, "K-step must be 16 or 32");
#define NT_KQ (BK / 4)               // float4 per tile row
#define NT_RPT (BM * NT_KQ / 256)    // tile rows staged per thread (2 or 4)
#define TN_RPT (BK / 8)              // reduction rows staged per thread (2 or 4)

struct NTArgs {
    const float* A; long lda; int a_rps; int a_pitch;
    const float* Bt; long ldb;
    float* C; long ldc; long c_split_stride;
    int M, N, K, k_chunk;
    int tiles_m, tiles_n;
    const float* bias;
    float* part_sum; float* part_m2;
    const float* zero;
    int stamp_half;      // diagnostics build only (xv_diag.h): the stamp buffer half of this launch, -1 = none
    // "whole tiles + shares" (xv_launch_gemm_nt): blocks [0, n_whole) take one whole tile each; the tiles behind them are cut into `shares`
    // equal K ranges, one block each (grid = n_whole + (tiles - n_whole) * shares); shares <= 1: every block a whole tile
    int n_whole, shares;
    float* slab;                // [(tiles - n_whole) * shares][128 * 128]: partial tiles in lane order
    unsigned* tickets;          // one per tile, zero between launches
    int row_store;              // epilogue through LDS with 16-byte row stores (nt_store_tile_rows): C / ldc 16-byte aligned, N % 4 == 0
};

// Out-of-range rows / k read this 16-byte zero page instead of being masked after the load: the
// select is on the ADDRESS, so every global_load is unconditional and hipcc neither branches
// around it nor drains vmcnt right behind it (the first build waited vmcnt(0) four times per
// K-step at the top of the loop: profiles/r01_first_bench.json, 53 % of the MFMA roof).
// The page is ordinary hipMalloc'd global memory handed in through the kernel arguments (a
// __device__ constant made the selected pointer generic -> flat_load, which also counts on
// lgkmcnt and so serialised the LDS fragment reads behind the global loads).
// The page is XV_ZERO_PAGE_FLOATS long: an out-of-range ROW is redirected to it ONCE (its base pointer), after which "base + k"
// stays inside the page for every k < K <= XV_ZERO_PAGE_FLOATS - so full K-steps stage with no per-step select at all.
static float* g_zero_page = nullptr;
static size_t g_zero_page_floats = 0;
// Grows on demand (a reduction longer than the page, e.g. the gradient through a 20 000-speaker loss head: K = the speaker count).
// The old page is left alive: launches already enqueued may still read it, and a page is a few hundred KB at most.
static int ensure_zero_page(size_t floats = XV_ZERO_PAGE_FLOATS) {
    if (g_zero_page && floats <= g_zero_page_floats) return 0;
    size_t want = XV_ZERO_PAGE_FLOATS;
    while (want < floats) want *= 2;
    float* p = nullptr;
    XV_CHECK_HIP(hipMalloc((void**)&p, want * sizeof(float)));
    XV_CHECK_HIP(hipMemset(p, 0, want * sizeof(float)));
    g_zero_page = p;
    g_zero_page_floats = want;
    return 0;
}
const float* xv_zero_page(size_t floats) { return ensure_zero_page(floats) ? nullptr : g_zero_page; }

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    // blocks b and b+8 share an XCD (round-robin dispatch): hand each XCD a contiguous run of
    // tiles so the n-tiles of one m-tile (same A rows) and neighbouring m-tiles (overlapping
    // context windows) hit the same L2.  Bijective for any nwg.
    int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Epilogue of the NT kernels: acc += bias, C tile <- acc.  The bias is added in a pass of its own in front of the stores:
// written as "v = acc + bias; if (in range) C = v" per element, hipcc put the bias load's s_waitcnt vmcnt(0) inside every
// predicated store block - and stores count on vmcnt too, so each of the 64 stores of a lane waited for the previous one to
// retire (a serialised ~30k-cycle tail per tile, with every workgroup of a one-round grid in it at the same time).  Interior
// tiles store without predication.
__device__ __forceinline__ void nt_store_tile(f32x16 (&acc)[2][2], float* __restrict__ C, long ldc, const float* __restrict__ bias,
                                              int m0, int n0, int M, int N, int wr, int wc, int li, int lh) {
    asm volatile("" : "+v"(wr), "+v"(wc), "+v"(li), "+v"(lh));      // (opaque copies: see nt_store_tile_rows)
    if (bias) {
        float bias_v[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wc * 64 + b * 32 + li;
            bias_v[b] = bias[min(n, N - 1)];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] += bias_v[b];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(acc[a][b]));      // keeps hipcc from sinking the adds back into the store blocks
    }
    float* c0 = C + (long)(m0 + wr * 64 + 4 * lh) * ldc + n0 + wc * 64 + li;
    if (m0 + BM <= M && n0 + BN <= N) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) c0[(long)(a * 32 + (r & 3) + 8 * (r >> 2)) * ldc + b * 32] = acc[a][b][r];
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int n = n0 + wc * 64 + b * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < M && n < N) c0[(long)(a * 32 + (r & 3) + 8 * (r >> 2)) * ldc + b * 32] = acc[a][b][r];
                }
            }
    }
}

// The same epilogue through the (idle) staging LDS: the accumulators of one 32-row block per wave - 64 rows x 128 columns of the tile, exactly
// the 32 KB of the two stages - are written as they stand (one column per lane: 128 consecutive bytes per half wave, no bank conflict), the
// workgroup meets, and every thread reads whole-row float4 and stores 16 bytes: a half wave stores one 512-byte row segment per instruction,
// 16 store instructions per lane and tile instead of 64 four-byte ones (whose latency chain was the 10 us epilogue of a 32-K-step tile,
// exposed whenever the workgroups of a launch end together: every one-round launch).  Two passes (a = 0, 1).  Needs C and ldc 16-byte aligned
// and N a multiple of 4; the caller falls back to nt_store_tile otherwise.  Ends behind a barrier (the LDS may be reused at once).
__device__ __forceinline__ void nt_store_tile_rows(f32x16 (&acc)[2][2], float* smem, float* __restrict__ C, long ldc, const float* __restrict__ bias,
                                                   int m0, int n0, int M, int N, int tid, int wr, int wc, int li, int lh) {
    // (opaque copies: in the evenly scheduled kernel this runs inside the loop over a workgroup's tiles, and hipcc otherwise hoists the lane
    // constants below out of it - live across the K loop, they were what that kernel spilled)
    asm volatile("" : "+v"(tid), "+v"(wr), "+v"(wc), "+v"(li), "+v"(lh));
    if (bias) {
        float bias_v[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) bias_v[b] = bias[min(n0 + wc * 64 + b * 32 + li, N - 1)];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] += bias_v[b];
    }
    // image [64 rows: wr * 32 + row of the 32-row block][128 columns]
    float* wbase = smem + (wr * 32 + 4 * lh) * BN + wc * 64 + li;
    const int rq = tid >> 5, cq = (tid & 31) * 4;                // reader: rows rq, rq + 8, ... of the image, columns cq .. cq + 3
    const bool col_ok = n0 + cq < N;                              // (N % 4 == 0: all four or none)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) wbase[((r & 3) + 8 * (r >> 2)) * BN + b * 32] = acc[a][b][r];
        __syncthreads();
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *(const f32x4*)(smem + (rq + 8 * i) * BN + cq);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = rq + 8 * i;                             // image row: wr' = q / 32, row of the block = q % 32
            const int m = m0 + (q >> 5) * 64 + a * 32 + (q & 31);
            if (m < M && col_ok) *(f32x4*)(C + (long)m * ldc + n0 + cq) = v[i];
        }
        __syncthreads();
    }
}

// The evenly scheduled kernel rotates the issue priority of its waves every K-step - priority = (K-step + the wave's slot in its SIMD) mod 4 -
// so that the co-resident workgroups take turns at the top and finish together; NOT in the one-workgroup-per-tile kernel, whose multi-round
// launches want their oldest workgroups to finish first (note 1).
// the lane's index from the hardware (two vector instructions) instead of from threadIdx.x: code behind the K loop of the evenly scheduled kernel
// rebuilds its thread coordinates from this and the (scalar) wave index, so nothing derived from threadIdx.x stays live - spilled - across the loop
__device__ __forceinline__ int xv_lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ int xv_wave_slot() { return __builtin_amdgcn_s_getreg(4 | (3 << 11)) & 15; }      // HW_ID.WAVE_ID: differs between the waves of one SIMD
__device__ __forceinline__ void xv_rot_prio(int x) {
    switch (x & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// (What did and did not move this kernel's K loop: DESIGN.md Appendix B, note 2.)
template <bool STATS, bool ROWS>
__global__ __launch_bounds__(256, XV_WGS_PER_CU) __attribute__((amdgpu_num_vgpr(128))) void xv_gemm_nt_kernel(NTArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BM * NT_PITCH];      // two slots of [A | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    XV_STAMP_ENTRY(p.stamp_half);
    unsigned long long stall = 0;
    (void)stall;

    // whole tiles first: the hardware places them first (n_whole is a multiple of 256 - the same number on every CU of an empty chip), the
    // short share blocks behind them are dealt into whatever slots are free
    const bool shared = p.shares > 1 && (int)blockIdx.x >= p.n_whole;      // (uniform)
    int t, share = 0;
    if (p.shares > 1) {
        if (shared) {
            const int r = xcd_swizzle(blockIdx.x - p.n_whole, gridDim.x - p.n_whole);      // (n_whole % 8 == 0: the XCD of a block is blockIdx % 8 in both ranges)
            t = p.n_whole + r / p.shares;
            share = r - (r / p.shares) * p.shares;
        } else {
            t = xcd_swizzle(blockIdx.x, p.n_whole);
        }
    } else {
        t = xcd_swizzle(blockIdx.x, gridDim.x);
    }
    const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    int k_begin = blockIdx.z * p.k_chunk;
    int k_end = min(p.K, k_begin + p.k_chunk);
    if (shared) {      // share i of a tile: K-steps [i * nk / shares, (i + 1) * nk / shares)
        const int nk_tile = (p.K + BK - 1) / BK;
        k_begin = (int)((long)share * nk_tile / p.shares) * BK;
        k_end = min(p.K, (int)((long)(share + 1) * nk_tile / p.shares) * BK);
    }
    // (uniform by construction; said explicitly because xv_dma16 takes "operand base + k" in SGPRs - the diagnostics build otherwise keeps
    // the share arithmetic in vector registers and the assembler rejects the DMA)
    k_begin = __builtin_amdgcn_readfirstlane(k_begin);
    k_end = __builtin_amdgcn_readfirstlane(k_end);
    const int nk = (k_end - k_begin + BK - 1) / BK;

    // ---- global -> LDS staging by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write.
    // One wave-instruction writes 1 KiB linearly (wave-uniform base + lane*16 B) = NT_RPI whole tile rows,
    // so lane l lands on row RPI*(RPT*wave+i) + l/KQ at chunk POSITION l%KQ; the XOR swizzle therefore
    // moves to the SOURCE: that lane fetches chunk (l%KQ) ^ f(row) of its row (rule: swizzle source +
    // read, never the DMA destination).  Out-of-range rows / k fetch the zero page.
    constexpr int NT_RPI = 64 / NT_KQ;          // tile rows per wave-instruction (16 at BK=16)
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane / NT_KQ, lpos = lane % NT_KQ;
    // per-lane byte offsets from p.A / p.Bt with the swizzled chunk folded in (xv_dma16: scalar base + 32-bit offset).  Rows outside the
    // matrix read row 0 - their products land in accumulator rows / columns that are neither stored nor counted; only k beyond k_end must
    // read zeros (it is summed into valid outputs): the ragged last K-step takes the zero page through the builtin's 64-bit form.
    const float* __restrict__ zp = p.zero;
    unsigned aoff[NT_RPT], boff[NT_RPT];
    int ksrc[NT_RPT];
#pragma unroll
    for (int i = 0; i < NT_RPT; ++i) {
        int row = NT_RPI * (NT_RPT * wave + i) + lrow;
        ksrc[i] = ((lpos ^ NT_SWZ(row)) << 2);
        int m = m0 + row;
        int mm = m < p.M ? m : 0;
        int seg = mm / p.a_rps, tt = mm - seg * p.a_rps;
        aoff[i] = (unsigned)((((long)seg * p.a_pitch + tt) * p.lda + ksrc[i]) * 4);
        int n = n0 + row;
        boff[i] = (unsigned)(((long)(n < p.N ? n : 0) * p.ldb + ksrc[i]) * 4);
    }
    const unsigned lds0 = xv_lds_addr(smem + NT_RPI * NT_RPT * uwave * NT_PITCH);
    auto gstage = [&](int kt, int buf) {
        const int k0 = k_begin + kt * BK;
        if (k0 + BK <= k_end) {        // full K-step (uniform): scalar bases + k0, the lane offsets never change
            const float* abase = p.A + k0;
            const float* bbase = p.Bt + k0;
#pragma unroll
            for (int i = 0; i < NT_RPT; ++i) {
                xv_dma16(abase, aoff[i], lds0 + (buf * (2 * BM * NT_PITCH) + NT_RPI * i * NT_PITCH) * 4);
                xv_dma16(bbase, boff[i], lds0 + (buf * (2 * BM * NT_PITCH) + BM * NT_PITCH + NT_RPI * i * NT_PITCH) * 4);
            }
        } else {                       // ragged last step: chunks at or beyond k_end come from the zero page
            float* sa = smem + buf * (2 * BM * NT_PITCH) + NT_RPI * NT_RPT * uwave * NT_PITCH;
            float* sb = sa + BM * NT_PITCH;
#pragma unroll
            for (int i = 0; i < NT_RPT; ++i) {
                const bool kv = k0 + ksrc[i] < k_end;
                const float* pa = (const float*)((const char*)(p.A + k0) + aoff[i]);
                const float* pb = (const float*)((const char*)(p.Bt + k0) + boff[i]);
                xv_dma16_ptr((kv ? pa : zp), sa + NT_RPI * i * NT_PITCH);
                xv_dma16_ptr((kv ? pb : zp), sb + NT_RPI * i * NT_PITCH);
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

    const int a_off = (wr * 64 + li) * NT_PITCH;
    const int b_off = (wc * 64 + li) * NT_PITCH;
    const int fsw = NT_SWZ(li);     // rows wr*64 + a*32 + li share f(li): the offsets are multiples of 16
    if (nk > 0) gstage(0, 0);
    xv_dma_wait_all();
    __syncthreads();
    XV_STAMP(p.stamp_half, 1);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gstage(kt + 1, buf ^ 1);
        const float* sa = smem + buf * (2 * BM * NT_PITCH);
        const float* sb = sa + BM * NT_PITCH;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 af[2], bf[2];
            const int pos = (((2 * q + lh) ^ fsw) << 2);
            af[0] = *(const f32x4*)(sa + a_off + pos);
            af[1] = *(const f32x4*)(sa + a_off + 32 * NT_PITCH + pos);
            bf[0] = *(const f32x4*)(sb + b_off + pos);
            bf[1] = *(const f32x4*)(sb + b_off + 32 * NT_PITCH + pos);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][e], bf[0][e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][e], bf[1][e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][e], bf[0][e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][e], bf[1][e], acc[1][1], 0, 0, 0);
            }
        }
        xv_dma_wait_all();      // the compiler does not see xv_dma16's loads
        XV_DIAG_BARRIER(stall);
    }
    XV_STAMP(p.stamp_half, 2);

    bool alive = true;
    if (shared) {
        // a share of a tile: publish it (lane-order slab: 1 KB of consecutive bytes per wave instruction), take the tile's ticket, and only
        // the last of the tile's blocks goes on: it sums the shares in K order - ((s0 + s1) + s2) + ... whoever it is; the first two commute,
        // so a block that holds share 0 or 1 keeps it in registers - and runs the ordinary epilogue (xv_handoff_* contract, xv_common.h)
        int& s_last = *(int*)smem;      // (the staging buffers are idle)
        const int first = (t - p.n_whole) * p.shares;
        float* mine = p.slab + (long)(first + share) * (BM * BN) + tid * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x4 v = {acc[0][0][r], acc[0][1][r], acc[1][0][r], acc[1][1][r]};
            xv_handoff_store4(mine + r * (256 * 4), v);
        }
        xv_handoff_drain();
        __syncthreads();
        if (tid == 0) s_last = xv_ticket_take(p.tickets + t, (unsigned)p.shares) ? 1 : 0;
        __syncthreads();
        const int last = s_last;
        __syncthreads();                // (s_last aliases the scratch the statistics use below)
        alive = last != 0;              // (uniform) the other shares of the tile are done
        if (alive && share > 1) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        }
        for (int i = 0; alive && i < p.shares; ++i) {
            const int v = share == 1 ? (i == 0 ? 1 : i == 1 ? 0 : i) : i;      // share 1: own, share 0, share 2, ...
            if (v == share && share <= 1) continue;
            const float* src = p.slab + (long)(first + v) * (BM * BN) + tid * 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 x[8];
                xv_handoff_load8(src + 8 * h * (256 * 4), 256 * 4, x);
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) {
                    const int r = 8 * h + k8;
                    acc[0][0][r] += x[k8][0]; acc[0][1][r] += x[k8][1]; acc[1][0][r] += x[k8][2]; acc[1][1][r] += x[k8][3];
                }
            }
        }
    }
    // ---- epilogue
    if (alive) {
        float* C = p.C + (long)blockIdx.z * p.c_split_stride;
        if (ROWS) nt_store_tile_rows(acc, smem, C, p.ldc, p.bias, m0, n0, p.M, p.N, tid, wr, wc, li, lh);
        else nt_store_tile(acc, C, p.ldc, p.bias, m0, n0, p.M, p.N, wr, wc, li, lh);
        if (STATS) xv_tile_stats_epilogue(acc, smem, tid, wr, wc, li, lh, m0, n0, p.M, p.N, tile_m, p.tiles_m, p.part_sum);
    }
    XV_STAMP_EXIT(p.stamp_half, stall);
}

// -------------------------------------------------------------------------------------
// NT, evenly scheduled ("stream-K"): the launch has P persistent workgroups (one co-resident round) and every one of them runs the
// same number of K-steps.  The (tile, K-step) pairs are numbered tile-major, u = tile * nk + kt; workgroup w owns the units
// [w * total / P, (w + 1) * total / P) - a run that may start and end in the middle of a tile.  A tile whose K-steps were shared
// stores its partial accumulators as slabs (the lane-order layout of the TN kernel), and the workgroup that finishes its share
// LAST sums them in K order (xv_handoff_* contract) and runs the ordinary epilogue (bias, store, BatchNorm statistics).
// It exists for tile-count quantisation (784 tiles on 768 slots, 2.3 tiles per CU in the shipped batch): DESIGN.md Appendix B, note 3.
struct NTSKArgs {
    NTArgs g;
    int P;                    // workgroups (grid size)
    int nk;                   // K-steps per tile
    long total;               // tiles * nk
    float* slab;              // [P][2][128*128]: first / last partial tile of each workgroup
    unsigned* tickets;        // one per tile, zero between launches
};

__device__ __forceinline__ int ntsk_owner(long u, int P, long total) { return (int)((((u + 1) * P) - 1) / total); }

// (Its context-window form of rounds 2-4 - the rows of one 16-channel chunk of x staged once for all taps - is gone: DESIGN.md Appendix B, note 4.)
#define XV_NT_SK_WPC 3                       // workgroups per CU of the even schedule: one co-resident round of 768 (2 / 4 per CU: S1 +0.5 %, round 6)
// Register budget: 128 VGPRs (4 waves per SIMD) although the launch is 3 workgroups per CU - the free fourth slot is worth more to the other
// stream than the 7-23 spilled set-up registers cost (note 5; tools/variant.sh unit xv_gemm.hip "sk168:-DXV_NT_SK_VGPRS=168 -DXV_NT_SK_OCC=3").
#ifndef XV_NT_SK_VGPRS
#define XV_NT_SK_VGPRS 128
#define XV_NT_SK_OCC XV_WGS_PER_CU
#endif
template <bool STATS, bool ROWS>
__global__ __launch_bounds__(256, XV_NT_SK_OCC) __attribute__((amdgpu_num_vgpr(XV_NT_SK_VGPRS))) void xv_gemm_nt_sk_kernel(NTSKArgs q) {
    const NTArgs& p = q.g;
    constexpr int A_SLOT = BM * NT_PITCH;       // floats per A slot
    constexpr int B_SLOT = BM * NT_PITCH;
    __shared__ __attribute__((aligned(16))) float smem[2 * A_SLOT + 2 * B_SLOT];   // [A slot 0 | A slot 1 | B slot 0 | B slot 1]
    int& s_last = *(int*)smem;      // (the staging buffers are idle when it is used)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int w = xcd_swizzle(blockIdx.x, gridDim.x);
    // (unit indices fit 32 bits - the launcher checks total < 2^31 -; as 64-bit scalars they were five register pairs live across the whole run)
    const int u_end = (int)((long)(w + 1) * q.total / q.P);
    int u = (int)((long)w * q.total / q.P);
    const int first_tile = (int)(u / q.nk);
    XV_STAMP_ENTRY(p.stamp_half);
    bool first_seg = true;
    (void)first_seg;

    constexpr int NT_RPI = 64 / NT_KQ;          // tile rows per wave-instruction (16 at BK=16)
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane / NT_KQ, lpos = lane % NT_KQ;
    const float* __restrict__ zp = p.zero;
    const int b_off = (wc * 64 + li) * NT_PITCH;
    const int fsw = NT_SWZ(li);
    const int wslot = xv_wave_slot();
    // LDS byte addresses this wave's DMA pieces land at (slot 0; xv_dma16)
    const unsigned lds_a = xv_lds_addr(smem + NT_RPI * NT_RPT * uwave * NT_PITCH);
    const unsigned lds_b = xv_lds_addr(smem + 2 * A_SLOT + NT_RPI * NT_RPT * uwave * NT_PITCH);

    // Order of a run that ends one tile and begins the next ([k0, nk) of tile t, then [0, k2) of tile t + 1): the BEGINNING of the next tile
    // first.  Every workgroup then walks K upwards from (about) 0 in step with the others, so the K-slices of the weight matrix in flight on
    // an XCD at any moment are the same few for all its workgroups, as in a one-workgroup-per-tile launch.  In run order the workgroups of
    // an XCD sit at all K offsets at once, their weight slices do not fit the L2 together and are re-fetched from the Infinity Cache
    // (1.45 -> 0.16 GB past the L2 per launch, same time: DESIGN.md Appendix B, note 6.)
    const int u_begin = u;
    const int u_mid = (u / q.nk + 1) * q.nk;                   // end of the first tile of the run
    const bool wrap_first = u % q.nk != 0 && u_mid < u_end && u_end - u_mid <= q.nk;
    for (int pass = 0; pass < 2; ++pass) {
    int u_stop = u_end;
    if (wrap_first) { u = pass == 0 ? u_mid : u_begin; u_stop = pass == 0 ? u_end : u_mid; }
    else if (pass == 1) break;
    while (u < u_stop) {
        const int tile = (int)(u / q.nk);
        const int kt0 = u - tile * q.nk;
        const int kt1 = min(q.nk, kt0 + (u_stop - u));
        const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
        const int m0 = tile_m * BM, n0 = tile_n * BN;

        // per-lane byte offsets of this tile from p.A / p.Bt (swizzled chunk folded in; xv_dma16).  Rows outside the operand read row 0:
        // their products only reach accumulator rows / columns that are never stored, counted or - in a shared tile - used after the sum
        unsigned aoff[NT_RPT];
        unsigned boff[NT_RPT];
        int ksrc[NT_RPT];
        int a_row[2];                           // LDS offsets of this lane's two fragment rows
        // (the ragged last K-step of a row needs each piece's k offset again: rebuilt from the hardware lane index there, not kept in a register)
        auto ksrc_now = [&](int i) {
            const int l = xv_lane_id();
            return (((l % NT_KQ) ^ NT_SWZ(NT_RPI * (NT_RPT * uwave + i) + l / NT_KQ)) << 2);
        };
#pragma unroll
        for (int i = 0; i < NT_RPT; ++i) {
            const int row = NT_RPI * (NT_RPT * wave + i) + lrow;
            ksrc[i] = ((lpos ^ NT_SWZ(row)) << 2);
            const int n = n0 + row;
            boff[i] = (unsigned)(((long)(n < p.N ? n : 0) * p.ldb + ksrc[i]) * 4);
        }
        {
#pragma unroll
            for (int i = 0; i < NT_RPT; ++i) {
                const int row = NT_RPI * (NT_RPT * wave + i) + lrow;
                const int m = m0 + row;
                const int mm = m < p.M ? m : 0;
                const int seg = mm / p.a_rps, tt = mm - seg * p.a_rps;
                aoff[i] = (unsigned)((((long)seg * p.a_pitch + tt) * p.lda + ksrc[i]) * 4);
            }
            a_row[0] = (wr * 64 + li) * NT_PITCH;
            a_row[1] = a_row[0] + 32 * NT_PITCH;
        }
        // K-step kt of a tile = columns [kt*BK, +BK) of the spliced row
        auto stage_b = [&](int kt, int slot) {
            const int k0 = kt * BK;
            if (k0 + BK <= p.K) {
                const float* bbase = p.Bt + k0;
#pragma unroll
                for (int i = 0; i < NT_RPT; ++i) xv_dma16(bbase, boff[i], lds_b + (slot * B_SLOT + NT_RPI * i * NT_PITCH) * 4);
            } else {                       // ragged last step of a row: chunks at or beyond K come from the zero page
                float* sb = smem + 2 * A_SLOT + slot * B_SLOT + NT_RPI * NT_RPT * uwave * NT_PITCH;
#pragma unroll
                for (int i = 0; i < NT_RPT; ++i) {
                    const float* pb = (const float*)((const char*)(p.Bt + k0) + boff[i]);
                    xv_dma16_ptr((k0 + ksrc_now(i) < p.K ? pb : zp), sb + NT_RPI * i * NT_PITCH);
                }
            }
        };
        auto stage_a = [&](int kt, int slot) {      // the tile's rows of K-step kt
            const int k0 = kt * BK;
            if (k0 + BK <= p.K) {          // full K-step (uniform): scalar base + k0, the lane offsets never change
                const float* abase = p.A + k0;
#pragma unroll
                for (int i = 0; i < NT_RPT; ++i) xv_dma16(abase, aoff[i], lds_a + (slot * A_SLOT + NT_RPI * i * NT_PITCH) * 4);
            } else {
                float* sa = smem + slot * A_SLOT + NT_RPI * NT_RPT * uwave * NT_PITCH;
#pragma unroll
                for (int i = 0; i < NT_RPT; ++i) {
                    const float* pa = (const float*)((const char*)(p.A + k0) + aoff[i]);
                    xv_dma16_ptr((k0 + ksrc_now(i) < p.K ? pa : zp), sa + NT_RPI * i * NT_PITCH);
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

        stage_a(kt0, 0);
        stage_b(kt0, 0);
        xv_dma_wait_all();      // (the compiler does not see xv_dma16's loads)
        __syncthreads();
        XV_STAMP_ONCE(p.stamp_half, 1, first_seg);
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            xv_rot_prio(kt + wslot);
            if (kt + 1 < kt1) {
                stage_a(kt + 1, buf ^ 1);
                stage_b(kt + 1, buf ^ 1);
            }
            const float* sa = smem + buf * A_SLOT;
            const float* sb = smem + 2 * A_SLOT + buf * B_SLOT;
#pragma unroll
            for (int qq = 0; qq < BK / 8; ++qq) {
                f32x4 af[2], bf[2];
                const int pos = (((2 * qq + lh) ^ fsw) << 2);
                af[0] = *(const f32x4*)(sa + a_row[0] + pos);
                af[1] = *(const f32x4*)(sa + a_row[1] + pos);
                bf[0] = *(const f32x4*)(sb + b_off + pos);
                bf[1] = *(const f32x4*)(sb + b_off + 32 * NT_PITCH + pos);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][e], bf[0][e], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][e], bf[1][e], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][e], bf[0][e], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][e], bf[1][e], acc[1][1], 0, 0, 0);
                }
            }
            xv_dma_wait_all();
            __syncthreads();
        }
        u += kt1 - kt0;
        __builtin_amdgcn_s_setprio(0);
        XV_STAMP(p.stamp_half, 2);

        if (kt0 != 0 || kt1 != q.nk) {
            // a shared tile: publish this share, take a ticket, and only the last of the tile's workgroups goes on
            // slab of a share: [register r][thread] float4 = (acc[0][0][r], acc[0][1][r], acc[1][0][r], acc[1][1][r]): 1 KB of consecutive
            // bytes per wave instruction, stores and loads alike
            const int tid_h = uwave * 64 + xv_lane_id();      // (rebuilt: keeps the slab addresses out of the registers that live across the K loop)
            float* mine = q.slab + ((long)w * 2 + (tile == first_tile ? 0 : 1)) * (BM * BN) + tid_h * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x4 v = {acc[0][0][r], acc[0][1][r], acc[1][0][r], acc[1][1][r]};
                xv_handoff_store4(mine + r * (256 * 4), v);
            }
            xv_handoff_drain();
            __syncthreads();
            const long t_first = (long)tile * q.nk;
            const int w_first = ntsk_owner(t_first, q.P, q.total), w_last = ntsk_owner(t_first + q.nk - 1, q.P, q.total);
            if (tid == 0) s_last = xv_ticket_take(q.tickets + tile, (unsigned)(w_last - w_first + 1)) ? 1 : 0;
            __syncthreads();
            const int last = s_last;                     // s_last aliases the staging buffers: every wave has read it before the barrier
            __syncthreads();                             // below lets anyone stage the next share or use the statistics scratch
            if (!last) continue;                         // (uniform)
            // The tile's value is ((s0 + s1) + s2) + ... in K order whoever arrives last.  The first two shares commute, so a workgroup that
            // holds share 0 or 1 keeps it in its registers and reads only the others (one slab for the usual two-share tile); a later
            // share is re-read from its slab at its place in the order.
            const int me = w - w_first;
            if (me > 1) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
            }
            for (int i = 0; i <= w_last - w_first; ++i) {
                const int v = w_first + (me == 1 ? (i == 0 ? 1 : i == 1 ? 0 : i) : i);      // me == 1: own, share 0, share 2, ...
                if (v == w && me <= 1) continue;
                const int v_first_tile = (int)(((long)v * q.total / q.P) / q.nk);
                const float* src = q.slab + ((long)v * 2 + (tile == v_first_tile ? 0 : 1)) * (BM * BN) + tid_h * 4;
                // eight 16-byte loads in flight per lane: a shared tile end is a latency chain (four in flight: no scratch, step +1.5 %; DESIGN.md Appendix B, note 7)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 x[8];
                    xv_handoff_load8(src + 8 * h * (256 * 4), 256 * 4, x);
#pragma unroll
                    for (int k8 = 0; k8 < 8; ++k8) {
                        const int r = 8 * h + k8;
                        acc[0][0][r] += x[k8][0]; acc[0][1][r] += x[k8][1]; acc[1][0][r] += x[k8][2]; acc[1][1][r] += x[k8][3];
                    }
                }
            }
        }
        const int lane_e = xv_lane_id(), tid_e = uwave * 64 + lane_e, wr_e = uwave >> 1, wc_e = uwave & 1, li_e = lane_e & 31, lh_e = lane_e >> 5;
        if (ROWS) nt_store_tile_rows(acc, smem, p.C, p.ldc, p.bias, m0, n0, p.M, p.N, tid_e, wr_e, wc_e, li_e, lh_e);
        else nt_store_tile(acc, p.C, p.ldc, p.bias, m0, n0, p.M, p.N, wr_e, wc_e, li_e, lh_e);
        if (STATS) {
            xv_tile_stats_epilogue(acc, smem, tid_e, wr_e, wc_e, li_e, lh_e, m0, n0, p.M, p.N, tile_m, p.tiles_m, p.part_sum);
            __syncthreads();                             // the statistics use the staging buffers as scratch
        }
    }
    }
    XV_STAMP_EXIT(p.stamp_half, 0);
}

// out[m][n] = sum_z slab[z][m][n] (+ bias[n])
__global__ void xv_splitk_reduce_kernel(const float* __restrict__ slab, int splits, long split_stride, int M, int N,
                                        int lds, const float* __restrict__ bias, float* __restrict__ out, long ldo) {
    XV_EW_PRIORITY();
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)M * N;
    for (; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int m = (int)(idx / N), n = (int)(idx - (long)m * N);
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += slab[z * split_stride + (long)m * lds + n];
        if (bias) v += bias[n];
        out[(long)m * ldo + n] = v;
    }
}

// ---- live launch timing (bench.py roofline leg) ----------------------------------------
namespace {
struct ProfRec { hipEvent_t a, b; int kind; double flops; };
bool g_prof_on = false;
uint32_t g_prof_mask = ~0u;
std::vector<ProfRec> g_prof;
size_t g_prof_cap = 0;
std::vector<hipEvent_t> g_prof_events;   // pool, two per record slot
}  // namespace

XvProfScope::XvProfScope(hipStream_t st, int kind, double flops) : s(st), idx(-1) {
    if (!g_prof_on || !((g_prof_mask >> kind) & 1u) || g_prof.size() >= g_prof_cap) return;
    idx = (int)g_prof.size();
    ProfRec r = {g_prof_events[2 * idx], g_prof_events[2 * idx + 1], kind, flops};
    g_prof.push_back(r);
    (void)hipEventRecord(r.a, s);
}
XvProfScope::~XvProfScope() { if (idx >= 0) (void)hipEventRecord(g_prof[idx].b, s); }

extern "C" int xv_profile_begin(int max_launches) { return xv_profile_begin_kinds(max_launches, ~0u); }

extern "C" int xv_profile_begin_kinds(int max_launches, uint32_t kind_mask) {
    XV_REQUIRE(max_launches > 0, "profile_begin: max_launches must be positive");
    g_prof_mask = kind_mask;
    while (g_prof_events.size() < (size_t)2 * max_launches) {
        hipEvent_t ev;
        // device-scope release: the pair brackets one launch on its own stream; a system-scope fence per record (the default) would be
        // measurement overhead inside the region bench.py times
        XV_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventReleaseToDevice));
        g_prof_events.push_back(ev);
    }
    g_prof.clear();
    g_prof_cap = (size_t)max_launches;
    g_prof_on = true;
    return 0;
}

extern "C" int xv_profile_end(int64_t launches[XV_PROFILE_KINDS], double ms[XV_PROFILE_KINDS], double flops[XV_PROFILE_KINDS]) {
    g_prof_on = false;
    for (int k = 0; k < XV_PROFILE_KINDS; ++k) { launches[k] = 0; ms[k] = 0.0; flops[k] = 0.0; }
    for (auto& r : g_prof) {
        XV_CHECK_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        XV_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
        launches[r.kind] += 1; ms[r.kind] += (double)t; flops[r.kind] += r.flops;
    }
    g_prof.clear();
    return 0;
}

// Tickets of the split hand-over: one zeroed uint32 per output tile, re-armed by the workgroup that takes the last one.  One array per
// STREAM: launches on a stream run in order, so the array is clean again before the next launch on it reads it, and launches on
// different streams (the engine runs weight gradients on two side streams) never share one.  Shared by the TN kernel and the
// evenly scheduled NT kernel: both leave every ticket at zero.
#define XV_TN_MAX_TILES 16384
static unsigned* tn_tickets_for(hipStream_t s) {
    static std::mutex mu;
    static std::map<hipStream_t, unsigned*> pool;
    std::lock_guard<std::mutex> lock(mu);
    auto it = pool.find(s);
    if (it != pool.end()) return it->second;
    unsigned* t = nullptr;
    if (hipMalloc((void**)&t, XV_TN_MAX_TILES * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(t, 0, XV_TN_MAX_TILES * sizeof(unsigned)) != hipSuccess) return nullptr;      // (synchronous: visible to every stream)
    pool[s] = t;
    return t;
}

// "whole tiles + shares" (xv_launch_gemm_nt): shares per remaining tile, 0 = the schedule does not apply or does not pay.
// K-steps on the busiest CU = the whole tiles + ceil(rem * S / 256) shares of nk / S K-steps and ~10 K-steps of hand-over each; S is chosen
// so that the shares come out as (just under) a whole number per CU - 72 remaining tiles: 3 shares each = 216, at most one per CU, not 4 (288:
// two on some CUs) - and small on a tie (the tile's last share sums S slabs alone).  tests/test_streamk_schedule.py restates it.
int xv_nt_shares(int tiles, int ksteps, bool stats, bool beside_wgrad, size_t ws_bytes) {
    const int rem_tiles = tiles % 256, whole_per_cu = tiles / 256;
    if (rem_tiles < 1 || rem_tiles > 128) return 0;
    if (!(whole_per_cu >= 2 || (whole_per_cu == 1 && stats)) || (beside_wgrad && whole_per_cu == 3)) return 0;
    long best = 0;
    int best_s = 0;
    for (int sh = 2; sh <= 16 && ksteps / sh >= 6; ++sh) {
        const long t = (long)whole_per_cu * ksteps + (long)xv_cdiv((long)rem_tiles * sh, 256) * (ksteps / sh + 10) + sh;
        if (!best_s || t < best) { best = t; best_s = sh; }
    }
    const long t_dp = (long)xv_cdiv(tiles, 256) * ksteps;
    if (!best_s || best + best / 32 >= t_dp || (size_t)rem_tiles * best_s * BM * BN * sizeof(float) > ws_bytes) return 0;
    return best_s;
}

// Which schedule an NT launch runs (DESIGN.md section 4 "fp32 GEMM design" states the three and where each wins; the measurements behind every
// constant here are DESIGN.md Appendix A, rounds 3-4, and profiles/r04_nt_whole_plus_shares.txt / r04_ab_variants.txt):
//   DP      one workgroup per tile, dealt to the CUs by the hardware: cost ceil(tiles / 256) K-loops on the busiest CU
//   SK      one co-resident round of <= 768 persistent workgroups with equal runs of K-steps, tiles shared through slabs: total / 256 K-steps
//           + ~15 K-steps of hand-over per workgroup; its K loop is 4-8 % slower than DP's, so it only wins on tile-count quantisation
//   SHARES  DP for the first floor(tiles / 256) * 256 tiles, every remaining tile cut into `shares` K ranges behind them (xv_nt_shares)
//   SPLIT   few tiles and no BatchNorm statistics: split-K over the whole chip + a slab-sum launch (extraction one utterance at a time,
//           segment-level layers of batches > 128 chunks)
enum { XV_NT_DP = 0, XV_NT_SK = 1, XV_NT_SHARES = 2, XV_NT_SPLIT = 3 };
struct XvNtPlan { int kind, p_sk, shares, splits; };
static XvNtPlan xv_nt_plan(int M, int N, int K, bool stats, bool co_running, bool have_ws, size_t ws_bytes, int forced /* XvEnv::nt_sched */) {
    const int tiles = xv_cdiv(M, BM) * xv_cdiv(N, BN), ksteps = xv_cdiv(K, BK);
    XvNtPlan pl = {XV_NT_DP, 1, 0, 1};
    const long total = (long)tiles * ksteps;
    const long p_sk = std::min<long>(std::min<long>(256L * XV_NT_SK_WPC, std::max<long>(1, total / 4)), 8L * tiles);
    const long t_sk = (total / p_sk) * xv_cdiv(p_sk, 256) + (15 * 16 / BK) * std::min<long>(XV_NT_SK_WPC, xv_cdiv(p_sk, 256));
    const long t_dp = tiles <= XV_RESIDENT_WGS ? (long)xv_cdiv(tiles, 256) * ksteps : total / 256 + ksteps / 2;
    bool sk = forced ? forced == 2 : t_sk + t_sk / 32 < t_dp;
    // Beside the weight-gradient stream (the data gradients) a launch with at least two whole tiles per CU is dealt one workgroup per tile: its
    // workgroups are never all resident at once there anyway (the other stream's kernel holds slots), so a partial last round is not the
    // cost `t_dp` prices, while the even schedule's slower K loop and hand-overs are.  [measured, round 6, same box, 6 alternated rounds,
    // profiles/r06_nt_co_running.txt] S1 (tdnn2's data gradient: 784 tiles, until now on the even schedule) -0.25 ... -0.4 %, S4 -0.4 %,
    // 64 x U{200..400} / S2 / S5 +-0.1 %; for every tile count -0.4 / -0.1 / +0.2 %, from three whole tiles per CU -0.3 / -0.3 / +0.1 %.
    if (!forced && co_running && tiles >= 512) sk = false;
    // ... and "whole tiles + shares" there only for long tiles (>= 100 K-steps): a share's hand-over (slab store, ticket, the last share sums the
    // others) is ~10 K-steps of latency whatever the tile's length, and beside the other stream's GEMM the partial round it avoids costs little.
    // [measured, round 6, same box, 4 alternated rounds, profiles/r06_nt_co_running.txt] S5 (ten layers of 32 / 96 K-step tiles at 128 x 400)
    // -1.0 %, every other shape within +-0.2 %; no shares at all beside the stream: S5 -1.2 %, 64 x U{200..400} +0.7 %, S2 +0.3 %.
    const int shares = forced || !have_ws || tiles > XV_TN_MAX_TILES || (co_running && ksteps < 100) ? 0 : xv_nt_shares(tiles, ksteps, stats, co_running, ws_bytes);
    if (shares) sk = false;
    const bool few = !stats && tiles < 192 && ksteps >= 8 && !forced;
    if (!few && sk) {
        const bool shared_tiles = total % p_sk != 0 || (total / p_sk) % ksteps != 0;
        if (!shared_tiles || ((size_t)p_sk * 2 * BM * BN * sizeof(float) <= ws_bytes && have_ws && tiles <= XV_TN_MAX_TILES)) {
            pl.kind = XV_NT_SK; pl.p_sk = (int)p_sk;
            return pl;
        }
    }
    int splits = 1;
    if (!stats && tiles < XV_RESIDENT_WGS / 2 && ksteps >= 8) {
        splits = std::max(1, std::min(XV_RESIDENT_WGS / tiles, ksteps / 4));
        const long np = (long)xv_align(N, 4);
        // the slabs are written and read back: beyond ~8 MB the reduce costs more than the extra workgroups buy
        const long slab_cap = std::max<long>(8, (8L << 20) / ((long)M * np * (long)sizeof(float)));
        if (splits > slab_cap) splits = (int)slab_cap;
        while (splits > 1 && (size_t)splits * M * np * sizeof(float) > ws_bytes) --splits;
    }
    if (splits > 1) { pl.kind = XV_NT_SPLIT; pl.splits = splits; }
    else if (shares) { pl.kind = XV_NT_SHARES; pl.shares = shares; }
    return pl;
}
// diagnostics (tools/pmc_traffic.py maps S1's layers to the kernels that ran them with this): the schedule of an NT problem given ample workspace
extern "C" int xv_debug_nt_schedule(int M, int N, int K, int stats, int co_running) {
    const XvEnv* env = xv_env();
    return xv_nt_plan(M, N, K, stats != 0, co_running != 0, true, (size_t)1 << 32, env ? env->nt_sched : 0).kind;
}

int xv_launch_gemm_nt(hipStream_t s, const XvGemmNT& g) {
    XV_REQUIRE(g.K % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0, "gemm_nt: K/lda/ldb must be multiples of 4 (K=%d lda=%ld ldb=%ld)", g.K, g.lda, g.ldb);
    XV_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.Bt % 16) == 0, "gemm_nt: operands must be 16-byte aligned");
    XV_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.a_rps > 0, "gemm_nt: empty problem");
    // xv_dma16 addresses every operand row as a 32-bit byte offset from the operand's base
    XV_REQUIRE(((long)xv_cdiv(g.M, g.a_rps) * g.a_pitch + 1) * g.lda * 4 < (1L << 32) && ((long)g.N + 1) * g.ldb * 4 < (1L << 32),
               "gemm_nt: an operand spans 4 GB or more (M=%d a_pitch=%d lda=%ld N=%d ldb=%ld): split the batch", g.M, g.a_pitch, g.lda, g.N, g.ldb);
    if (ensure_zero_page((size_t)g.K)) return 1;
    NTArgs p;
    p.zero = g_zero_page;
    p.stamp_half = xv_diag_half(g.M, g.N, g.K);
    p.A = g.A; p.lda = g.lda; p.a_rps = g.a_rps; p.a_pitch = g.a_pitch;
    p.Bt = g.Bt; p.ldb = g.ldb;
    p.M = g.M; p.N = g.N; p.K = g.K;
    p.tiles_m = xv_cdiv(g.M, BM); p.tiles_n = xv_cdiv(g.N, BN);
    p.bias = g.bias; p.part_sum = nullptr; p.part_m2 = nullptr;
    p.n_whole = 0; p.shares = 0; p.slab = nullptr; p.tickets = nullptr;
#ifdef XV_NT_NO_ROWS      // (A/B build constant: the four-byte-store epilogue everywhere)
    p.row_store = 0;
#else
    p.row_store = (g.N % 4 == 0 && g.ldc % 4 == 0 && ((uintptr_t)g.C % 16) == 0) ? 1 : 0;
#endif
    const int tiles = p.tiles_m * p.tiles_n;
    const int ksteps = xv_cdiv(g.K, BK);
    const XvEnv* env = xv_env();
    if (!env) return 2;
    const XvNtPlan pl = xv_nt_plan(g.M, g.N, g.K, g.bn_part != nullptr, g.co_running != 0, g.ws != nullptr, g.ws_bytes, env->nt_sched);
    const int stats_kind = g.bn_part ? 0 : 1;      // XvProfScope kind
    if (pl.kind == XV_NT_SK) {
        NTSKArgs q;
        q.nk = ksteps;
        q.total = (long)tiles * ksteps;
        XV_REQUIRE(q.total < (1L << 31), "gemm_nt: %d tiles x %d K-steps exceed the evenly scheduled kernel's 32-bit unit index", tiles, ksteps);
        q.P = pl.p_sk;
        const bool shared_tiles = q.total % q.P != 0 || (q.total / q.P) % ksteps != 0;
        p.C = g.C; p.ldc = g.ldc; p.c_split_stride = 0; p.k_chunk = ksteps * BK;
        p.part_sum = g.bn_part; p.part_m2 = nullptr;
        q.g = p;
        q.slab = (float*)g.ws;
        q.tickets = shared_tiles ? tn_tickets_for(s) : nullptr;
        XV_REQUIRE(!shared_tiles || (q.tickets && ((uintptr_t)q.slab % 16) == 0), "gemm_nt: hand-over buffers unavailable");
        XvProfScope prof(s, stats_kind, 2.0 * g.M * g.N * g.K);
        if (g.bn_part) {
            if (p.row_store) hipLaunchKernelGGL((xv_gemm_nt_sk_kernel<true, true>), dim3(q.P), dim3(256), 0, s, q);
            else hipLaunchKernelGGL((xv_gemm_nt_sk_kernel<true, false>), dim3(q.P), dim3(256), 0, s, q);
        } else {
            if (p.row_store) hipLaunchKernelGGL((xv_gemm_nt_sk_kernel<false, true>), dim3(q.P), dim3(256), 0, s, q);
            else hipLaunchKernelGGL((xv_gemm_nt_sk_kernel<false, false>), dim3(q.P), dim3(256), 0, s, q);
        }
        XV_LAUNCH_CHECK();
        return 0;
    }
    if (pl.kind != XV_NT_SPLIT) {      // one workgroup per tile, or whole tiles + shares of the remaining ones
        p.C = g.C; p.ldc = g.ldc; p.c_split_stride = 0; p.k_chunk = ksteps * BK;
        dim3 grid(tiles, 1, 1);
        if (pl.kind == XV_NT_SHARES) {
            p.n_whole = tiles / 256 * 256;
            p.shares = pl.shares;
            p.slab = (float*)g.ws;
            p.tickets = tn_tickets_for(s);
            XV_REQUIRE(p.tickets && ((uintptr_t)p.slab % 16) == 0, "gemm_nt: hand-over buffers unavailable");
            grid.x = p.n_whole + (tiles - p.n_whole) * pl.shares;
        }
        XvProfScope prof(s, stats_kind, 2.0 * g.M * g.N * g.K);
        if (g.bn_part) {
            p.part_sum = g.bn_part;
            p.part_m2 = nullptr;
            if (p.row_store) hipLaunchKernelGGL((xv_gemm_nt_kernel<true, true>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((xv_gemm_nt_kernel<true, false>), grid, dim3(256), 0, s, p);
        } else {
            if (p.row_store) hipLaunchKernelGGL((xv_gemm_nt_kernel<false, true>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((xv_gemm_nt_kernel<false, false>), grid, dim3(256), 0, s, p);
        }
        XV_LAUNCH_CHECK();
        return 0;
    }
    // split-K over the whole chip + a slab-sum launch (few tiles, no BatchNorm statistics to emit)
    int splits = pl.splits;
    const long np = (long)xv_align(g.N, 4);
    XV_REQUIRE((size_t)splits * g.M * np * sizeof(float) <= g.ws_bytes, "gemm_nt: workspace too small (%zu bytes)", g.ws_bytes);
    p.C = (float*)g.ws; p.ldc = np; p.c_split_stride = (long)g.M * np;
#ifndef XV_NT_NO_ROWS
    p.row_store = g.N % 4 == 0 ? 1 : 0;      // (the slab is 256-byte aligned, np and M * np are multiples of 4)
#endif
    p.k_chunk = xv_cdiv(ksteps, splits) * BK;
    splits = xv_cdiv(g.K, p.k_chunk);
    p.bias = nullptr;
    dim3 grid(tiles, 1, splits);
    {
        XvProfScope prof(s, 1, 2.0 * g.M * g.N * g.K);
        if (p.row_store) hipLaunchKernelGGL((xv_gemm_nt_kernel<false, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((xv_gemm_nt_kernel<false, false>), grid, dim3(256), 0, s, p);
    }
    XV_LAUNCH_CHECK();
    long total = (long)g.M * g.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(xv_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)g.ws, splits,
                       (long)g.M * np, g.M, g.N, (int)np, g.bias, g.C, g.ldc);
    XV_LAUNCH_CHECK();
    return 0;
}

// -------------------------------------------------------------------------------------
// TN: weight gradients
// -------------------------------------------------------------------------------------
struct TNArgs {
    const float* A; long lda; int a_pitch;
    const float* B; long ldb; int b_pitch;
    int rps; float inv_rps;
    float* P;
    int M, N, R, r_chunk;
    int tiles_m, tiles_n;
    const float* zero;
    int ahead;      // K-step schedule (see the kernel)
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS image [r][128] (output index contiguous, exactly as it sits in HBM: no transpose).
// Fragment reads are ds_read_b64: lane i of a lane-half takes output rows 2i and 2i+1 of the
// wave's 64 at reduction row r = 2*ks + half, which feed the two 32x32 accumulators in that
// direction (the MFMA only needs A and B to agree on r).  So accumulator (a,b) register reg of
// lane l holds  m = m0 + wr*64 + 2*row(reg,l) + a,  n = n0 + wc*64 + 2*(l&31) + b.
// (Round 5 built wave groups on top of this kernel - 512 / 1 024-thread workgroups whose groups take consecutive reduction chunks of ONE tile
// and add their accumulators through LDS before a single slab store, 1/2 or 1/4 of the slab bytes: parity-green, no faster alone, 3 % slower
// in the step; removed.  profiles/r05_tn_wave_groups.txt, DESIGN.md Appendix A.)
#define XV_TN_STAGES 2
#ifndef XV_TN_AHEAD_MIN
#define XV_TN_AHEAD_MIN 16      // K-steps per workgroup from which the two-steps-ahead DMA schedule is used (xv_launch_gemm_tn; 96 until the staging addresses went scalar)
#endif
__global__ __launch_bounds__(256, XV_WGS_PER_CU) void xv_gemm_tn_kernel(TNArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[XV_TN_STAGES * 2 * BK * BM];   // [slot][A|B][BK][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // 1-D grid over (split, tile): after the XCD swizzle every XCD owns a contiguous run, i.e. whole
    // reduction chunks, so the rows of A and B that one chunk touches are fetched into ONE L2
    // (the (tiles, splits) grid of the first build spread every chunk over all 8 XCDs: 51 % hits).
    const int v = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tiles = p.tiles_m * p.tiles_n;
    const int split = v / tiles, t = v - split * tiles;
    const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int r_begin = split * p.r_chunk;
    const int r_end = min(p.R, r_begin + p.r_chunk);
    const int nk = (r_end - r_begin + BK - 1) / BK;

    // LDS-DMA staging: one wave-instruction = 1 KiB = two whole [r][128] rows of the image; lane l lands on
    // row 2*(RPT*wave+i) + l/32, columns 4*(l%32)..+3.  The reduction-row -> address map (spliced view)
    // is evaluated per lane without an integer divide (r < 2^24, float quotient off by <= 1).
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int lc = (lane & 31) * 4;
    const bool a_cv = (m0 + lc) < p.M, b_cv = (n0 + lc) < p.N;
    const float* __restrict__ zp = p.zero;
    const float* abase = p.A + (a_cv ? m0 + lc : 0);
    const float* bbase = p.B + (b_cv ? n0 + lc : 0);
    auto gstage_ragged = [&](int kt, int buf) {      // a K-step with rows at or beyond r_end: those read the zero page (they are summed)
        float* sa = smem + buf * (2 * BK * BM) + 2 * TN_RPT * uwave * BM;
        float* sb = sa + BK * BM;
#pragma unroll
        for (int i = 0; i < TN_RPT; ++i) {
            int r = r_begin + kt * BK + 2 * (TN_RPT * wave + i) + (lane >> 5);
            bool rv = r < r_end;
            int seg = (int)((float)r * p.inv_rps);
            int tt = r - seg * p.rps;
            seg += (tt >= p.rps) - (tt < 0);
            tt = r - seg * p.rps;
            const float* pa = (rv && a_cv) ? abase + ((long)seg * p.a_pitch + tt) * p.lda : zp;
            const float* pb = (rv && b_cv) ? bbase + ((long)seg * p.b_pitch + tt) * p.ldb : zp;
            xv_dma16_ptr(pa, sa + 2 * i * BM);
            xv_dma16_ptr(pb, sb + 2 * i * BN);
        }
    };
    // Full K-steps (every row below r_end): the K-step's first row r0 is resolved to (segment, frame) in SCALAR registers and advanced
    // there - frame += BK, and on crossing the segment's last frame the base also skips the rows between two segments; a lane adds its
    // constant offset (its row j of the 16, its four columns), taken from the "next segment" copy when frame + j lies beyond the segment:
    // one compare and two selects per DMA pair.  (The first build resolved row -> (segment, frame) -> 64-bit address from scratch for every
    // K-step: ~40 vector instructions per wave and step; rounds 1-4 advanced per-lane offsets: 14 - and a vector instruction takes its ALU
    // cycles from the fp32 MFMAs.)  Columns outside the matrix read column 0: their products are never stored.
    const bool steady = p.rps >= BK;      // at most one segment boundary per step
    int jrow[TN_RPT];
    unsigned aoff[TN_RPT], boff[TN_RPT], aoff_w[TN_RPT], boff_w[TN_RPT];
    const unsigned a_step = (unsigned)(BK * p.lda * 4), b_step = (unsigned)(BK * p.ldb * 4);
    const unsigned a_skip = (unsigned)((long)(p.a_pitch - p.rps) * p.lda * 4), b_skip = (unsigned)((long)(p.b_pitch - p.rps) * p.ldb * 4);
#pragma unroll
    for (int i = 0; i < TN_RPT; ++i) {
        jrow[i] = 2 * (TN_RPT * wave + i) + (lane >> 5);
        aoff[i] = (unsigned)(((long)jrow[i] * p.lda + (a_cv ? m0 + lc : 0)) * 4);
        boff[i] = (unsigned)(((long)jrow[i] * p.ldb + (b_cv ? n0 + lc : 0)) * 4);
        aoff_w[i] = aoff[i] + a_skip;
        boff_w[i] = boff[i] + b_skip;
    }
    // scalar state of the next K-step to stage: frame of its first row within its segment, byte offsets of that row in A and B
    int s_tt;
    unsigned s_a, s_b;
    {
        const int r0 = __builtin_amdgcn_readfirstlane(min(r_begin, p.R - 1));
        const int seg0 = r0 / p.rps;
        s_tt = r0 - seg0 * p.rps;
        s_a = (unsigned)(((long)seg0 * p.a_pitch + s_tt) * p.lda * 4);
        s_b = (unsigned)(((long)seg0 * p.b_pitch + s_tt) * p.ldb * 4);
    }
    const unsigned lds0 = xv_lds_addr(smem + 2 * TN_RPT * uwave * BM);
    auto gstage = [&](int kt, int buf) {
        if (!steady || r_begin + (kt + 1) * BK > r_end) {
            gstage_ragged(kt, buf);
            return;
        }
        // (kt counts up by one per call, so the scalar state is at step kt here)
        const float* sa = (const float*)((const char*)p.A + s_a);
        const float* sb = (const float*)((const char*)p.B + s_b);
        const int thr = p.rps - s_tt;      // rows j >= thr of this K-step belong to the next segment
#pragma unroll
        for (int i = 0; i < TN_RPT; ++i) {
            const bool w = jrow[i] >= thr;
            xv_dma16(sa, w ? aoff_w[i] : aoff[i], lds0 + (buf * (2 * BK * BM) + 2 * i * BM) * 4);
            xv_dma16(sb, w ? boff_w[i] : boff[i], lds0 + (buf * (2 * BK * BM) + BK * BM + 2 * i * BN) * 4);
        }
        s_tt += BK;
        s_a += a_step;
        s_b += b_step;
        if (s_tt >= p.rps) { s_tt -= p.rps; s_a += a_skip; s_b += b_skip; }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int a_off = lh * BM + wr * 64 + 2 * li;
    const int b_off = lh * BN + wc * 64 + 2 * li;
    // Schedule of a K-step, two forms.  AHEAD: all sixteen fragment reads of stage kt, the first sixteen MFMAs, then - in mid-step - the
    // wait for the DMA of stage kt + 1, the workgroup barrier (every wave has read stage kt: its slot is free; stage kt + 1 is visible),
    // the DMA of stage kt + 2 into the slot just freed, and the other sixteen MFMAs: a stage's loads have a whole K-step to land.
    // sched_barrier pins the order.  Plain: the DMA of stage kt + 1 at the top of step kt, wait + barrier where hipcc puts them (right
    // behind the fragment reads - the last MEMORY operations of the step - i.e. in front of all 32 MFMAs).  The launcher chooses per
    // problem (TNArgs::ahead).
    if (nk > 0) gstage(0, 0);
    xv_dma_wait_all();      // (the compiler does not see xv_dma16's loads)
    __syncthreads();
    auto k_loop = [&](auto ahead_c) {
        constexpr bool AHEAD = decltype(ahead_c)::value;
        if (AHEAD && nk > 1) gstage(1, 1);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (!AHEAD && kt + 1 < nk) gstage(kt + 1, buf ^ 1);
            const float* sa = smem + buf * (2 * BK * BM) + a_off;
            const float* sb = smem + buf * (2 * BK * BM) + BK * BM + b_off;
            f32x2 af[BK / 4], bf[BK / 4], an[BK / 4], bn[BK / 4];
#pragma unroll
            for (int j = 0; j < BK / 4; ++j) {
                af[j] = *(const f32x2*)(sa + 2 * j * BM);
                bf[j] = *(const f32x2*)(sb + 2 * j * BN);
            }
#pragma unroll
            for (int j = 0; j < BK / 4; ++j) {
                an[j] = *(const f32x2*)(sa + 2 * (BK / 4 + j) * BM);
                bn[j] = *(const f32x2*)(sb + 2 * (BK / 4 + j) * BN);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < BK / 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j].x, bf[j].x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j].x, bf[j].y, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j].y, bf[j].x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j].y, bf[j].y, acc[1][1], 0, 0, 0);
            }
            if (AHEAD) {
                __builtin_amdgcn_sched_barrier(0);
                xv_dma_wait_all();
                __syncthreads();
                if (kt + 2 < nk) gstage(kt + 2, buf);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < BK / 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(an[j].x, bn[j].x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(an[j].x, bn[j].y, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(an[j].y, bn[j].x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(an[j].y, bn[j].y, acc[1][1], 0, 0, 0);
            }
            if (!AHEAD) {
                xv_dma_wait_all();
                __syncthreads();
            }
        }
    };
    if (p.ahead) k_loop(std::true_type{});
    else k_loop(std::false_type{});

    // (Summing the split partials of DIFFERENT workgroups inside this kernel - the last arrival adds the others' slabs - was built and dropped in
    // round 3: one workgroup reads splits x 64 KB serially at the end of the launch; DESIGN.md Appendix A.)
    // The slab tile goes out through the group's (idle) staging LDS in whole rows of 16 bytes per lane, as in nt_store_tile_rows: pass a writes
    // the tile's rows 2 q + a (q = wr * 32 + the accumulator's row) as a [64][128] image - a lane holds two neighbouring columns of a row, one
    // ds_write_b64 - and every thread stores eight float4 of whole rows (16 store instructions per lane instead of 32 eight-byte ones).
    // (M and N are multiples of 4 and the slab base is 16-byte aligned: xv_launch_gemm_tn.)
    float* P = p.P + (long)split * p.M * p.N;
    __syncthreads();                              // (every fragment read of the last stage is done)
    float* wbase = smem + (wr * 32 + 4 * lh) * BN + wc * 64 + 2 * li;
    const int rq = tid >> 5, cq = (tid & 31) * 4;
    const bool col_ok = n0 + cq < p.N;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x2 v = {acc[a][0][r], acc[a][1][r]};
            *(f32x2*)(wbase + ((r & 3) + 8 * (r >> 2)) * BN) = v;
        }
        __syncthreads();
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *(const f32x4*)(smem + (rq + 8 * i) * BN + cq);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + 2 * (rq + 8 * i) + a;
            if (m < p.M && col_ok) *(f32x4*)(P + (long)m * p.N + n0 + cq) = v[i];
        }
        if (a == 0) __syncthreads();
    }
}


// ---- weight gradient of a layer with 129 ... 160 input rows (tdnn1: 5 taps x 32 padded feature channels) ----------------------------
// On 128 x 128 tiles those rows are two tiles of which 3/8 of the second is real: 1.6 x the MFMAs, on the last weight gradient of every step.
// Here a workgroup owns ALL rows x 128 columns: wave w = columns 32 w .. 32 w + 31, five 32 x 32 accumulators down the rows (rows 0..127
// interleaved in pairs as in xv_gemm_tn_kernel - one ds_read_b64 feeds two of them - rows 128..159 straight).  LDS image per stage: A [16][160]
// (linear: an LDS-DMA piece of 1 KB is 1.6 rows, every lane resolves its own (row, column)) + B [16][128]; 2 workgroups per CU, 512 of them =
// one round.  (67 -> 55 us alone at S1; workgroup counts and schedules tried: DESIGN.md Appendix B, note 8.)
#define TNW_M 160
#define TNW_A_PIECES (BK * TNW_M / 256)      // 1 KB pieces per stage: 10
#define TNW_B_PIECES (BK * BN / 256)         // 8
#define TNW_WGS 512                         // (768 / 1 024 workgroups were slower alone: Appendix B, note 8; 256 / 384 no different in the step: round 6)
__global__ __launch_bounds__(256, 2) void xv_gemm_tn160_kernel(TNArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (TNW_M + BN)];      // [slot][A [16][160] | B [16][128]]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int v = xcd_swizzle(blockIdx.x, gridDim.x);
    const int split = v / p.tiles_n, tile_n = v - split * p.tiles_n;
    const int n0 = tile_n * BN;
    const int r_begin = split * p.r_chunk;
    const int r_end = min(p.R, r_begin + p.r_chunk);
    const int nk = (r_end - r_begin + BK - 1) / BK;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const float* __restrict__ zp = p.zero;
    constexpr int SLOT = BK * (TNW_M + BN);
    // pieces: A piece q (q = wave, wave + 4, wave + 8 < 10) = floats [256 q, 256 q + 256) of the stage's A image; B piece q (q = 2 wave, 2 wave + 1) =
    // rows 2 q, 2 q + 1 of B.  Per lane and piece: the reduction row within the K-step and the byte offset of its 16 bytes in that row.
    constexpr int NA = 3, NB = 2;
    int rowA[NA], rowB[NB];
    unsigned colA[NA], colB[NB];
    bool pieceA[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = wave + 4 * i;
        pieceA[i] = q < TNW_A_PIECES;
        const int f = 256 * q + 4 * lane;
        rowA[i] = f / TNW_M;
        const int c = f - rowA[i] * TNW_M;
        colA[i] = (unsigned)((c < p.M ? c : 0) * 4);      // columns beyond M (M < 160) read column 0: their products are never stored
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        rowB[i] = 2 * (NB * wave + i) + (lane >> 5);
        const int c = n0 + 4 * (lane & 31);
        colB[i] = (unsigned)((c < p.N ? c : 0) * 4);
    }
    // a K-step with rows at or beyond r_end (they read the zero page: they are summed) or of a layer with segments shorter than a K-step: every
    // lane resolves reduction row -> (segment, frame) -> address
    auto gstage_ragged = [&](int kt, int buf) {
        float* sa = smem + buf * SLOT;
        float* sb = sa + BK * TNW_M;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (!pieceA[i]) continue;      // (uniform per wave)
            const int r = r_begin + kt * BK + rowA[i];
            int seg = (int)((float)r * p.inv_rps);
            int tt = r - seg * p.rps;
            seg += (tt >= p.rps) - (tt < 0);
            tt = r - seg * p.rps;
            const float* pa = r < r_end ? (const float*)((const char*)(p.A + ((long)seg * p.a_pitch + tt) * p.lda) + colA[i]) : zp;
            xv_dma16_ptr(pa, sa + 256 * (uwave + 4 * i));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int r = r_begin + kt * BK + rowB[i];
            int seg = (int)((float)r * p.inv_rps);
            int tt = r - seg * p.rps;
            seg += (tt >= p.rps) - (tt < 0);
            tt = r - seg * p.rps;
            const float* pb = r < r_end ? (const float*)((const char*)(p.B + ((long)seg * p.b_pitch + tt) * p.ldb) + colB[i]) : zp;
            xv_dma16_ptr(pb, sb + 256 * (NB * uwave + i));
        }
    };
    // full K-steps: the scheme of xv_gemm_tn_kernel - the K-step's first row lives in scalar registers as (frame within its segment, byte
    // offsets in A and B) and advances there; a lane adds the constant offset of its row of the 16 and of its columns, from the "next
    // segment" copy when frame + row lies beyond the segment.  (Rounds 4's form resolved every lane's row with a float multiply, two
    // conversions and 64-bit address arithmetic per piece: ~100 vector instructions per wave and K-step beside 20 MFMAs.)
    const bool steady = p.rps >= BK;
    const unsigned a_skip = (unsigned)((long)(p.a_pitch - p.rps) * p.lda * 4), b_skip = (unsigned)((long)(p.b_pitch - p.rps) * p.ldb * 4);
    const unsigned a_step = (unsigned)(BK * p.lda * 4), b_step = (unsigned)(BK * p.ldb * 4);
    unsigned offA[NA], offA_w[NA], offB[NB], offB_w[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) { offA[i] = (unsigned)((long)rowA[i] * p.lda * 4) + colA[i]; offA_w[i] = offA[i] + a_skip; }
#pragma unroll
    for (int i = 0; i < NB; ++i) { offB[i] = (unsigned)((long)rowB[i] * p.ldb * 4) + colB[i]; offB_w[i] = offB[i] + b_skip; }
    int s_tt;
    unsigned s_a, s_b;
    {
        const int r0 = __builtin_amdgcn_readfirstlane(min(r_begin, p.R - 1));
        const int seg0 = r0 / p.rps;
        s_tt = r0 - seg0 * p.rps;
        s_a = (unsigned)(((long)seg0 * p.a_pitch + s_tt) * p.lda * 4);
        s_b = (unsigned)(((long)seg0 * p.b_pitch + s_tt) * p.ldb * 4);
    }
    const unsigned lds_a0 = xv_lds_addr(smem + 256 * uwave), lds_b0 = xv_lds_addr(smem + BK * TNW_M + 256 * NB * uwave);
    auto gstage = [&](int kt, int buf) {
        if (!steady || r_begin + (kt + 1) * BK > r_end) {
            gstage_ragged(kt, buf);
            return;
        }
        // (kt counts up by one per call, so the scalar state is at step kt here)
        const float* sa = (const float*)((const char*)p.A + s_a);
        const float* sb = (const float*)((const char*)p.B + s_b);
        const int thr = p.rps - s_tt;      // rows >= thr of this K-step belong to the next segment
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (!pieceA[i]) continue;      // (uniform per wave)
            xv_dma16(sa, rowA[i] >= thr ? offA_w[i] : offA[i], lds_a0 + (buf * SLOT + 256 * 4 * i) * 4);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) xv_dma16(sb, rowB[i] >= thr ? offB_w[i] : offB[i], lds_b0 + (buf * SLOT + 256 * i) * 4);
        s_tt += BK;
        s_a += a_step;
        s_b += b_step;
        if (s_tt >= p.rps) { s_tt -= p.rps; s_a += a_skip; s_b += b_skip; }
    };

    f32x16 acc[5];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int a_off = lh * TNW_M + 2 * li;              // rows 2 li, 2 li + 1 (and + 64) of reduction row 2 ks + lh
    const int a4_off = lh * TNW_M + 128 + li;           // row 128 + li
    const int b_off = lh * BN + 32 * wave + li;
    // (plain double buffering: the two-steps-ahead form of xv_gemm_tn_kernel was slower here, Appendix B note 8)
    if (nk > 0) gstage(0, 0);
    xv_dma_wait_all();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gstage(kt + 1, buf ^ 1);
        const float* sa = smem + buf * SLOT;
        const float* sb = sa + BK * TNW_M;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x2 a01[BK / 4], a23[BK / 4];
            float a4[BK / 4], bf[BK / 4];
#pragma unroll
            for (int j = 0; j < BK / 4; ++j) {
                const int ks = h * (BK / 4) + j;
                a01[j] = *(const f32x2*)(sa + 2 * ks * TNW_M + a_off);
                a23[j] = *(const f32x2*)(sa + 2 * ks * TNW_M + a_off + 64);
                a4[j] = sa[2 * ks * TNW_M + a4_off];
                bf[j] = sb[2 * ks * BN + b_off];
            }
#pragma unroll
            for (int j = 0; j < BK / 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01[j].x, bf[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01[j].y, bf[j], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a23[j].x, bf[j], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a23[j].y, bf[j], acc[3], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], bf[j], acc[4], 0, 0, 0);
            }
        }
        xv_dma_wait_all();
        __syncthreads();
    }
    // slab [split][M][N]: accumulator a < 4, register r of lane (li, lh) = row 64 (a / 2) + 2 row(r, lh) + a % 2; accumulator 4 = row 128 + row(r, lh);
    // column n0 + 32 wave + li
    float* P = p.P + (long)split * p.M * p.N;
    const int n = n0 + 32 * wave + li;
    if (n < p.N) {
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ri = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = a < 4 ? 64 * (a >> 1) + 2 * ri + (a & 1) : 128 + ri;
                if (m < p.M) P[(long)m * p.N + n] = acc[a][r];
            }
    }
}

static bool tn_wide_rows(int M) { return M > BM && M <= TNW_M; }

// Decomposition of a weight-gradient GEMM: `splits` slabs (workgroups per tile), `chunk` reduction rows per workgroup.  One co-resident round:
// XV_WGS_PER_CU (4) workgroups fit a CU (LDS 32 KB, 128 VGPRs each), i.e. 1 024 - more costs a second round (2 x on the first build, +2 % in
// the round-5 step at 1 536 / 2 048), fewer leaves matrix pipes idle.  The history of this choice (768 / 896 workgroups, an even schedule,
// unsplit segment-level gradients, tdnn1 as 128 ... 512 workgroups, wave groups) is DESIGN.md Appendix A.
struct XvTnPlan { int splits, chunk; };
static XvTnPlan xv_tn_plan(int M, int N, int R, bool direct = false) {
    const int tiles = xv_cdiv(M, BM) * xv_cdiv(N, BN);
    const int ksteps = xv_cdiv(R, BK);
    XvTnPlan pl = {1, 0};
    if (tn_wide_rows(M)) {      // xv_gemm_tn160_kernel: one round of 512 workgroups (2 per CU), at least 4 K-steps each
        const int splits = std::max(1, std::min(TNW_WGS / xv_cdiv(N, BN), ksteps / 4));
        pl.chunk = xv_cdiv(ksteps, splits) * BK;
        pl.splits = xv_cdiv(R, pl.chunk);
        return pl;
    }
    const int min_ksteps = 2;                                          // fewest K-steps a workgroup is given
    int splits = std::max(1, XV_RESIDENT_WGS / tiles);
    if (splits > ksteps / min_ksteps) splits = std::max(1, ksteps / min_ksteps);
    // A short reduction over many tiles (the loss head's weight gradient: 232 tiles x 8 K-steps at 128 chunks) is not split: four splits of two
    // K-steps each wrote and re-read 60 MB of slabs beside the latency-bound chain of the segment layers (its d-out launch ran 45 instead of
    // 23 us); one workgroup per tile can store the result where it belongs (xv_engine.hip, loss_head_wgrad)
    // - only for the caller that asks for it (XvGemmTN::direct: its unsplit result is stored straight into the destination); a LAYER's
    // weight gradient with many tiles and a short reduction (small-batch fine-tuning) keeps its splits - unsplit it would leave most of the
    // chip idle
    if (direct && ksteps <= 16 && tiles >= 128) splits = 1;
    // A layer with few tiles (a 512 x 512 dense layer: 16) gets ONE workgroup per CU, not a co-resident round of four: 16 tiles x 16 splits =
    // 256 workgroups.  [measured, round 6, same box, alternated, profiles/r06_tn_few_tiles.txt] against 64 splits (1 024 workgroups, 67 MB of
    // slabs for a 1 MB gradient): S1 -0.4 ... -0.7 %, 64 x U{200..400} -0.1 ... -0.4 %, S2 -0.4 %, S4 -0.4 %, S5 -0.1 ... -0.4 %; alone the
    // layer's weight gradient 111.7 -> 108.4 us.  8 / 12 / 20 / 24 splits are slower than 16 (256 workgroups = the same load on every CU);
    // the same cap on the 48-tile layers (tdnn5: 21 -> 16 splits = 768 workgroups) is neutral, on the 80 / 112-tile layers harmful.
    if (tiles <= 16) splits = std::min(splits, std::max(1, 256 / tiles));
    pl.chunk = xv_cdiv(ksteps, splits) * BK;
    pl.splits = xv_cdiv(R, pl.chunk);
    return pl;
}
int xv_tn_splits(int M, int N, int R) { return xv_tn_plan(M, N, R).splits; }
int xv_tn_splits_direct(int M, int N, int R) { return xv_tn_plan(M, N, R, true).splits; }

int xv_launch_gemm_tn(hipStream_t s, const XvGemmTN& g) {
    XV_REQUIRE(g.M % 4 == 0 && g.N % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0,
               "gemm_tn: M/N/lda/ldb must be multiples of 4 (M=%d N=%d lda=%ld ldb=%ld)", g.M, g.N, g.lda, g.ldb);
    XV_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0, "gemm_tn: operands must be 16-byte aligned");
    XV_REQUIRE(g.M > 0 && g.N > 0 && g.R > 0 && g.splits >= 1, "gemm_tn: empty problem");
    if (ensure_zero_page()) return 1;
    TNArgs p;
    p.zero = g_zero_page;
    XV_REQUIRE(g.a_rps == g.b_rps && g.a_rps > 0, "gemm_tn: both operands must share the rows-per-segment");
    XV_REQUIRE(g.R < (1 << 24), "gemm_tn: at most 2^24 reduction rows");
    {   // xv_dma16 addresses every operand row as a 32-bit byte offset from the operand's base
        const long segs = xv_cdiv(g.R, g.a_rps);
        XV_REQUIRE((segs * g.a_pitch + 1) * g.lda * 4 < (1L << 32) && (segs * g.b_pitch + 1) * g.ldb * 4 < (1L << 32),
                   "gemm_tn: an operand spans 4 GB or more (%ld segments of %d / %d rows, lda=%ld ldb=%ld): split the batch", segs, g.a_pitch,
                   g.b_pitch, g.lda, g.ldb);
    }
    p.A = g.A; p.lda = g.lda; p.a_pitch = g.a_pitch;
    p.B = g.B; p.ldb = g.ldb; p.b_pitch = g.b_pitch;
    p.rps = g.a_rps;
    // Rows without gaps between the segments (pitch == rows per segment in both operands: every dense layer, whose "segments" are single
    // frames) are ONE segment of R rows: the kernel's cheap K-step form - scalar bases, lane offsets that advance by a constant - needs
    // at least BK rows per segment, and with one-row segments every K-step took the form meant for ragged ends (a row -> address
    // resolution and a 64-bit address per lane: tdnn4 / tdnn5 ran 104 / 115 TF against 128-135 for the layers with taps; round 5).
    if (g.a_pitch == g.a_rps && g.b_pitch == g.b_rps) { p.rps = g.R; p.a_pitch = g.R; p.b_pitch = g.R; }
    p.inv_rps = 1.0f / (float)p.rps;
    p.P = g.P; p.M = g.M; p.N = g.N; p.R = g.R;
    p.tiles_m = xv_cdiv(g.M, BM); p.tiles_n = xv_cdiv(g.N, BN);
    const XvTnPlan pl = xv_tn_plan(g.M, g.N, g.R, g.direct != 0);
    XV_REQUIRE(pl.splits == g.splits, "gemm_tn: splits must come from xv_tn_splits%s (%d vs %d)", g.direct ? "_direct" : "", pl.splits, g.splits);
    p.r_chunk = pl.chunk;
    p.ahead = p.r_chunk / BK >= XV_TN_AHEAD_MIN;      // (Appendix B, note 9)
    const int wgs = p.tiles_m * p.tiles_n * pl.splits;
    {
        XvProfScope prof(s, 2, 2.0 * g.M * g.N * g.R);
        if (tn_wide_rows(g.M)) hipLaunchKernelGGL(xv_gemm_tn160_kernel, dim3(p.tiles_n * pl.splits), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(xv_gemm_tn_kernel, dim3(wgs), dim3(256), 0, s, p);
    }
    XV_LAUNCH_CHECK();
    return 0;
}

// -------------------------------------------------------------------------------------
// Public op-level wrappers around the two GEMMs
// -------------------------------------------------------------------------------------
extern "C" size_t xv_op_workspace_bytes(int rows, int cols_in, int cols_out) {
    // split slabs: at most XV_RESIDENT_WGS (+ one ragged round) workgroup tiles of 128x128 floats, plus column partials
    size_t slabs = (size_t)(2 * XV_RESIDENT_WGS + 256) * BM * BN * sizeof(float);      // NT: two shared tiles per workgroup; TN: two segments
    size_t wg = (size_t)xv_align(cols_in, BM) * xv_align(cols_out, BN) * sizeof(float) * 8;
    size_t part = ((size_t)xv_cdiv(rows > 0 ? rows : 1, 64) * 2 + 2) * (size_t)(cols_out > cols_in ? cols_out : cols_in) * sizeof(float);
    size_t m = slabs > wg ? slabs : wg;
    return xv_align((m > part ? m : part) + 4096, 256);
}

extern "C" int xv_affine_forward(void* stream, const float* x, int segs, int t_in, int c_pad, int k, const float* wt,
                                 const float* bias, float* z, int o, int ldz, float* bn_part, void* ws, size_t ws_bytes) {
    XV_REQUIRE(segs > 0 && k >= 1 && t_in >= k && c_pad > 0 && o > 0 && ldz >= o, "affine_forward: bad shape (t_in=%d k=%d)", t_in, k);
    XvGemmNT g = {};
    g.A = x; g.lda = c_pad; g.a_rps = t_in - k + 1; g.a_pitch = t_in;
    g.Bt = wt; g.ldb = (long)k * c_pad;
    g.C = z; g.ldc = ldz;
    g.M = segs * (t_in - k + 1); g.N = o; g.K = k * c_pad;
    g.bias = bias; g.bn_part = bn_part; g.ws = ws; g.ws_bytes = ws_bytes;
    return xv_launch_gemm_nt((hipStream_t)stream, g);
}

int xv_affine_dgrad_ld(hipStream_t stream, const float* dz_pad, int ldo, int segs, int t_out, int o, int k, const float* wf, float* dx, int c,
                       void* ws, size_t ws_bytes) {
    XV_REQUIRE(segs > 0 && k >= 1 && t_out >= 1 && o > 0 && c > 0 && ldo >= o && (ldo == o || k == 1), "affine_dgrad: bad shape (a row pitch needs k = 1)");
    XvGemmNT g = {};
    g.A = dz_pad; g.lda = ldo; g.a_rps = t_out + k - 1; g.a_pitch = t_out + 2 * (k - 1);
    g.Bt = wf; g.ldb = (long)k * o;
    g.C = dx; g.ldc = c;
    g.M = segs * (t_out + k - 1); g.N = c; g.K = k * o;
    g.ws = ws; g.ws_bytes = ws_bytes;
    g.co_running = 1;          // the data gradient runs beside the layer's weight gradient (xv_engine.hip: two streams)
    return xv_launch_gemm_nt(stream, g);
}
extern "C" int xv_affine_dgrad(void* stream, const float* dz_pad, int segs, int t_out, int o, int k, const float* wf, float* dx,
                               int c, void* ws, size_t ws_bytes) {
    return xv_affine_dgrad_ld((hipStream_t)stream, dz_pad, o, segs, t_out, o, k, wf, dx, c, ws, ws_bytes);
}

// out[(j*C + c)][n] = sum_z P[z][j*c_pad + c][n] (+ l2 * w[(j*C + c)][n])
// Block = 64 column quads (16 B per lane) x 4; splits are added in split order with eight loads in flight.  ZSPLIT (many splits, few
// output rows - tdnn1: 150 rows x 128 splits): the 4 thread groups of a block take contiguous runs of the splits of ONE row and their sums
// are added in group order (a fixed association: bit-reproducible).  Otherwise the 4 groups are 4 rows and a thread adds all splits of
// its quad.  Row blocks are grid-strided (at most ~2 048 workgroups: a launch of many short workgroups crawls beside a GEMM).
// (The first build read 4 bytes per lane with two 64-bit divisions per element: 45 us for tdnn2's 63 MB of slabs alone on the chip -
// and the last of these launches sits between the last GEMM of a step and the update.)
#define WR_FLIGHT 8
template <bool ZSPLIT>
__global__ __launch_bounds__(256) void xv_wgrad_reduce_kernel(const float* __restrict__ P, int splits, long slab, int rows, int C, int c_pad,
                                                              int n_in, int nq_out, const float* __restrict__ w, long ldw, float l2,
                                                              float* __restrict__ out, long ldo) {
    XV_EW_PRIORITY();
    __shared__ f32x4 part[ZSPLIT ? 4 : 1][64];
    const int qx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + qx;
    const bool qv = q < nq_out;
    const int per = ZSPLIT ? (splits + 3) / 4 : splits;
    const int z0 = ZSPLIT ? g * per : 0, z1 = min(splits, z0 + per);
    for (int rb = blockIdx.y; rb * (ZSPLIT ? 1 : 4) < rows; rb += gridDim.y) {
        const int row = ZSPLIT ? rb : 4 * rb + g;
        const bool rv = row < rows;
        const int rr = rv ? row : 0;
        const int j = rr / C, c = rr - j * C;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (qv && rv) {
            const float* src = P + ((long)j * c_pad + c) * n_in + 4 * q;
            for (int z = z0; z < z1; z += WR_FLIGHT) {
                f32x4 t[WR_FLIGHT];
#pragma unroll
                for (int u = 0; u < WR_FLIGHT; ++u) t[u] = *(const f32x4*)(src + (long)min(z + u, z1 - 1) * slab);
#pragma unroll
                for (int u = 0; u < WR_FLIGHT; ++u)
                    if (z + u < z1) v += t[u];
            }
        }
        if (ZSPLIT) {
            __syncthreads();      // (the previous row's sums have been read)
            part[g][qx] = v;
            __syncthreads();
            if (g != 0) continue;
            v = ((part[0][qx] + part[1][qx]) + part[2][qx]) + part[3][qx];
        }
        if (qv && rv) {
            if (w) v += l2 * *(const f32x4*)(w + (long)row * ldw + 4 * q);
            *(f32x4*)(out + (long)row * ldo + 4 * q) = v;
        }
    }
}

int xv_launch_wgrad_reduce(hipStream_t s, const float* P, int splits, int k, int C, int c_pad, int n_in, int n_out, const float* w,
                           long ldw, float l2, float* out, long ldo) {
    XV_REQUIRE(n_out % 4 == 0 && n_in % 4 == 0 && ldo % 4 == 0 && (!w || ldw % 4 == 0) && ((uintptr_t)P % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                   (!w || ((uintptr_t)w % 16) == 0),
               "wgrad_reduce: widths and leading dimensions must be multiples of 4 floats, buffers 16-byte aligned (n_out=%d)", n_out);
    const int rows = k * C, gx = xv_cdiv(n_out / 4, 64);
    const bool zsplit = splits >= 32 && (long)rows * gx < 1024;
    const int row_blocks = zsplit ? rows : xv_cdiv(rows, 4);
    const dim3 grid(gx, std::min(row_blocks, std::max(1, 2048 / gx)));      // (256 ... 4 096: no difference in the step, profiles/r06_tn_few_tiles.txt)
    if (zsplit)
        hipLaunchKernelGGL(xv_wgrad_reduce_kernel<true>, grid, dim3(256), 0, s, P, splits, (long)k * c_pad * n_in, rows, C, c_pad, n_in, n_out / 4, w,
                           ldw, l2, out, ldo);
    else
        hipLaunchKernelGGL(xv_wgrad_reduce_kernel<false>, grid, dim3(256), 0, s, P, splits, (long)k * c_pad * n_in, rows, C, c_pad, n_in, n_out / 4, w,
                           ldw, l2, out, ldo);
    XV_LAUNCH_CHECK();
    return 0;
}

int xv_affine_wgrad_ld(hipStream_t stream, const float* x, int segs, int t_in, int c_pad, int k, int c, const float* dz, int ldo, int dz_seg_pitch,
                       int dz_row0, int o, const float* kernel, float l2_scale, float* dkernel, void* ws, size_t ws_bytes) {
    XV_REQUIRE(segs > 0 && k >= 1 && t_in >= k && c_pad >= c && o > 0 && o % 4 == 0 && ldo >= o && ldo % 4 == 0,
               "affine_wgrad: bad shape (o=%d must be a multiple of 4)", o);
    const int t_out = t_in - k + 1;
    XvGemmTN g = {};
    g.A = x; g.lda = c_pad; g.a_rps = t_out; g.a_pitch = t_in;
    g.B = dz + (long)dz_row0 * ldo; g.ldb = ldo; g.b_rps = t_out; g.b_pitch = dz_seg_pitch;
    g.M = k * c_pad; g.N = o; g.R = segs * t_out;
    g.splits = xv_tn_splits(g.M, g.N, g.R);
    XV_REQUIRE((size_t)g.splits * g.M * g.N * sizeof(float) <= ws_bytes, "affine_wgrad: workspace too small (%zu needed)",
               (size_t)g.splits * g.M * g.N * sizeof(float));
    g.P = (float*)ws;
    int rc = xv_launch_gemm_tn(stream, g);
    if (rc) return rc;
    return xv_launch_wgrad_reduce(stream, g.P, g.splits, k, c, c_pad, o, o, l2_scale != 0.f ? kernel : nullptr, o,
                                  l2_scale, dkernel, o);
}
extern "C" int xv_affine_wgrad(void* stream, const float* x, int segs, int t_in, int c_pad, int k, int c, const float* dz,
                               int dz_seg_pitch, int dz_row0, int o, const float* kernel, float l2_scale, float* dkernel, void* ws,
                               size_t ws_bytes) {
    return xv_affine_wgrad_ld((hipStream_t)stream, x, segs, t_in, c_pad, k, c, dz, o, dz_seg_pitch, dz_row0, o, kernel, l2_scale, dkernel, ws, ws_bytes);
}
