// Internal helpers shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "xvector_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define XV_WAVE 64

void xv_set_error(const char* fmt, ...);

#define XV_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            xv_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

#define XV_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            xv_set_error(__VA_ARGS__);   \
            return 2;                    \
        }                                \
    } while (0)

#define XV_LAUNCH_CHECK()                                                               \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess) {                                                         \
            xv_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

// Environment switches of the native library: the few INTEGRATION.md section 6 documents, read once (thread-safe), value-checked.
// xv_env() returns null (xv_last_error says why) when a known switch holds a value it does not understand; an XV_* variable that nothing in
// this package reads is named once on stderr and ignored (it may belong to another program).  Every entry point that consults a switch
// fails with that message.
struct XvEnv {
    int segment_fused;      // XV_SEGMENT_FUSED=0|1 (default 1): the segment-level layers as one launch each (xv_skinny.hip)
    int nt_sched;           // XV_NT_SCHED=dp|sk: force the schedule of the fp32 NT GEMM (0 = chosen per problem, 1 = dp, 2 = sk); diagnostics
    int dz_slots;           // XV_DZ_SLOTS=2: the two-slot dz ring in fp32 mode too (what an arena too large for a slot per layer gets; A/B and test switch)
    int conv_wr;            // XV_CONV_WR=4: 256-row tiles of the f16x3 context-window GEMM (kept parity-tested, off by default)
};
const XvEnv* xv_env();

// Wave priority of the kernels that are NOT fp32-MFMA GEMMs (element-wise, reductions, the segment-level chain).  On gfx950 a vector
// instruction and an fp32-input MFMA share one ALU (profiles/r04_valu_mfma_probe.txt) and the arbiter serves the highest priority, then the
// OLDEST wave: beside a GEMM whose waves always have an MFMA ready, a younger element-wise wave at priority 0 issues only in the gaps -
// such kernels ran 6-10 x slower beside the weight-gradient GEMM even with a free slot on every CU (profiles/r05_ew_priority.txt).  At
// priority 3 their few vector instructions go first; the GEMM loses the cycles they take, nothing else.
#ifndef XV_EW_PRIO
#define XV_EW_PRIO 3
#endif
#if XV_EW_PRIO
#define XV_EW_PRIORITY() __builtin_amdgcn_s_setprio(XV_EW_PRIO)
#else
#define XV_EW_PRIORITY() ((void)0)
#endif
// ... except the kernels that only fill a side stream with hours of slack (weight-layout copies and the loss head's weight preparation
// beside the forward GEMMs, the loss head's weight gradient beside the first data gradient): at priority 3 they finished three times
// sooner than anyone needs them and took 11 us from tdnn2's forward GEMM (profiles/r05_ew_priority.txt)
#define XV_EW_FILLER() ((void)0)

static inline int xv_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t xv_align(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- cross-workgroup hand-over inside one launch (split sums, last-arriver epilogues) --------------------------------------------
// Protocol (xv_skinny.hip, xv_loss.hip, xv_gemm.hip): a producer workgroup writes its partial results with xv_handoff_store, every
// storing wave drains them (xv_handoff_drain), the workgroup meets at a barrier, ONE lane takes a ticket (xv_ticket_take); the workgroup
// whose ticket is the last one reads every partial with xv_handoff_load after a second barrier.  All three are RELAXED agent-scope
// atomics: on gfx950 that is `global_store/load ... sc1` (written through to / read from the memory level all XCDs share) and a
// device-scope RMW - exactly the hand-off MI355X_MICROARCH.md ("Valid forms", first row of the sc1 table) measures as valid, and the
// only one that does not pay `buffer_wbl2 sc1` (a write-back of the XCD's WHOLE L2, 12 us per split on the segment kernels) per
// workgroup, which is what a release fence / release RMW lowers to.  Under the HIP memory model alone a relaxed ticket carries no
// happens-before edge, so this is an ARCHITECTURE contract, not a language one: the guard below stops any other target from
// compiling it silently, and tests/test_gpu_ops.py::test_split_handoff_stress replays the hand-over thousands of times
// against the unsplit result.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "xv_handoff_* relies on gfx950 sc1 write-through stores / L2-bypassing loads; re-derive the hand-over for this target"
#endif
__device__ __forceinline__ void xv_handoff_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float xv_handoff_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xv_handoff_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// 16-byte forms for tile-sized partials (64 KB per workgroup): `global_store_dwordx4 ... sc1` / `global_load_dwordx4 ... sc1` (the widths
// the guide's hand-off table rates fastest: a 4-byte sc1 store costs ~6x the time per byte).  HIP has no 16-byte atomic, so these are
// inline assembly: the compiler does not count them in its own s_waitcnt bookkeeping, which only ever makes ITS waits longer (vmcnt
// retires in issue order); the loads wait for themselves, the stores are drained by xv_handoff_drain.
// The s_nop behind the store is the ISA's "VMEM store of more than 64 bits, then a write of its data VGPRs" hazard (2 wait states):
// hipcc pads it for its own stores but cannot see through inline assembly, and the very next instruction here usually builds the
// next float4 in the same registers.  [measured, round 3: without it one register in sixteen of every shared tile arrived corrupted]
__device__ __forceinline__ void xv_handoff_store4(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// eight loads in flight: p + i * stride floats, i = 0..7 (a slab in [register][thread] order: every wave instruction moves 1 KB of
// consecutive bytes - [measured, round 3] with a lane's four float4 side by side, i.e. 16-byte pieces at a 64-byte stride per
// instruction, a shared tile end of the evenly scheduled GEMM cost ~19 us: every piece is a fabric transaction of its own)
__device__ __forceinline__ void xv_handoff_load8(const float* p, int stride, f32x4 (&v)[8]) {
    const float *p1 = p + stride, *p2 = p + 2 * stride, *p3 = p + 3 * stride, *p4 = p + 4 * stride, *p5 = p + 5 * stride, *p6 = p + 6 * stride,
                *p7 = p + 7 * stride;
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
        "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
        "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
        "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)
        : "memory");
}

// LDS-DMA of 16 bytes per lane (global_load_lds_dwordx4) in the address form that costs the issuing wave least: a wave-uniform 64-bit
// base in SGPRs + ONE 32-bit byte offset per lane, M0 = the LDS byte address the wave's 1 KB lands at.  [measured, round 3, tdnn2 forward at
// S1, stamps] with per-lane 64-bit addresses (a v_lshl_add_u64 + a two-VGPR address read per instruction, which is what hipcc makes of the
// builtin inside a loop - it folds a scalar base back into the vector address) the MFMA pipe was 0.867 occupied; the same loads fed from one
// hot KiB 0.883 (so it is not the memory side); this form 0.939: 488 -> 449 us, 132 -> 143 TF.  The compiler does not see the loads:
// callers wait with xv_dma_wait_all() before the barrier that publishes a stage, and every operand row must lie within 4 GB of the base.
// (M0 is a reserved register: naming it in the clobber list is rejected.  hipcc sets it immediately in front of each of its own uses and
// keeps nothing in it across statements; the kernels that use this form issue ALL their LDS-DMA through it or xv_dma16_ptr.)
__device__ __forceinline__ void xv_dma16(const float* sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_byte_addr), "v"(voff), "s"(sbase) : "memory");
}
// the same instruction with a 64-bit address per lane (the ragged K-steps, whose out-of-range pieces read a zero page that may lie more
// than 4 GB from the operand).  Every LDS-DMA of these kernels goes through one of the two forms, so hipcc never materialises M0 for a
// DMA of its own in them and cannot hoist or merge an M0 initialisation across these statements.  `lds` = a wave-uniform
// shared-memory pointer.
__device__ __forceinline__ void xv_dma16_ptr(const void* gaddr, const void* lds) {
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(la), "v"(gaddr) : "memory");
}
__device__ __forceinline__ void xv_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// the LDS byte address of a wave-uniform shared-memory pointer, as a scalar
__device__ __forceinline__ unsigned xv_lds_addr(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const void*)p);
}
// one lane per workgroup, after xv_handoff_drain + __syncthreads(): true for the workgroup that arrives last of `expected`; that
// lane also re-arms the ticket for the next launch (every other arrival has already been counted)
__device__ __forceinline__ bool xv_ticket_take(unsigned* ticket, unsigned expected) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = (t == expected - 1u);
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return last;
}

// ---- GEMM geometry shared between launchers and the engine -------------------------
#define XV_TILE_M 128
#define XV_TILE_N 128
#ifndef XV_TILE_K
#define XV_TILE_K 16
#endif
// workgroups of 256 threads that are resident per CU (== waves per SIMD); LDS and VGPR budgets of
// both GEMM kernels are sized for it, and the split policies aim at one co-resident round.
#ifndef XV_WGS_PER_CU
#define XV_WGS_PER_CU (XV_TILE_K == 16 ? 4 : 2)
#endif
#define XV_RESIDENT_WGS (256 * XV_WGS_PER_CU)

// A device page of at least `floats` (and XV_ZERO_PAGE_FLOATS) zeros (xv_gemm.hip): out-of-range rows / k of a GEMM operand are redirected
// to it (the select is on the address, loads stay unconditional); "base + k" stays inside the page for every k < K <= floats
#define XV_ZERO_PAGE_FLOATS 16384
const float* xv_zero_page(size_t floats = XV_ZERO_PAGE_FLOATS);

// Internal GEMM launchers (xv_gemm.hip).
// C[m][n] (+)= sum_k A[rowmap(m)][k] * Bt[n][k]   ("NT", both operands k-contiguous)
// rowmap(m) = (m / a_rps) * a_pitch + (m % a_rps) rows of lda floats.
struct XvGemmNT {
    const float* A; long lda; int a_rps; int a_pitch;
    const float* Bt; long ldb;
    float* C; long ldc;
    int M, N, K;
    const float* bias;      // optional, [N]
    float* bn_part;         // optional, [4][tiles_m][N]: sum, centred squares, min, max (xv_epilogue.h)
    void* ws; size_t ws_bytes;
    int co_running;         // 1: another GEMM shares the chip (the backward pass: data gradient beside weight gradient) - see xv_launch_gemm_nt
};
int xv_launch_gemm_nt(hipStream_t s, const XvGemmNT& g);
// xv_affine_dgrad / xv_affine_wgrad (include/xvector_hip.h) for a dz whose rows are ldo >= o floats apart (the pooled layer's, on the 128-byte grid)
int xv_affine_dgrad_ld(hipStream_t stream, const float* dz_pad, int ldo, int segs, int t_out, int o, int k, const float* wf, float* dx, int c,
                       void* ws, size_t ws_bytes);
int xv_affine_wgrad_ld(hipStream_t stream, const float* x, int segs, int t_in, int c_pad, int k, int c, const float* dz, int ldo, int dz_seg_pitch,
                       int dz_row0, int o, const float* kernel, float l2_scale, float* dkernel, void* ws, size_t ws_bytes);

// P[z][m][n] = sum_{r in chunk z} A[amap(r)][m] * B[bmap(r)][n]   ("TN", reduction over rows)
struct XvGemmTN {
    const float* A; long lda; int a_rps; int a_pitch;   // [R] rows mapped, M columns used
    const float* B; long ldb; int b_rps; int b_pitch;   // [R] rows mapped, N columns used
    int M, N, R;
    float* P;            // slabs [splits][M][N]
    int splits;          // chosen by xv_tn_splits (xv_tn_splits_direct when `direct`)
    int direct;          // 1: the caller stores an unsplit result straight into its destination (P = the [M][N] result when splits == 1): a short
                         // reduction over many tiles is then not split at all (the loss head's weight gradient)
};
int xv_tn_splits(int M, int N, int R);
int xv_tn_splits_direct(int M, int N, int R);
int xv_nt_shares(int tiles, int ksteps, bool stats, bool beside_wgrad, size_t ws_bytes);      // NT: shares per remaining tile of the "whole tiles + shares" schedule (0: not used)
int xv_launch_gemm_tn(hipStream_t s, const XvGemmTN& g);
int xv_launch_wgrad_reduce(hipStream_t s, const float* P, int splits, int k, int C, int c_pad, int n_in, int n_out, const float* w,
                           long ldw, float l2, float* out, long ldo);

// ---- split-precision (f16x3) GEMMs on fp16 planes (xv_gemm16.hip) ------------------------------
struct XvGemm16NT {
    const void* A; long lda; long a_plane; int a_rps; int a_pitch;   // planes [2][rows][lda] fp16
    const void* Bt; long ldb; long b_plane;
    float* C; long ldc;
    int M, N, K;
    const float* bias;
    float* bn_part;                       // optional [4][tiles_m][N]
    const unsigned* a_amax; const unsigned* b_amax;   // device: float bits of the operands' max |x| (scale source)
    // optional data-gradient epilogue (xv_epilogue.h XvBwdStats): C is d a of a BN+ReLU layer whose pre-BN tensor is bwd_z [M][N]
    const float* bwd_z; const float* bwd_scale; const float* bwd_shift; const float* bwd_mean; const float* bwd_invstd;
    float* bwd_part;                      // [tiles_m][3][N]
};
int xv_launch_gemm16_nt(hipStream_t s, const XvGemm16NT& g);
struct XvGemm16TN {
    const void* A; long lda; long a_plane; int a_pitch;
    const void* B; long ldb; long b_plane; int b_pitch;
    int rps;
    int M, N, R;
    float* P; int splits;                 // slabs [splits][M][N]; splits from xv_tn16_splits
    const unsigned* a_amax; const unsigned* b_amax;
};
int xv_tn16_splits(int M, int N, int R);
int xv_launch_gemm16_tn(hipStream_t s, const XvGemm16TN& g);

// All kernel-layout weight copies of one optimiser step in two launches (xv_elementwise.hip): the per-layer
// prep / amax / split launches (~25 of 5 us each) were 5 % of a step.
enum { XV_PREP_T32 = 0,     // wt[o][j*c_pad + c] = w[(j*C + c)*O + o]                (fp32, forward layout)
       XV_PREP_F32 = 1,     // wf[c][(k-1-j)*o_ld + o] = w[(j*C + c)*O + o]           (fp32, tap-flipped data-gradient layout)
       XV_PREP_T16 = 2,     // as T32, written as two fp16 planes scaled by pow2(*amax)
       XV_PREP_F16 = 3 };   // as F32, planes
struct XvPrepJob {
    int type, k, C, O, c_pad, o_ld;
    int tiles_x, tile0;              // 32x32 tiles per row of tiles, first global tile index
    const float* w;
    void* dst;
    long plane;                      // plane stride in elements (16-bit types)
    const unsigned* amax;
};
#define XV_PREP_PAD 8            // not a weight: rows [O][C] of w copied into [O][c_pad] with zero pad columns (the features of a step, riding on the first layer's launch)
#define XV_PREP_MAX_JOBS 32      // two layouts x (XV_MAX_FRAME_LAYERS + 2 segment + 2 attention-key layers)
struct XvPrepJobs { int n, total_tiles; XvPrepJob j[XV_PREP_MAX_JOBS]; };
#define XV_AMAX_MAX_JOBS 16
struct XvAmaxJobs { int n; const float* x[XV_AMAX_MAX_JOBS]; size_t count[XV_AMAX_MAX_JOBS]; unsigned* out[XV_AMAX_MAX_JOBS]; };
int xv_prep_add(XvPrepJobs& J, int type, const float* w, int k, int C, int O, int c_pad, int o_ld, void* dst, long plane, const unsigned* amax);
int xv_launch_weight_prep(hipStream_t s, const XvPrepJobs& J);
int xv_launch_amax_multi(hipStream_t s, const XvAmaxJobs& J);

// Segment-level BatchNorm (rows = chunks per batch, a few hundred at most) in ONE launch each way: statistics, moving
// averages, scale/shift and the activation (forward); both reductions and dz (backward).  The three-kernel forms are
// built for 25 k-row tensors; on 128 rows their launches and gaps were what the layer cost.  (xv_elementwise.hip)
#define XV_BN_SMALL_MAX_ROWS 4096
int xv_bn_small_forward(hipStream_t s, const float* z, int rows, int n, const float* gamma, const float* beta, float eps, float momentum,
                        int unbiased_moving, float* moving_mean, float* moving_var, float* mean, float* invstd, float* scale, float* shift,
                        int relu, float* a);
int xv_bn_small_backward(hipStream_t s, const float* da, const float* z, int rows, int n, const float* gamma, const float* mean,
                         const float* invstd, const float* scale, const float* shift, int relu, float* dz, float* dgamma, float* dbeta,
                         float* dbias);

// Engine-internal form of xv_bn_relu_backward_split / _pooled_split / _split_from_part: the upstream gradient is `da`, or the
// (weighted) pooling backward of (pool_out, dpool) when pool_out != null; ext_part: reductions already done by a GEMM epilogue;
// zero_amax = false: *dz_amax was zeroed by the caller (one memset per backward pass instead of one per layer).
struct XvBnBwdSplit { const float* da; const float* pool_out; const float* dpool; const float* weights; int pool_t;
                      const float* ext_part; int ext_chunks; bool zero_amax; const float* wpos; const float* pamax; };
int xv_bn_relu_backward_split_ex(hipStream_t s, const XvBnBwdSplit& x, const float* z, int segs, int t, int n, const float* gamma,
                                 const float* mean, const float* invstd, const float* scale, const float* shift, const float* zmin,
                                 const float* zmax, int relu, int pad, void* dz_planes, int ldp, size_t plane_stride, uint32_t* dz_amax,
                                 float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes);

// Segment-level GEMM C[M][N] = A[M][K] . Bt[N][K]^T, M <= 128 rows (the chunks of one batch), with the split-K sum and the consumer's
// per-column work in the same launch (xv_skinny.hip).  `ws` holds the [splits][N/32][128][32] slabs, `tickets` one zeroed uint32 per
// 32 columns (left zeroed).  Optional row term before the epilogue: acc[m][n] += (row_norm[m] > 0 ? row_coef[m] / row_norm[m] : 0) * X[m][n].
enum { XV_SK_PLAIN = 0,      // C = acc + bias
       XV_SK_BN_FWD = 1,     // C = z = acc + bias; training-mode BatchNorm over the M rows (+ activation) -> a_out; vectors and moving averages written
       XV_SK_BN_BWD = 2 };   // acc = d a of a BatchNorm(+activation) layer with pre-BN tensor z: C = dz, dgamma / dbeta / dbias / dalpha written
struct XvSkinny {
    const float* A; long lda;
    const float* Bt; long ldb;
    int M, N, K;
    float* C; long ldc;                      // also the leading dimension of z and a_out
    const float* bias;
    const float* row_coef; const float* row_norm; const float* X; long ldx;
    int epi;
    const float* gamma; const float* beta; float eps, momentum; int unbiased; float* mmean; float* mvar;
    float* mean; float* invstd; float* scale; float* shift;      // written by BN_FWD, read by BN_BWD
    int relu; const float* slope; float* a_out;
    const float* z; float* dgamma; float* dbeta; float* dbias; float* dalpha;
    void* ws; size_t ws_bytes; uint32_t* tickets;
};
int xv_launch_skinny(hipStream_t s, const XvSkinny& g);
size_t xv_skinny_tickets(int max_n);

// Statistics pooling fused with the last frame layer's BatchNorm, engine forms (xv_elementwise.hip): the forward also writes
// wpos [b][c] = the share of each chunk's frame weights on frames with an active ReLU; given that, the BatchNorm backward gets its
// two reductions in closed form from the pooled statistics instead of a pass over z (plain ReLU / no activation); amax [b][c] = each
// chunk's largest activation, which bounds |d a| for the split-precision dz scale (XvBnBwdSplit.wpos / .pamax, unit frame weights)
// frames (optional, device [b]) / shrink: chunk i pools only its first frames[i] - shrink rows (batched extraction: utterances of different
// lengths padded to t rows; shrink = the frames the frame layers consumed)
int xv_stat_pool_forward_bn_ex(hipStream_t s, const float* z, int b, int t, int c, const float* scale, const float* shift, int relu,
                               const float* weights, float* out, float* wpos, float* amax, const int32_t* frames = nullptr, int shrink = 0,
                               int ldz = 0 /* floats per row of z; 0 = c */);
// softmax over the first frames[i] - shrink scores of chunk i (weights beyond are 0); frames == nullptr: all t (xv_attention.hip)
int xv_softmax_segments_ex(hipStream_t s, const float* score, int b, int t, float* weights, const int32_t* frames, int shrink);
int xv_bn_relu_backward_pooled_ex(hipStream_t s, const float* pool_out, const float* dpool, const float* weights, const float* wpos, int b, int t,
                                  const float* z, int n, const float* gamma, const float* mean, const float* invstd, const float* scale,
                                  const float* shift, int relu, float* dz, float* dgamma, float* dbeta, float* dbias, void* ws, size_t ws_bytes,
                                  int ldz = 0 /* floats per row of z and dz; 0 = n; a pitch needs wpos (the closed form) and a plain ReLU */);

// xv_margin_softmax_rows in one launch (mean folded in through a ticket) that also writes ||x[r]|| (xv_loss.hip)
int xv_margin_softmax_rows_ex(hipStream_t s, int kind, const float* logits, int rows, int n, int ldl, const float* x, int c,
                              const int32_t* labels, float m, float lambda, float* dlogits, float* dnorm, float* row_loss,
                              float* loss_out, float* xnorm, uint32_t* ticket);

// Activation behind a BatchNorm in the layer being processed (network_relu_type, tdnn.py:24-30): y > 0 ? y : slope[c] * y.
// slope == nullptr: ReLU.  Set by the engine around a layer's calls (prelu: the layer's alpha variable, with dalpha = its gradient;
// lrelu: a constant 0.2 vector); every entry point that takes a `relu` flag reads it (xv_elementwise.hip).
struct XvActContext { const float* slope; float* dalpha; };
void xv_set_act_context(const float* slope, float* dalpha);
XvActContext xv_act_context();
// a = act(z) (da == nullptr) or dz = da * act'(z) + d alpha (segment-level tensors, rows <= XV_BN_SMALL_MAX_ROWS): a layer with the
// activation but no BatchNorm in front (tdnn7 with last_layer_no_bn)
int xv_act_small(hipStream_t s, const float* da, const float* z, int rows, int n, float* out);

// Live launch timing (xv_profile_begin/end): brackets one GEMM launch with hipEvents on its stream.
struct XvProfScope {
    hipStream_t s; int idx;
    XvProfScope(hipStream_t st, int kind, double flops);
    ~XvProfScope();
};
