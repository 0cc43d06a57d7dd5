// What the chip sustains on fp32-input MFMA loops (diagnostics; not the product path).  MI355X lowers its clock under MFMA load on
// random data (MI355X_MICROARCH.md, DVFS give-back), so the roof a GEMM kernel can reach is "FLOP per cycle x the clock the loop holds".
// This program measures that product for the two fp32-input shapes, with and without the LDS fragment reads a GEMM needs:
//   shape 0: v_mfma_f32_32x32x2_f32, wave tile 64x64 = 2x2 accumulators
//   shape 1: v_mfma_f32_16x16x4_f32, wave tile 64x64 = 4x4 accumulators
//   lds 0: operands stay in registers; lds 1: every operand fragment re-read from LDS by ds_read_b128 (one per 4 k-steps, as the GEMM does)
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_clock_probe.cpp -o tools/mfma_clock_probe
//   usage: tools/mfma_clock_probe [ms_per_config=60]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one "K-step" = 16 k = the work of one BK=16 GEMM step on a 64x64 wave tile: 32 MFMAs (32x32x2) or 64 MFMAs (16x16x4)
template <int SHAPE, int LDS>
__global__ __launch_bounds__(256, 4) void mfma_loop(const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[2 * 128 * 16];      // A [128][16] | B [128][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 128 * 16; i += 256) sm[i] = src[(blockIdx.x & 63) * 4096 + i];
    __syncthreads();
    const int wr = wave >> 1, wc = wave & 1;
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (SHAPE == 0) {
        const int li = lane & 31, lh = lane >> 5;
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        const float* sa = sm + (wr * 64 + li) * 16;
        const float* sb = sm + 128 * 16 + (wc * 64 + li) * 16;
        f32x4 af[2][2], bf[2][2];
        for (int q = 0; q < 2; ++q) {
            const int pos = (((2 * q + lh) ^ ((li >> 2) & 3)) << 2);
            af[q][0] = *(const f32x4*)(sa + pos); af[q][1] = *(const f32x4*)(sa + 32 * 16 + pos);
            bf[q][0] = *(const f32x4*)(sb + pos); bf[q][1] = *(const f32x4*)(sb + 32 * 16 + pos);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (LDS) {
                    int pos = (((2 * q + lh) ^ ((li >> 2) & 3)) << 2);
                    asm volatile("" : "+v"(pos));       // opaque: the reads stay inside the loop
                    af[q][0] = *(const f32x4*)(sa + pos); af[q][1] = *(const f32x4*)(sa + 32 * 16 + pos);
                    bf[q][0] = *(const f32x4*)(sb + pos); bf[q][1] = *(const f32x4*)(sb + 32 * 16 + pos);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][0][e], bf[q][0][e], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][0][e], bf[q][1][e], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][1][e], bf[q][0][e], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][1][e], bf[q][1][e], acc[1][1], 0, 0, 0);
                }
            }
            if (!LDS) asm volatile("" : "+v"(af[0][0]), "+v"(bf[0][0]));
        }
        float s = 0.f;
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
        out[blockIdx.x * 256 + tid] = s;
    } else {
        const int li = lane & 15, lq = lane >> 4;
        f32x4 acc[4][4];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* sa = sm + (wr * 64 + li) * 16 + lq * 4;
        const float* sb = sm + 128 * 16 + (wc * 64 + li) * 16 + lq * 4;
        f32x4 af[4], bf[4];
        for (int a = 0; a < 4; ++a) { af[a] = *(const f32x4*)(sa + a * 16 * 16); bf[a] = *(const f32x4*)(sb + a * 16 * 16); }
        for (int it = 0; it < iters; ++it) {
            if (LDS) {
                int pos = 0;
                asm volatile("" : "+v"(pos));
#pragma unroll
                for (int a = 0; a < 4; ++a) { af[a] = *(const f32x4*)(sa + pos + a * 16 * 16); bf[a] = *(const f32x4*)(sb + pos + a * 16 * 16); }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
            if (!LDS) asm volatile("" : "+v"(af[0]), "+v"(bf[0]));
        }
        float s = 0.f;
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
        out[blockIdx.x * 256 + tid] = s;
    }
    if (tid == 0) {
        stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int SHAPE, int LDS>
static void run(const char* name, int wgs_per_cu, double ms_target, const float* src, float* out, unsigned long long* stamps, float data_scale) {
    const int grid = 256 * wgs_per_cu;
    const double flop_per_iter_wave = 2.0 * 64 * 64 * 16;
    int iters = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    // calibrate one launch to ~2 ms, then run back-to-back launches for ms_target and report the last half
    hipLaunchKernelGGL((mfma_loop<SHAPE, LDS>), dim3(grid), dim3(256), 0, 0, src, out, stamps, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((mfma_loop<SHAPE, LDS>), dim3(grid), dim3(256), 0, 0, src, out, stamps, iters);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    iters = (int)(iters * 2.0 / ms);
    const int launches = (int)(ms_target / 2.0) + 2;
    for (int i = 0; i < launches / 2; ++i) hipLaunchKernelGGL((mfma_loop<SHAPE, LDS>), dim3(grid), dim3(256), 0, 0, src, out, stamps, iters);
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < launches / 2; ++i) hipLaunchKernelGGL((mfma_loop<SHAPE, LDS>), dim3(grid), dim3(256), 0, 0, src, out, stamps, iters);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> st(grid * 2);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    double clk = 0, cyc = 0;
    for (int i = 0; i < grid; ++i) { clk += (double)st[2 * i] / (double)st[2 * i + 1] * 0.1; cyc += (double)st[2 * i]; }
    clk /= grid; cyc /= grid;
    const double flops = flop_per_iter_wave * 4 * grid * (double)iters * (launches / 2);
    const double tf = flops / (ms * 1e-3) / 1e12;
    // cycles per SIMD the MFMAs of one workgroup need: iters * 16 k / (2 or 4 k per MFMA) * (4 or 16 blocks) * (64 or 32 cycles) = iters * 2048
    const double busy = (double)iters * 2048.0 * wgs_per_cu / cyc;
    printf("%-34s data x%-3g %d wg/CU  %7.1f TF  in-kernel clock %.3f GHz  MFMA-pipe busy %.3f  (%.0f%% of 157.3)\n", name, data_scale, wgs_per_cu, tf, clk,
           busy, tf / 157.3 * 100);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

int main(int argc, char** argv) {
    const double ms_target = argc > 1 ? atof(argv[1]) : 60.0;
    float *src, *out; unsigned long long* stamps;
    CK(hipMalloc((void**)&src, 64 * 4096 * sizeof(float)));
    CK(hipMalloc((void**)&out, 1024 * 256 * sizeof(float)));
    CK(hipMalloc((void**)&stamps, 1024 * 2 * 8));
    for (float scale : {1.f, 0.f}) {
        std::vector<float> h(64 * 4096);
        std::mt19937 g(1); std::normal_distribution<float> d(0.f, 1.f);
        for (auto& v : h) v = d(g) * scale;
        CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (int w : {1, 2, 3, 4}) {
            run<0, 0>("32x32x2, operands in registers", w, ms_target, src, out, stamps, scale);
            run<1, 0>("16x16x4, operands in registers", w, ms_target, src, out, stamps, scale);
            run<0, 1>("32x32x2 + ds_read_b128 fragments", w, ms_target, src, out, stamps, scale);
            run<1, 1>("16x16x4 + ds_read_b128 fragments", w, ms_target, src, out, stamps, scale);
        }
    }
    return 0;
}
