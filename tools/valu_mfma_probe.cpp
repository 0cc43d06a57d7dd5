// Do the matrix pipe and the vector ALU of a gfx950 SIMD run fp32 work at the same time?  (diagnostics; not the product path)
// Both are quoted at 157.3 TFLOP/s fp32 (v_mfma_f32_32x32x2_f32: 4 096 FLOP per wave instruction, one per 64 cycles and SIMD;
// v_pk_fma_f32: 256 FLOP per wave instruction, one per 4 cycles and SIMD).  An MFMA occupies the issue port for one slot and then runs
// in the matrix pipe; this program measures what a loop sustains that issues NV independent v_pk_fma_f32 behind every MFMA -
// NV = 0 (matrix only), no MFMA (vector only), and mixes - at 1, 2 and 4 waves per SIMD, with the clock the loop holds.
// Operands stay in registers: this is the ceiling of the two pipes together, not a GEMM.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_mfma_probe.cpp -o tools/valu_mfma_probe
//   usage: tools/valu_mfma_probe [ms_per_config=40]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one loop trip = 4 x (NM MFMAs, NV packed FMAs)
template <int NM, int NV>
__global__ __launch_bounds__(256, 4) void mix_loop(const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters) {
    const int tid = threadIdx.x;
    float a = src[tid], b = src[256 + tid];
    f32x2 va = {src[512 + tid], src[768 + tid]}, vb = {src[1024 + tid], src[1280 + tid]};
    f32x16 macc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) macc[j][r] = 0.f;
    f32x2 vacc[16];
    for (int k = 0; k < 16; ++k) vacc[k] = f32x2{0.f, 0.f};
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // volatile asm keeps the order (hipcc would also split most packed FMAs into two v_fma_f32: half the rate)
            if (NM) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(macc[j]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int k = (j * NV + v) & 15;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(vacc[k]) : "v"(va), "v"(vb));
            }
        }
        asm volatile("" : "+v"(a), "+v"(va));
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += macc[j][r];
    for (int k = 0; k < 16; ++k) s += vacc[k].x + vacc[k].y;
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) {
        stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int NM, int NV>
static void run(int wgs_per_cu, double ms_target, const float* src, float* out, unsigned long long* stamps) {
    const int grid = 256 * wgs_per_cu;
    int iters = 4000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((mix_loop<NM, NV>), dim3(grid), dim3(256), 0, 0, src, out, stamps, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((mix_loop<NM, NV>), dim3(grid), dim3(256), 0, 0, src, out, stamps, iters);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    iters = (int)(iters * 2.0 / ms) + 1;
    const int launches = (int)(ms_target / 2.0) + 2;
    for (int i = 0; i < launches / 2; ++i) hipLaunchKernelGGL((mix_loop<NM, NV>), dim3(grid), dim3(256), 0, 0, src, out, stamps, iters);
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < launches / 2; ++i) hipLaunchKernelGGL((mix_loop<NM, NV>), dim3(grid), dim3(256), 0, 0, src, out, stamps, iters);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> st(grid * 2);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    double clk = 0;
    for (int i = 0; i < grid; ++i) clk += (double)st[2 * i] / (double)st[2 * i + 1] * 0.1;
    clk /= grid;
    const double waves = (double)grid * 4, trips = (double)iters * (launches / 2);
    const double mf = waves * trips * 4 * NM * 4096.0, vf = waves * trips * 4 * NV * 256.0;
    const double sec = ms * 1e-3;
    printf("  %d MFMA : %2d pk_fma, %d waves/SIMD: matrix %6.1f TF + vector %6.1f TF = %6.1f TF   clock %.3f GHz   (%.1f cycles per MFMA slot)\n", NM, NV,
           wgs_per_cu, mf / sec * 1e-12, vf / sec * 1e-12, (mf + vf) / sec * 1e-12, clk,
           clk * 1e9 * sec / (trips * 4 * wgs_per_cu));
    fflush(stdout);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

int main(int argc, char** argv) {
    const double ms = argc > 1 ? atof(argv[1]) : 40.0;
    float *src, *out;
    unsigned long long* stamps;
    std::vector<float> h(2048);
    srand(1);
    for (auto& x : h) x = (float)(rand() & 0xffff) / 65536.f - 0.5f;
    CK(hipMalloc(&src, h.size() * 4));
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 1024 * 256 * 4));
    CK(hipMalloc(&stamps, 1024 * 2 * 8));
    for (int w : {1, 2, 4}) {
        printf("%d waves per SIMD\n", w);
        run<1, 0>(w, ms, src, out, stamps);
        run<0, 16>(w, ms, src, out, stamps);
        run<1, 4>(w, ms, src, out, stamps);
        run<1, 8>(w, ms, src, out, stamps);
        run<1, 12>(w, ms, src, out, stamps);
        run<1, 15>(w, ms, src, out, stamps);
        run<1, 16>(w, ms, src, out, stamps);
    }
    return 0;
}
