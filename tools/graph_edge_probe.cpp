// Is a cross-stream edge cheaper inside a hipGraph than as hipEventRecord + hipStreamWaitEvent?  (diagnostics; not the product path)
// The chains of tools/sync_cost_probe.cpp - N kernels of ~D us on stream `s` with, between consecutive kernels, nothing ("plain"), a fork to a
// side stream ("chain": record on s, side waits, a tiny kernel on side) or a join from it ("waitside": a tiny kernel on side, record there, s
// waits) - enqueued directly, and the SAME enqueue sequence captured once (hipStreamBeginCapture on s; the side stream joins the capture through
// its first event wait and is joined back at the end) and replayed with hipGraphLaunch.  Time per kernel of the chain, events on `s` around it.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/graph_edge_probe.cpp -o tools/graph_edge_probe
//   usage: tools/graph_edge_probe [kernels=40] [us_per_kernel=10] [workgroups=256]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void spin_kernel(unsigned long long ticks, float* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float x = (float)threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) x = x * 1.0001f + 0.5f;
    if (x == 12345.678f) sink[0] = x;
}
__global__ void tiny_kernel(float* sink) { if (threadIdx.x == 9999) sink[1] = 1.f; }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 40;
    const int us = argc > 2 ? atoi(argv[2]) : 10;
    const int wgs = argc > 3 ? atoi(argv[3]) : 256;
    const unsigned long long ticks = (unsigned long long)us * 100;
    hipStream_t s, side;
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, least));
    std::vector<hipEvent_t> ev(n + 2);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    float* sink;
    CK(hipMalloc(&sink, 64));
    printf("chain of %d kernels x %d us x %d workgroups\n", n, us, wgs);

    // mode 0 plain, 1 chain (fork per kernel), 2 waitside (join per kernel)
    auto enqueue = [&](int mode, bool capturing) {
        if (capturing && mode != 0) {                       // bring the side stream into the capture
            CK(hipEventRecord(ev[n], s));
            CK(hipStreamWaitEvent(side, ev[n], 0));
        }
        for (int i = 0; i < n; ++i) {
            if (mode == 2) {
                hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, sink);
                CK(hipEventRecord(ev[i], side));
                CK(hipStreamWaitEvent(s, ev[i], 0));
            }
            hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, s, ticks, sink);
            if (mode == 1) {
                CK(hipEventRecord(ev[i], s));
                CK(hipStreamWaitEvent(side, ev[i], 0));
                hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, sink);
            }
        }
        if (capturing && mode != 0) {                       // ... and back
            CK(hipEventRecord(ev[n + 1], side));
            CK(hipStreamWaitEvent(s, ev[n + 1], 0));
        }
    };
    const char* names[] = {"plain", "chain", "waitside"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int graph = 0; graph < 2; ++graph) {
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            size_t nodes = 0;
            if (graph) {
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                enqueue(mode, true);
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphGetNodes(g, nullptr, &nodes));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s));                  // warm
                CK(hipDeviceSynchronize());
            }
            std::vector<float> ms;
            for (int rep = 0; rep < 9; ++rep) {
                CK(hipEventRecord(t0, s));
                if (graph) CK(hipGraphLaunch(ge, s));
                else enqueue(mode, false);
                CK(hipEventRecord(t1, s));
                CK(hipEventSynchronize(t1));
                CK(hipStreamSynchronize(side));
                float t;
                CK(hipEventElapsedTime(&t, t0, t1));
                ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("%-9s %-7s %7.2f us per kernel (min %.2f)%s\n", names[mode], graph ? "graph" : "streams", 1e3 * ms[ms.size() / 2] / n, 1e3 * ms[0] / n,
                   graph ? (std::string("   ") + std::to_string(nodes) + " nodes").c_str() : "");
            fflush(stdout);
            if (ge) CK(hipGraphExecDestroy(ge));
            if (g) CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
