// What a cross-stream hand-over costs the stream that hands over (diagnostics; not the product path).
// A chain of N identical kernels of ~D microseconds each on stream `s`; between consecutive kernels one of:
//   plain      nothing
//   record     hipEventRecord(ev, s)                                  (nobody waits)
//   chain      hipEventRecord(ev, s); hipStreamWaitEvent(side, ev); a tiny kernel on side
//   waitdone   hipStreamWaitEvent(s, ev_done)                         (ev_done completed long ago on the side stream)
//   waitside   a tiny kernel on side; hipEventRecord(ev, side); hipStreamWaitEvent(s, ev)
//   extstop    the kernel launched with hipExtLaunchKernelGGL(stopEvent = ev); hipStreamWaitEvent(side, ev); a tiny kernel on side
//   writeval   hipStreamWriteValue32(s, sig, i)
//   flagwait   the kernel's last workgroup stores i to signal memory; hipStreamWaitValue32(side, sig, i, Gte); a tiny kernel on side
//   nextflag   the NEXT kernel on `s` stores i to signal memory as it starts; hipStreamWaitValue32(side, sig, i, Gte); a tiny kernel on side
//   record_nf / chain_nf   record / chain with events created hipEventDisableSystemFence
// Prints the time per kernel of the chain on `s` (hipEvent timing around the whole chain, median of several repetitions).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sync_cost_probe.cpp -o tools/sync_cost_probe
//   usage: tools/sync_cost_probe [kernels=40] [us_per_kernel=10] [workgroups=256] [mode mask=127: bit m = run mode m; flagwait / nextflag / record_nf / chain_nf (bits 7-10) and the start-flag check (2048) on request]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// spins for `ticks` of the 100 MHz real-time counter; the last workgroup to finish stores `value` to *flag (if flag != nullptr)
__global__ __launch_bounds__(256) void spin_kernel(unsigned long long ticks, unsigned* counter, unsigned* flag, unsigned value, float* sink,
                                                   unsigned* start_flag) {
    // start_flag: the first thread of the launch announces "everything in front of me on this stream is complete" (the kernel starts behind
    // its predecessor's end-of-kernel release) - a hand-over of the PREVIOUS kernel's results that costs this stream no packet
    if (start_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(start_flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float x = (float)threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) x = x * 1.0001f + 0.5f;
    if (x == 12345.678f) sink[0] = x;
    if (flag) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x - 1) {
                __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

__global__ void tiny_kernel(float* sink) { if (threadIdx.x == 9999) sink[1] = 1.f; }

// correctness of the stop-event hand-over: the producer spins, THEN fills buf with `value`; the consumer (other stream, behind the event)
// counts the elements that differ
__global__ __launch_bounds__(256) void fill_late_kernel(unsigned long long ticks, unsigned* buf, size_t n, unsigned value) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) buf[i] = value;
}
__global__ __launch_bounds__(256) void check_kernel(const unsigned* buf, size_t n, unsigned value, unsigned* errors) {
    unsigned bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) bad += buf[i] != value;
    if (bad) atomicAdd(errors, bad);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 40;
    const int us = argc > 2 ? atoi(argv[2]) : 10;
    const int wgs = argc > 3 ? atoi(argv[3]) : 256;
    const int mask = argc > 4 ? atoi(argv[4]) : 127;
    const unsigned long long ticks = (unsigned long long)us * 100;
    hipStream_t s, side;
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, least));
    std::vector<hipEvent_t> ev(n);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t ev_done, t0, t1;
    CK(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    float* sink;
    unsigned* counter;
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&counter, 64));
    CK(hipMemset(counter, 0, 64));
    unsigned* sig = nullptr;
    int can_wait = 0;
    CK(hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0));
    if (can_wait && hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory) != hipSuccess) { sig = nullptr; (void)hipGetLastError(); }      // (8 bytes exactly: any other size is refused)
    if (sig) { sig[0] = 0; sig[1] = 0; }
    printf("chain of %d kernels x %d us x %d workgroups; stream wait-value support %d, signal memory %s\n", n, us, wgs, can_wait, sig ? "yes" : "no");
    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, sink);
    CK(hipEventRecord(ev_done, side));
    CK(hipDeviceSynchronize());

    if (mask & 256) {      // stop-event hand-over: does the waiter see everything the kernel wrote?
        const size_t nbuf = 16u << 20;
        unsigned *buf, *errors;
        CK(hipMalloc(&buf, nbuf * 4));
        CK(hipMalloc(&errors, 4));
        CK(hipMemset(errors, 0, 4));
        CK(hipMemset(buf, 0, nbuf * 4));
        CK(hipDeviceSynchronize());
        for (unsigned it = 1; it <= 200; ++it) {
            hipExtLaunchKernelGGL(fill_late_kernel, dim3(1024), dim3(256), 0, s, nullptr, ev[it % n], 0, ticks, buf, nbuf, it);
            hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, sink);       // more work on s behind the producer
            CK(hipStreamWaitEvent(side, ev[it % n], 0));
            hipLaunchKernelGGL(check_kernel, dim3(1024), dim3(256), 0, side, buf, nbuf, it, errors);
            CK(hipEventRecord(ev_done, side));
            CK(hipStreamWaitEvent(s, ev_done, 0));                                 // WAR: the next fill waits for the check
        }
        CK(hipDeviceSynchronize());
        unsigned herr = 0;
        CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost));
        printf("stop-event hand-over check: 200 rounds x 64 MB, %u stale elements seen by the waiter\n", herr);
    }
    std::vector<hipEvent_t> evnf(n);
    for (auto& e : evnf) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    if (sig && (mask & 2048)) {      // does the waiter behind a start flag see everything the kernel IN FRONT of the flagging kernel wrote?
        const size_t nbuf = 16u << 20;
        unsigned *buf, *errors, herr = 0;
        CK(hipMalloc(&buf, nbuf * 4));
        CK(hipMalloc(&errors, 4));
        CK(hipMemset(errors, 0, 4));
        unsigned ep = 1000000;
        for (int round = 0; round < 200; ++round) {
            ++ep;
            hipLaunchKernelGGL(fill_late_kernel, dim3(1024), dim3(256), 0, s, 500ull, buf, nbuf, ep);
            hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, s, 200ull, counter, (unsigned*)nullptr, ep, sink, sig);
            CK(hipStreamWaitValue32(side, sig, ep, hipStreamWaitValueGte, 0xFFFFFFFFu));
            hipLaunchKernelGGL(check_kernel, dim3(1024), dim3(256), 0, side, buf, nbuf, ep, errors);
            CK(hipEventRecord(ev[0], side));
            CK(hipStreamWaitEvent(s, ev[0], 0));      // (the next round's fill waits for this round's check)
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost));
        printf("start-flag hand-over check: 200 rounds x 64 MB, %u stale elements seen by the waiter\n", herr);
    }
    const char* names[] = {"plain", "record", "chain", "waitdone", "waitside", "extstop", "writeval", "flagwait", "nextflag", "record_nf", "chain_nf"};
    unsigned epoch = 0;
    for (int mode = 0; mode < 11; ++mode) {
        if (!((mask >> mode) & 1)) continue;
        if ((mode == 6 || mode == 7 || mode == 8) && !sig) { printf("%-9s skipped (no signal memory)\n", names[mode]); continue; }
        std::vector<float> ms;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(t0, s));
            for (int i = 0; i < n; ++i) {
                ++epoch;
                if (mode == 3) CK(hipStreamWaitEvent(s, ev_done, 0));
                if (mode == 4) {
                    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, sink);
                    CK(hipEventRecord(ev[i], side));
                    CK(hipStreamWaitEvent(s, ev[i], 0));
                }
                if (mode == 5) {
                    hipExtLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, s, nullptr, ev[i], 0, ticks, counter, (unsigned*)nullptr, 0u, sink, (unsigned*)nullptr);
                } else if (mode == 7) {
                    hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, s, ticks, counter, sig, epoch, sink, (unsigned*)nullptr);
                } else if (mode == 8) {
                    hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, s, ticks, counter, (unsigned*)nullptr, epoch, sink, sig);
                } else {
                    hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, s, ticks, counter, (unsigned*)nullptr, 0u, sink, (unsigned*)nullptr);
                }
                if (mode == 1 || mode == 2) CK(hipEventRecord(ev[i], s));
                if (mode == 2 || mode == 5) {
                    CK(hipStreamWaitEvent(side, ev[i], 0));
                    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, sink);
                }
                if (mode == 6) CK(hipStreamWriteValue32(s, sig, epoch, 0));
                if (mode == 9 || mode == 10) CK(hipEventRecord(evnf[i], s));
                if (mode == 10) {
                    CK(hipStreamWaitEvent(side, evnf[i], 0));
                    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, sink);
                }
                if (mode == 7 || mode == 8) {
                    CK(hipStreamWaitValue32(side, sig, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, sink);
                }
            }
            CK(hipEventRecord(t1, s));
            CK(hipEventSynchronize(t1));
            CK(hipStreamSynchronize(side));
            float t;
            CK(hipEventElapsedTime(&t, t0, t1));
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-9s %7.2f us per kernel (min %.2f)\n", names[mode], 1e3 * ms[ms.size() / 2] / n, 1e3 * ms[0] / n);
        fflush(stdout);
    }
    return 0;
}
