// Stand-alone micro-benchmark of the fp32 MFMA GEMM entry points of libxvector_hip.so (diagnostics, not a test, not the product path).
// No Python / torch: starts in milliseconds, so one gpurun call can compare many variant libraries on the same box.
//   build:  hipcc -O2 -std=c++17 tools/gemm_probe.cpp -o tools/gemm_probe -ldl
//   usage:  tools/gemm_probe <libxvector_hip.so> [B=128] [T=200] [reps=20] [stamp_dump.json|-] [stamp_layer=tdnn2]
// Prints per layer (tdnn2..tdnn5 shapes of model/tdnn.py:57-127) forward / data-gradient / weight-gradient time and TFLOP/s.
// If the library exports xv_debug_read_stamps (built with -DXV_DIAG=1|2, csrc/xv_diag.h) the per-workgroup stamps of the last tdnn2 (or [stamp_layer]) forward
// launch are analysed: workgroups per CU, per-phase cycles, start / end skew.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int (*fwd_t)(void*, const float*, int, int, int, int, const float*, const float*, float*, int, int, float*, void*, size_t);
typedef int (*dgrad_t)(void*, const float*, int, int, int, int, const float*, float*, int, void*, size_t);
typedef int (*wgrad_t)(void*, const float*, int, int, int, int, int, const float*, int, int, int, const float*, float, float*, void*, size_t);
typedef size_t (*wsb_t)(int, int, int);
typedef const char* (*err_t)(void);
typedef int (*stamps_t)(void*, size_t);

static float* dev_random(size_t n, float scale, unsigned seed) {
    std::vector<float> h(n);
    std::mt19937 g(seed);
    std::normal_distribution<float> d(0.f, 1.f);
    for (size_t i = 0; i < n; ++i) h[i] = d(g) * scale;
    float* p; CK(hipMalloc((void**)&p, n * sizeof(float)));
    CK(hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return p;
}

// [measured, round 3] a kernel with LDS fragment reads runs 5-15 % slower for the first ~50 ms after the chip was idle (launch periods
// of one back-to-back burst: 549 -> 519 us, 488 -> 459 us; an MFMA-only body is flat), so every configuration is warmed with >= 150 ms of
// back-to-back launches before the timed burst, which follows WITHOUT a host synchronisation in between.
template <class F> static double time_us(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 8; ++i) f();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float wms; CK(hipEventElapsedTime(&wms, a, b));
    const int warm = (int)(150.0 / (wms / 8.0 + 1e-3)) + 1;
    for (int i = 0; i < warm; ++i) f();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s lib [B] [T] [reps] [stamps.json]\n", argv[0]); return 2; }
    const int B = argc > 2 ? atoi(argv[2]) : 128, T = argc > 3 ? atoi(argv[3]) : 200, reps = argc > 4 ? atoi(argv[4]) : 20;
    const char* dump = (argc > 5 && strcmp(argv[5], "-")) ? argv[5] : nullptr;
    const char* stamp_layer = argc > 6 ? argv[6] : "tdnn2";
    const float scale = getenv("XV_DATA_SCALE") ? (float)atof(getenv("XV_DATA_SCALE")) : 1.f;
    const char* only = getenv("XV_PROBE_ONLY"); if (only && !*only) only = nullptr;         // e.g. "tdnn2" : restrict the layers
    const char* ops = getenv("XV_PROBE_OPS"); if (ops && !*ops) ops = nullptr;           // subset of "fdw" (forward, dgrad, wgrad)
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    fwd_t fwd = (fwd_t)dlsym(h, "xv_affine_forward");
    dgrad_t dgrad = (dgrad_t)dlsym(h, "xv_affine_dgrad");
    wgrad_t wgrad = (wgrad_t)dlsym(h, "xv_affine_wgrad");
    wsb_t wsb = (wsb_t)dlsym(h, "xv_op_workspace_bytes");
    err_t lerr = (err_t)dlsym(h, "xv_last_error");
    stamps_t stamps = (stamps_t)dlsym(h, "xv_debug_read_stamps");
    stamps_t stamps_prev = (stamps_t)dlsym(h, "xv_debug_read_stamps_prev");
    if (!fwd || !dgrad || !wgrad || !wsb) { fprintf(stderr, "missing symbols\n"); return 1; }

    struct L { const char* name; int t_in, c, k, o; bool extra; };
    // tdnn1..5: model/tdnn.py:35-127; XV_PROBE_EXTRA=1 adds the attention key network of nnet_conf/*_tdnn4_att.json (pooling.py:84-96: 512 -> 1500 ->
    // 1500 on the pooled frames) and a k = 3 layer of the extended stack (BASELINE configs[4]: contexts 5 1 3 1 3 1 3 1 1 1)
    const L layers[] = {{"tdnn1", T, 32, 5, 512, false}, {"tdnn2", T - 4, 512, 5, 512, false}, {"tdnn3", T - 8, 512, 7, 512, false},
                        {"tdnn4", T - 14, 512, 1, 512, false}, {"tdnn5", T - 14, 512, 1, 1500, false},
                        {"att_key0", T - 14, 512, 1, 1500, true}, {"att_key1", T - 14, 1500, 1, 1500, true}, {"ext_k3", T - 4, 512, 3, 512, true}};
    const bool extra = getenv("XV_PROBE_EXTRA") && atoi(getenv("XV_PROBE_EXTRA"));
    double sum_us = 0, sum_fl = 0;
    for (const L& l : layers) {
        if (l.extra && !extra) continue;
        if (only && !strstr(only, l.name)) continue;
        const int segs = l.k > 1 ? B : B * l.t_in, tin = l.k > 1 ? l.t_in : 1, tout = tin - l.k + 1;
        const long rows_in = (long)segs * tin, rows_out = (long)segs * tout;
        float* x = dev_random(rows_in * l.c, scale, 1);
        float* kern = dev_random((size_t)l.k * l.c * l.o, 0.05f * scale, 2);      // TF layout [k][c][o]; used as-is for every operand role
        float* wt = dev_random((size_t)l.o * l.k * l.c, 0.05f * scale, 3);         // forward layout [o][k*c]
        float* wf = dev_random((size_t)l.c * l.k * l.o, 0.05f * scale, 4);         // data-gradient layout [c][k*o]
        float* bias = dev_random(l.o, scale, 5);
        const int pitch = tout + 2 * (l.k - 1);
        float* dzp = dev_random((size_t)segs * pitch * l.o, scale, 6);
        float *z, *dx, *dk, *part; void* ws;
        CK(hipMalloc((void**)&z, rows_out * l.o * sizeof(float)));
        CK(hipMalloc((void**)&dx, (size_t)segs * (tout + l.k - 1) * l.c * sizeof(float)));
        CK(hipMalloc((void**)&dk, (size_t)l.k * l.c * l.o * sizeof(float)));
        const int tiles_m = (int)((rows_out + 127) / 128);
        CK(hipMalloc((void**)&part, (size_t)4 * tiles_m * l.o * sizeof(float)));
        size_t wsn = std::max(wsb((int)rows_out, l.k * l.c, l.o), (size_t)1024 * 128 * 128 * sizeof(float) + (1u << 20));   // up to 1 024 weight-gradient slabs
        CK(hipMalloc(&ws, wsn));
        const double fl = 2.0 * rows_out * l.k * l.c * l.o;
        const double fl2 = 2.0 * segs * (tout + l.k - 1) * (double)l.k * l.o * l.c;
        auto chk = [&](int rc, const char* what) { if (rc) { fprintf(stderr, "%s %s failed: %s\n", l.name, what, lerr ? lerr() : "?"); exit(1); } };
        if (!ops || strchr(ops, 'f')) {
            double us = time_us([&] { chk(fwd(nullptr, x, segs, tin, l.c, l.k, wt, bias, z, l.o, l.o, part, ws, wsn), "fwd"); }, reps);
            printf("%s fwd   M=%6ld K=%5d N=%5d  %8.1f us  %6.1f TF\n", l.name, rows_out, l.k * l.c, l.o, us, fl / us / 1e6);
            sum_us += us; sum_fl += fl;
        }
        if (!ops || strchr(ops, 'd')) {
            double us = time_us([&] { chk(dgrad(nullptr, dzp, segs, tout, l.o, l.k, wf, dx, l.c, ws, wsn), "dgrad"); }, reps);
            printf("%s dgrad M=%6ld K=%5d N=%5d  %8.1f us  %6.1f TF\n", l.name, (long)segs * (tout + l.k - 1), l.k * l.o, l.c, us, fl2 / us / 1e6);
            sum_us += us; sum_fl += fl2;
        }
        if (!ops || strchr(ops, 'w')) {
            double us = time_us([&] { chk(wgrad(nullptr, x, segs, tin, l.c, l.k, l.c, dzp, pitch, l.k - 1, l.o, kern, 1e-2f, dk, ws, wsn), "wgrad"); }, reps);
            printf("%s wgrad M=%6d N=%5d R=%6ld  %8.1f us  %6.1f TF (incl. slab sum)\n", l.name, l.k * l.c, l.o, rows_out, us, fl / us / 1e6);
            sum_us += us; sum_fl += fl;
        }
        if (getenv("XV_PROBE_PERIODS") && !strcmp(l.name, "tdnn2")) {
            // launch-to-launch periods inside one back-to-back burst (an event after every launch)
            std::vector<hipEvent_t> ev(reps + 1);
            for (auto& e : ev) CK(hipEventCreate(&e));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(ev[0], 0));
            for (int i = 0; i < reps; ++i) { chk(fwd(nullptr, x, segs, tin, l.c, l.k, wt, bias, z, l.o, l.o, part, ws, wsn), "fwd"); CK(hipEventRecord(ev[i + 1], 0)); }
            CK(hipDeviceSynchronize());
            printf("  periods (us):");
            for (int i = 0; i < reps; ++i) { float ms; CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); printf(" %.0f", ms * 1e3); }
            printf("\n");
            for (auto& e : ev) CK(hipEventDestroy(e));
        }
        if (stamps && !strcmp(l.name, stamp_layer)) {
            // the stamps of the LAST launch of a back-to-back sequence: the clock the chip sustains, not the one an idle chip starts with
            for (int i = 0; i < reps; ++i) chk(fwd(nullptr, x, segs, tin, l.c, l.k, wt, bias, z, l.o, l.o, part, ws, wsn), "fwd");
            CK(hipDeviceSynchronize());
            const int nwg = std::min(4096, tiles_m * ((l.o + 127) / 128));
            std::vector<unsigned long long> st((size_t)nwg * 8);
            chk(stamps(st.data(), st.size() * sizeof(unsigned long long)), "stamps");
            // group by CU = (xcc, se, sh?, cu): HW_ID bits cu 11:8, sh 12, se 15:13
            std::map<unsigned, std::vector<int>> cu;
            unsigned long long t_min = ~0ull, t_max = 0, rt_min = ~0ull, rt_max = 0;
            std::vector<double> dur, pro, loop, epi, stall;
            for (int w = 0; w < nwg; ++w) {
                const unsigned long long* s = &st[(size_t)w * 8];
                unsigned hw = (unsigned)(s[4] & 0xffffffffu), xcc = (unsigned)(s[4] >> 32) & 0xf;
                unsigned key = (xcc << 16) | (hw & 0xff00);
                cu[key].push_back(w);
                rt_min = std::min(rt_min, s[5]); rt_max = std::max(rt_max, s[6]);
                dur.push_back((double)(s[3] - s[0])); pro.push_back((double)(s[1] - s[0])); loop.push_back((double)(s[2] - s[1]));
                epi.push_back((double)(s[3] - s[2])); stall.push_back((double)s[7]);
                (void)t_min; (void)t_max;
            }
            auto stat = [](std::vector<double> v, const char* nm) {
                std::sort(v.begin(), v.end());
                double m = 0; for (double x : v) m += x; m /= v.size();
                printf("  %-10s mean %10.0f  min %10.0f  p50 %10.0f  p90 %10.0f  max %10.0f cycles\n", nm, m, v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back());
            };
            printf("stamps: %d workgroups on %zu CUs; kernel span %.1f us (realtime 100 MHz)\n", nwg, cu.size(), (rt_max - rt_min) / 100.0);
            std::map<int, int> hist;
            for (auto& kv : cu) hist[(int)kv.second.size()]++;
            for (auto& kv : hist) printf("  CUs with %d workgroups: %d\n", kv.first, kv.second);
            stat(dur, "total"); stat(pro, "prologue"); stat(loop, "mainloop"); stat(epi, "epilogue"); stat(stall, "wait+barrier");
            // shader clock: cycles per realtime tick over the workgroups
            double clk = 0; int nclk = 0;
            for (int w = 0; w < nwg; ++w) { const unsigned long long* s = &st[(size_t)w * 8]; if (s[6] > s[5]) { clk += (double)(s[3] - s[0]) / (double)(s[6] - s[5]) * 0.1; ++nclk; } }
            printf("  in-kernel clock %.3f GHz\n", clk / std::max(1, nclk));
            {   // MFMA-pipe occupancy per CU: every tile needs ksteps * (BK/2 k-pairs) * 4 MFMAs * 64 cycles on each of the 4 SIMDs
                const double tile_cycles = (double)(l.k * l.c / 2) * 4 * 64;
                std::vector<double> busy;
                for (auto& kv : cu) {
                    unsigned long long a = ~0ull, b = 0;
                    for (int w : kv.second) { a = std::min(a, st[(size_t)w * 8 + 0]); b = std::max(b, st[(size_t)w * 8 + 3]); }
                    busy.push_back(tile_cycles * kv.second.size() / (double)(b - a));
                }
                std::sort(busy.begin(), busy.end());
                double m = 0; for (double x : busy) m += x; m /= busy.size();
                printf("  MFMA-pipe occupancy per CU (tile MFMA cycles / CU span): mean %.3f  min %.3f  max %.3f\n", m, busy.front(), busy.back());
            }
            // per-CU picture of the first few CUs: start offsets and durations of their workgroups (realtime, us)
            int shown = 0;
            for (auto& kv : cu) {
                if (shown++ >= 6) break;
                printf("  CU %06x:", kv.first);
                for (int w : kv.second) { const unsigned long long* s = &st[(size_t)w * 8]; printf(" [wg %d simd %u: %.1f..%.1f us]", w, (unsigned)(s[4] >> 4) & 3, (s[5] - rt_min) / 100.0, (s[6] - rt_min) / 100.0); }
                printf("\n");
            }
            if (stamps_prev) {
                std::vector<unsigned long long> sp((size_t)nwg * 8);
                chk(stamps_prev(sp.data(), sp.size() * sizeof(unsigned long long)), "stamps_prev");
                unsigned long long prev_end = 0, prev_begin = ~0ull, cur_begin = ~0ull;
                for (int w = 0; w < nwg; ++w) { prev_end = std::max(prev_end, sp[(size_t)w * 8 + 6]); prev_begin = std::min(prev_begin, sp[(size_t)w * 8 + 5]);
                                                cur_begin = std::min(cur_begin, st[(size_t)w * 8 + 5]); }
                printf("  previous launch: first entry -> last exit %.1f us; last exit of it -> first entry of this launch %.1f us; period %.1f us\n",
                       (prev_end - prev_begin) / 100.0, ((double)cur_begin - (double)prev_end) / 100.0, (cur_begin - prev_begin) / 100.0);
            }
            if (dump) {
                FILE* f = fopen(dump, "w");
                if (f) { fprintf(f, "[");
                    for (int w = 0; w < nwg; ++w) { const unsigned long long* s = &st[(size_t)w * 8];
                        fprintf(f, "%s[%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu]", w ? "," : "", s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]); }
                    fprintf(f, "]\n"); fclose(f); }
            }
        }
        CK(hipFree(x)); CK(hipFree(kern)); CK(hipFree(wt)); CK(hipFree(wf)); CK(hipFree(bias)); CK(hipFree(dzp));
        CK(hipFree(z)); CK(hipFree(dx)); CK(hipFree(dk)); CK(hipFree(part)); CK(hipFree(ws));
    }
    if (sum_us > 0) printf("sum %.1f us  %.1f TF\n", sum_us, sum_fl / sum_us / 1e6);
    return 0;
}
