// Diagnostic (GPU): per-layer timing of the Winograd 3x3 stride-1 kernels against the direct fp32 MFMA kernels, on the layer
// shapes of the 32-base U-Net at 256 windows of 256^2.  Links the in-tree library and calls its internal launchers.
//
//   hipcc -O2 --offload-arch=gfx950 -I neurips-cellseg_amd/csrc tests/diagnostics/wino_layer_bench.hip \
//         -L neurips-cellseg_amd/cellseg_amd -lcellseg_hip -Wl,-rpath,'$ORIGIN/../../neurips-cellseg_amd/cellseg_amd' \
//         -o tests/diagnostics/wino_layer_bench
//   CS_WINO=2 tests/diagnostics/wino_layer_bench [n_windows]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "kernels.h"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                         \
            return 1;                                                              \
        }                                                                          \
    } while (0)

struct Layer {
    int hw, cin, cout;
};

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    const Layer layers[] = {{128, 32, 32}, {64, 64, 64}, {32, 128, 128}, {16, 256, 256}, {16, 256, 512}, {16, 512, 512}};
    hipStream_t s;
    CK(hipStreamCreate(&s));
    cs::PackCtx* pc = cs::pack_ctx_create(64);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Layer& L : layers) {
        const size_t npix = (size_t)n * L.hw * L.hw;
        std::vector<float> hx(npix * L.cin), hw((size_t)L.cout * L.cin * 9), hb(L.cout);
        unsigned rng = 12345u;
        auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 65536.f - 0.5f; };
        for (auto& v : hx) v = rnd() * 2.f;
        for (auto& v : hw) v = rnd() * 0.2f;
        for (auto& v : hb) v = rnd();
        float *x, *w, *b, *ww, *wd, *y1, *y2;
        double *st1, *st2;
        CK(hipMalloc(&x, hx.size() * 4));
        CK(hipMalloc(&w, hw.size() * 4));
        CK(hipMalloc(&b, hb.size() * 4));
        CK(hipMalloc(&ww, cs::packed_wino_floats(L.cout, L.cin) * 4));
        CK(hipMalloc(&wd, cs::packed_conv_floats(L.cout, L.cin, 3) * 4));
        CK(hipMalloc(&y1, npix * L.cout * 4));
        CK(hipMalloc(&y2, npix * L.cout * 4));
        CK(hipMalloc(&st1, (size_t)n * L.cout * 16));
        CK(hipMalloc(&st2, (size_t)n * L.cout * 16));
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(st1, 0, (size_t)n * L.cout * 16));
        CK(hipMemset(st2, 0, (size_t)n * L.cout * 16));
        if (cs::pack_wino_weights(pc, s, w, false, L.cout, L.cin, ww) || cs::pack_conv_weights(pc, s, w, false, L.cout, L.cin, 3, wd, 0, L.cout)) {
            printf("pack failed: %s\n", cs_last_error());
            return 1;
        }
        cs::ConvSrc a;
        a.ptr = x; a.c = L.cin; a.pitch = L.cin;
        float2* tab = nullptr;
        float* slope = nullptr;
        if (!getenv("WINO_RAW")) {          // lazy InstanceNorm + PReLU of the producer, as in the network (WINO_RAW=1: raw input)
            std::vector<float2> ht((size_t)n * L.cin);
            for (auto& v : ht) v = make_float2(1.f + 0.25f * rnd(), 0.5f * rnd());
            const float hs = 0.25f;
            CK(hipMalloc(&tab, ht.size() * 8));
            CK(hipMalloc(&slope, 4));
            CK(hipMemcpy(tab, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(slope, &hs, 4, hipMemcpyHostToDevice));
            a.tab = tab; a.slope = slope; a.tab_c = L.cin;
        }
        double* const st1_arg = getenv("WINO_NOSTATS") ? nullptr : st1;    // diagnostics: the epilogue without the fp64 statistics
        auto run_w = [&]() { return cs::launch_conv_wino(s, a, ww, b, y1, L.cout, L.cout, st1_arg, n, L.hw, L.hw); };
        auto run_d = [&]() { return cs::launch_conv_mfma(s, 3, 1, a, cs::ConvSrc(), wd, b, y2, L.cout, L.cout, st2, n, L.hw, L.hw); };
        if (run_w() || run_d()) {
            printf("launch failed: %s\n", cs_last_error());
            return 1;
        }
        CK(hipStreamSynchronize(s));
        // correctness: Winograd vs direct (both fp32), output and statistics of the single warm-up launch
        std::vector<float> o1(npix * L.cout), o2(npix * L.cout);
        std::vector<double> s1((size_t)n * L.cout * 2), s2((size_t)n * L.cout * 2);
        CK(hipMemcpy(o1.data(), y1, o1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(o2.data(), y2, o2.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s1.data(), st1, s1.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s2.data(), st2, s2.size() * 8, hipMemcpyDeviceToHost));
        double err = 0, amax = 0, serr = 0;
        for (size_t i = 0; i < o1.size(); ++i) {
            const double d = fabs((double)o1[i] - o2[i]);
            if (!(d <= err)) err = d;
            if (fabs(o2[i]) > amax) amax = fabs(o2[i]);
        }
        for (size_t i = 0; i < s1.size(); ++i) {
            const double d = fabs(s1[i] - s2[i]) / (fabs(s2[i]) + 1.0);
            if (!(d <= serr)) serr = d;
        }
        const int reps = 20;
        float ms_w = 0, ms_d = 0;
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) run_w();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_w, e0, e1));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) run_d();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_d, e0, e1));
        const double direct_gf = 2.0 * npix * L.cout * L.cin * 9 / 1e9, wino_gf = direct_gf * 16 / 36;
        const double us_w = ms_w / reps * 1e3, us_d = ms_d / reps * 1e3;
        printf("%3d^2 %3d->%3d  wino %7.1f us (%5.1f TF issued = %4.1f%% of 157.3; %5.1f TF direct-equivalent)   direct %7.1f us (%5.1f TF)   "
               "max|dy| %.2e of %.1f, stats rel %.1e\n",
               L.hw, L.cin, L.cout, us_w, wino_gf / us_w * 1e3, wino_gf / us_w * 1e3 / 157.3 * 100, direct_gf / us_w * 1e3, us_d,
               direct_gf / us_d * 1e3, err, amax, serr);
        if (cs::wino4_fits(L.cin, L.cout, L.hw, L.hw)) {        // the same layer as Winograd F(4x4,3x3)
            float* w4;
            CK(hipMalloc(&w4, cs::packed_wino4_floats(L.cout, L.cin) * 4));
            if (cs::pack_wino4_weights(pc, s, w, L.cout, L.cin, w4)) { printf("pack4 failed: %s\n", cs_last_error()); return 1; }
            CK(hipMemset(st1, 0, (size_t)n * L.cout * 16));
            CK(hipMemset(y1, 0, npix * L.cout * 4));
            auto run_4 = [&]() { return cs::launch_conv_wino4(s, a, w4, b, y1, L.cout, L.cout, st1_arg, n, L.hw, L.hw); };
            if (run_4()) { printf("launch4 failed: %s\n", cs_last_error()); return 1; }
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(o1.data(), y1, o1.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(s1.data(), st1, s1.size() * 8, hipMemcpyDeviceToHost));
            double err4 = 0, serr4 = 0;
            size_t worst = 0;
            for (size_t i = 0; i < o1.size(); ++i) {
                const double d = fabs((double)o1[i] - o2[i]);
                if (!(d <= err4)) { err4 = d; worst = i; }
            }
            for (size_t i = 0; i < s1.size(); ++i) {
                const double d = fabs(s1[i] - s2[i]) / (fabs(s2[i]) + 1.0);
                if (!(d <= serr4)) serr4 = d;
            }
            float ms_4 = 0;
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) run_4();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_4, e0, e1));
            const double us_4 = ms_4 / reps * 1e3, f4_gf = direct_gf * 36 / 144;
            const size_t wpix = worst / L.cout;
            printf("      F(4x4): %7.1f us (%5.1f TF issued = %4.1f%% of 157.3; %5.1f TF direct-equivalent)   max|dy| %.2e (at img %zu y %zu x %zu ch %zu), stats rel %.1e\n",
                   us_4, f4_gf / us_4 * 1e3, f4_gf / us_4 * 1e3 / 157.3 * 100, direct_gf / us_4 * 1e3, err4, wpix / ((size_t)L.hw * L.hw),
                   (wpix / L.hw) % L.hw, wpix % L.hw, worst % L.cout, serr4);
            (void)hipFree(w4);
        }
        if (tab) { (void)hipFree(tab); (void)hipFree(slope); }
        for (void* ptr : {(void*)x, (void*)w, (void*)b, (void*)ww, (void*)wd, (void*)y1, (void*)y2, (void*)st1, (void*)st2}) (void)hipFree(ptr);
    }
    cs::pack_ctx_destroy(pc);
    return 0;
}
