// Diagnostic (GPU): the transposed convolutions of the decoder, Winograd F(2,2) form (kernels_convt_wino.hip) against the direct fp32
// MFMA kernel (kernels_conv.hip: launch_convt_mfma), on the decoder shapes of the 32-base U-Net at 256 windows of 256^2 and on odd sizes.
//   hipcc -O2 --offload-arch=gfx950 -I neurips-cellseg_amd/csrc tests/diagnostics/convt_layer_bench.hip -L neurips-cellseg_amd/cellseg_amd \
//         -lcellseg_hip -Wl,-rpath,'$ORIGIN/../../neurips-cellseg_amd/cellseg_amd' -o tests/diagnostics/convt_layer_bench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "kernels.h"

#define CK(x)                                                  \
    do {                                                       \
        hipError_t e_ = (x);                                   \
        if (e_ != hipSuccess) {                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));     \
            return 1;                                          \
        }                                                      \
    } while (0)

struct Layer { int h, w, ca, cb, cout, n; bool sum2; };

int main(int argc, char** argv) {
    const int n0 = argc > 1 ? atoi(argv[1]) : 256;
    const Layer layers[] = {{16, 16, 256, 512, 128, n0, false}, {32, 32, 128, 128, 64, n0, false}, {64, 64, 64, 64, 32, n0, false},
                            {32, 32, 128, 128, 64, n0, true},   {64, 64, 64, 64, 32, n0, true},
                            {13, 21, 32, 48, 64, 5, false},     {7, 9, 16, 16, 32, 3, true}};
    hipStream_t s;
    CK(hipStreamCreate(&s));
    cs::PackCtx* pc = cs::pack_ctx_create(64);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Layer& L : layers) {
        const int n = L.n, cin = L.ca + L.cb;
        const size_t npix = (size_t)n * L.h * L.w, opix = npix * 4;
        unsigned rng = 777u;
        auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 65536.f - 0.5f; };
        std::vector<float> ha(npix * L.ca), hb(npix * L.cb), hb2(npix * L.cb), hw((size_t)cin * L.cout * 9), hbias(L.cout);
        for (auto& v : ha) v = rnd() * 2.f;
        for (auto& v : hb) v = rnd() * 2.f;
        for (auto& v : hb2) v = rnd() * 2.f;
        for (auto& v : hw) v = rnd() * 0.2f;
        for (auto& v : hbias) v = rnd();
        float *a, *b, *b2, *w, *bias, *wd, *wt, *y1, *y2;
        double *st1, *st2;
        CK(hipMalloc(&a, ha.size() * 4)); CK(hipMalloc(&b, hb.size() * 4)); CK(hipMalloc(&b2, hb2.size() * 4));
        CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&bias, hbias.size() * 4));
        CK(hipMalloc(&wd, cs::packed_conv_floats(L.cout, cin, 3) * 4));
        CK(hipMalloc(&wt, cs::packed_convt_wino_floats(L.cout, cin) * 4));
        CK(hipMalloc(&y1, opix * L.cout * 4)); CK(hipMalloc(&y2, opix * L.cout * 4));
        CK(hipMalloc(&st1, (size_t)n * L.cout * 16)); CK(hipMalloc(&st2, (size_t)n * L.cout * 16));
        CK(hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(b2, hb2.data(), hb2.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(bias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(st1, 0, (size_t)n * L.cout * 16)); CK(hipMemset(st2, 0, (size_t)n * L.cout * 16));
        CK(hipMemset(y1, 0, opix * L.cout * 4)); CK(hipMemset(y2, 0, opix * L.cout * 4));
        if (cs::pack_conv_weights(pc, s, w, true, L.cout, cin, 3, wd, 0, L.cout) || cs::pack_convt_wino_weights(pc, s, w, L.cout, cin, wt)) {
            printf("pack failed: %s\n", cs_last_error());
            return 1;
        }
        cs::ConvSrc sa, sb;
        sa.ptr = a; sa.c = L.ca; sa.pitch = L.ca;
        sb.ptr = b; sb.c = L.cb; sb.pitch = L.cb;
        float2 *tab = nullptr, *tab2 = nullptr;
        float* slope = nullptr;
        if (L.sum2) {                                  // the tensor from below as a SUM source: two tensors, two lazy norm tables
            std::vector<float2> ht((size_t)n * L.cb), ht2((size_t)n * L.cb);
            for (auto& v : ht) v = make_float2(1.f + 0.25f * rnd(), 0.5f * rnd());
            for (auto& v : ht2) v = make_float2(1.f + 0.25f * rnd(), 0.5f * rnd());
            const float hs = 0.25f;
            CK(hipMalloc(&tab, ht.size() * 8)); CK(hipMalloc(&tab2, ht2.size() * 8)); CK(hipMalloc(&slope, 4));
            CK(hipMemcpy(tab, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(tab2, ht2.data(), ht2.size() * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(slope, &hs, 4, hipMemcpyHostToDevice));
            sb.tab = tab; sb.slope = slope; sb.tab_c = L.cb;
            sb.ptr2 = b2; sb.tab2 = tab2; sb.slope2 = slope;
        }
        auto run_w = [&]() { return cs::launch_convt_wino(s, sa, sb, wt, bias, y1, L.cout, L.cout, st1, n, L.h, L.w); };
        auto run_d = [&]() { return cs::launch_convt_mfma(s, sa, sb, wd, bias, y2, L.cout, L.cout, st2, n, L.h, L.w); };
        if (run_w() || run_d()) { printf("launch failed: %s\n", cs_last_error()); return 1; }
        CK(hipStreamSynchronize(s));
        std::vector<float> o1(opix * L.cout), o2(opix * L.cout);
        std::vector<double> s1((size_t)n * L.cout * 2), s2((size_t)n * L.cout * 2);
        CK(hipMemcpy(o1.data(), y1, o1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(o2.data(), y2, o2.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s1.data(), st1, s1.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s2.data(), st2, s2.size() * 8, hipMemcpyDeviceToHost));
        double err = 0, amax = 0, serr = 0;
        size_t worst = 0;
        for (size_t i = 0; i < o1.size(); ++i) {
            const double d = fabs((double)o1[i] - o2[i]);
            if (!(d <= err)) { err = d; worst = i; }
            if (fabs(o2[i]) > amax) amax = fabs(o2[i]);
        }
        for (size_t i = 0; i < s1.size(); ++i) {
            const double d = fabs(s1[i] - s2[i]) / (fabs(s2[i]) + 1.0);
            if (!(d <= serr)) serr = d;
        }
        const int reps = n >= 64 ? 20 : 3;
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
        const double direct_gf = 2.0 * npix * L.cout * (double)cin * 9 / 1e9, wino_gf = direct_gf * 25 / 36;
        const double us_w = ms_w / reps * 1e3, us_d = ms_d / reps * 1e3;
        const size_t wp = worst / L.cout;
        printf("%3dx%-3d %3d+%3d->%3d n %3d %s  F(2,2) %7.1f us (%5.1f TF issued = %4.1f%% of 157.3)   direct %7.1f us (%5.1f TF = %4.1f%%)   x%.2f   "
               "max|dy| %.2e of %.1f (img %zu y %zu x %zu ch %zu), stats rel %.1e\n",
               L.h, L.w, L.ca, L.cb, L.cout, n, L.sum2 ? "sum2" : "    ", us_w, wino_gf / us_w * 1e3, wino_gf / us_w * 1e3 / 157.3 * 100, us_d,
               direct_gf / us_d * 1e3, direct_gf / us_d * 1e3 / 157.3 * 100, us_d / us_w, err, amax, wp / ((size_t)4 * L.h * L.w),
               (wp / (2 * L.w)) % (2 * L.h), wp % (2 * L.w), worst % L.cout, serr);
        for (void* ptr : {(void*)a, (void*)b, (void*)b2, (void*)w, (void*)bias, (void*)wd, (void*)wt, (void*)y1, (void*)y2, (void*)st1, (void*)st2,
                          (void*)tab, (void*)tab2, (void*)slope})
            if (ptr) (void)hipFree(ptr);
    }
    cs::pack_ctx_destroy(pc);
    return 0;
}
