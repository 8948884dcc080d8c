// Diagnostic (GPU) microbenchmark: issue cost of v_fma_f32 / v_pk_fma_f32 / v_pk_add_f32 with 1, 2 and 4 waves per SIMD, alone and beside fp32 MFMAs
// (v_mfma_f32_32x32x2_f32): the fp32 matrix instruction executes on the vector ALU, so VALU work of ANY wave on the SIMD adds to the matrix time (DESIGN 3.1a″).
//   hipcc -O2 --offload-arch=gfx950 tests/diagnostics/valu_mfma_rate.hip -o /tmp/valu_mfma_rate && /tmp/valu_mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters, float s) {
    float a[8];
    f2 b[8];
    f16v acc = {0};
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = f2{a[i], a[i] + 1.f}; }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(b[i]) : "v"(f2{s, s}));
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b[i]) : "v"(f2{s, s}));
        } else if (MODE == 3) {       // 8 MFMA 32x32x2 + 32 v_fma interleaved
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
            }
        } else if (MODE == 4) {       // 8 MFMA only
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, acc, 0, 0, 0);
            }
        } else if (MODE == 5) {       // 8 MFMA + 32 v_pk_fma
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(b[i]) : "v"(f2{s, s}));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + b[i][0] + b[i][1];
    for (int i = 0; i < 16; ++i) r += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[1 + (threadIdx.x >> 6) * 2] = t0; cyc[2 + (threadIdx.x >> 6) * 2] = t1; }
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8 * 64);
    const int iters = 2000;
    const char* names[6] = {"32 v_fma_f32", "32 v_pk_fma_f32", "32 v_pk_add_f32", "8 mfma32x32x2 + 32 v_fma", "8 mfma32x32x2", "8 mfma32x32x2 + 32 v_pk_fma"};
    for (int threads : {256, 512, 1024}) {
        for (int m = 0; m < 6; ++m) {
            for (int rep = 0; rep < 2; ++rep) {
                switch (m) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0001f); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0001f); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0001f); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0001f); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0001f); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0001f); break;
                }
                hipDeviceSynchronize();
            }
            long long h[64]; hipMemcpy(h, cyc, 8 * 64, hipMemcpyDeviceToHost);
            long long lo = h[1], hi = h[2];
            for (int w = 0; w < threads / 64; ++w) { if (h[1 + 2 * w] < lo) lo = h[1 + 2 * w]; if (h[2 + 2 * w] > hi) hi = h[2 + 2 * w]; }
            printf("%4d threads/block (%d waves/SIMD)  %-30s  %.1f cycles per loop body (first start -> last end), wave 0 alone %.1f\n", threads, threads / 256, names[m], (double)(hi - lo) / iters, (double)(h[2] - h[1]) / iters);
        }
    }
    return 0;
}
