// Diagnostic (GPU): variants of the planar image -> pixel records conversion (kernels_first.hip: image_records_kernel) on a 3 x 4096 x 4096 slide.
//   hipcc -O3 --offload-arch=gfx950 tests/diagnostics/records_bench.hip -o tests/diagnostics/records_bench
// Measured (MI355X, round 4): one pixel per thread 76 us (fp32 records, 6.2 TB/s), eight per thread 83 / 60 us (fp32 / bf16), four
// consecutive pixels with 16-byte loads 106 / 62 us; a plain 201 MB copy 66 us.  (rocprofv3 shows ~350-400 us for this kernel at the head
// of a slide - the first kernel after the previous slide's tail; the engine's own events put it at ~0.1 ms.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned bf16_bits(float f) { unsigned u = __builtin_bit_cast(unsigned, f); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; }
__device__ inline uint2 to_bf16(f32x4 v) { return make_uint2(bf16_bits(v.x) | (bf16_bits(v.y) << 16), bf16_bits(v.z) | (bf16_bits(v.w) << 16)); }

template <bool AS16, int PER>
__global__ __launch_bounds__(256) void strided(const float* image, int64_t sc, int64_t sh, int h, int w, void* rec) {
    const int x0 = blockIdx.x * (256 * PER) + threadIdx.x, y = blockIdx.y;
    const float* row = image + (int64_t)y * sh;
    const int64_t i0 = (int64_t)y * w;
    f32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int x = x0 + 256 * k; const float* s = row + (x < w ? x : 0); v[k] = f32x4{s[0], s[sc], s[2 * sc], 0.f}; }
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int x = x0 + 256 * k; if (x < w) { if constexpr (AS16) ((uint2*)rec)[i0 + x] = to_bf16(v[k]); else ((f32x4*)rec)[i0 + x] = v[k]; } }
}
// a thread takes 4 consecutive pixels: three 16-byte loads, 4 records
template <bool AS16>
__global__ __launch_bounds__(256) void quad(const float* image, int64_t sc, int64_t sh, int h, int w, void* rec) {
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= w) return;
    const float* row = image + (int64_t)y * sh + x;
    const f32x4 r = *(const f32x4*)row, g = *(const f32x4*)(row + sc), b = *(const f32x4*)(row + 2 * sc);
    const int64_t i0 = (int64_t)y * w + x;
    if constexpr (AS16) {
        uint4* dst = (uint4*)((uint2*)rec + i0);
        const uint2 a0 = to_bf16(f32x4{r.x, g.x, b.x, 0.f}), a1 = to_bf16(f32x4{r.y, g.y, b.y, 0.f}), a2 = to_bf16(f32x4{r.z, g.z, b.z, 0.f}), a3 = to_bf16(f32x4{r.w, g.w, b.w, 0.f});
        dst[0] = make_uint4(a0.x, a0.y, a1.x, a1.y);
        dst[1] = make_uint4(a2.x, a2.y, a3.x, a3.y);
    } else {
        f32x4* dst = (f32x4*)rec + i0;
        dst[0] = f32x4{r.x, g.x, b.x, 0.f}; dst[1] = f32x4{r.y, g.y, b.y, 0.f}; dst[2] = f32x4{r.z, g.z, b.z, 0.f}; dst[3] = f32x4{r.w, g.w, b.w, 0.f};
    }
}
__global__ void copyk(const f32x4* a, f32x4* b, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) b[i] = a[i]; }

int main() {
    const int h = 4096, w = 4096;
    float* img; void* rec;
    hipMalloc(&img, (size_t)3 * h * w * 4); hipMalloc(&rec, (size_t)h * w * 16);
    hipMemset(img, 0, (size_t)3 * h * w * 4); hipMemset(rec, 0, (size_t)h * w * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto fn, double mb) {
        fn(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.1f us  %6.2f TB/s\n", name, ms * 100, mb / (ms / 10 * 1e-3) / 1e6);
    };
    const int64_t sc = (int64_t)h * w;
    time("strided<fp32,1>", [&] { hipLaunchKernelGGL((strided<false, 1>), dim3(w / 256, h), dim3(256), 0, 0, img, sc, w, h, w, rec); }, 201 + 268);
    time("strided<fp32,8>", [&] { hipLaunchKernelGGL((strided<false, 8>), dim3(w / 2048, h), dim3(256), 0, 0, img, sc, w, h, w, rec); }, 201 + 268);
    time("strided<bf16,8>", [&] { hipLaunchKernelGGL((strided<true, 8>), dim3(w / 2048, h), dim3(256), 0, 0, img, sc, w, h, w, rec); }, 201 + 134);
    time("quad<fp32>", [&] { hipLaunchKernelGGL((quad<false>), dim3(w / 1024, h), dim3(256), 0, 0, img, sc, w, h, w, rec); }, 201 + 268);
    time("quad<bf16>", [&] { hipLaunchKernelGGL((quad<true>), dim3(w / 1024, h), dim3(256), 0, 0, img, sc, w, h, w, rec); }, 201 + 134);
    time("copy 201 MB", [&] { hipLaunchKernelGGL(copyk, dim3((unsigned)((size_t)3 * h * w / 4 / 256)), dim3(256), 0, 0, (const f32x4*)img, (f32x4*)rec, (size_t)3 * h * w / 4); }, 402);
    return 0;
}
