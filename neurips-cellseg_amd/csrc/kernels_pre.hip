// gfx950 input normalisation (row N of SURVEY.md 8(a), "next" row 8(f) rank 2): the reference's per-channel
// percentile stretch, bit-exact, on the device.
//
// Replaces /root/reference/baseline/predict.py:14-21 (normalize_channel) and :94-108:
//   gray -> 3 channels, > 3 channels -> first 3; per channel with any non-zero pixel:
//     p1, p99 = np.percentile(non-zero values, [1, 99])        (linear interpolation, numpy's _lerp in float64)
//     if p99 - p1 > 0.001:  u8 = uint8( (clip(x, p1, p99) - p1) / (p99 - p1) * 255.0 + 0.0 )   (float64, truncation)
//     else:                 u8 = x.astype(uint8)               (wraps for uint16 - reference quirk, kept)
//   out = float32( float64(u8) / float64(max over the whole uint8 image) )
// Integer images (uint8 / uint16) only: percentiles come from an exact 65536-bin histogram, so no sort is needed and
// the result is bit-identical to numpy's.  HBM-bound byte work: one histogram pass, one rescale pass, one scale pass.
#include "kernels.h"

namespace cs {

namespace {

struct PreParams {
    const void* img;      // [h][w][c_in] uint8 or uint16, contiguous
    int is_u16, h, w, c_in;
    unsigned int* hist;   // [3][65536]
    double* pct;          // [3][4]: p_lo, p_hi, mode (0 zero channel, 1 stretch, 2 raw cast), unused
    unsigned char* u8;    // [h][w][3]
    unsigned int* vmax;   // scalar
    float* out;           // [h][w][3]
};

__device__ __forceinline__ unsigned int load_px(const PreParams& p, int64_t pix, int ch) {
    const int c = p.c_in == 1 ? 0 : ch;                    // gray images are replicated to 3 channels
    const int64_t i = pix * p.c_in + c;
    return p.is_u16 ? (unsigned int)reinterpret_cast<const unsigned short*>(p.img)[i]
                    : (unsigned int)reinterpret_cast<const unsigned char*>(p.img)[i];
}

__global__ __launch_bounds__(256) void pre_hist_kernel(PreParams p) {
    const int64_t n = (int64_t)p.h * p.w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            if (p.c_in == 1 && ch > 0) break;              // identical channels: histogram once
            const unsigned int v = load_px(p, i, ch);
            if (v) atomicAdd(&p.hist[ch * 65536 + v], 1u);
        }
    }
}

// one block per channel: exact order statistics from the histogram, then numpy's linear interpolation
__global__ __launch_bounds__(1024) void pre_percentile_kernel(PreParams p, double q_lo, double q_hi) {
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long total_s;
    __shared__ double vals[4];
    const int ch = blockIdx.x;
    const unsigned int* hist = p.hist + (p.c_in == 1 ? 0 : ch) * 65536;
    const int t = threadIdx.x;                             // thread t owns bins [64 t, 64 t + 64)
    unsigned long long mine = 0;
    for (int b = 0; b < 64; ++b) mine += hist[t * 64 + b];
    part[t] = mine;
    __syncthreads();
    if (t == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; ++i) {
            const unsigned long long c = part[i];
            part[i] = run;                                 // exclusive prefix
            run += c;
        }
        total_s = run;
    }
    __syncthreads();
    const unsigned long long n = total_s;
    double* out = p.pct + ch * 4;
    if (n == 0) {
        if (t == 0) { out[0] = 0.0; out[1] = 0.0; out[2] = 0.0; }
        return;
    }
    // numpy 'linear' method: virtual_index = (n - 1) * q, q in [0, 1]
    double vi[2], gamma[2];
    long long lo[2], hi[2];
    const double qq[2] = {q_lo / 100.0, q_hi / 100.0};
    for (int k = 0; k < 2; ++k) {
        vi[k] = (double)(n - 1) * qq[k];
        double fl = floor(vi[k]);
        lo[k] = (long long)fl;
        hi[k] = lo[k] + 1;
        gamma[k] = vi[k] - fl;
        if (lo[k] < 0) lo[k] = 0;
        if (hi[k] > (long long)n - 1) hi[k] = (long long)n - 1;
        if (lo[k] > (long long)n - 1) lo[k] = (long long)n - 1;
    }
    // value at rank r = smallest v with cumulative(v) > r; each thread checks whether the rank falls inside its bins
    const unsigned long long base = part[t];
    for (int k = 0; k < 4; ++k) {
        const unsigned long long r = (unsigned long long)((k & 1) ? hi[k >> 1] : lo[k >> 1]);
        if (r >= base && r < base + mine) {
            unsigned long long run = base;
            for (int b = 0; b < 64; ++b) {
                run += hist[t * 64 + b];
                if (run > r) {
                    vals[k] = (double)(t * 64 + b);
                    break;
                }
            }
        }
    }
    __syncthreads();
    if (t == 0) {
        double pr[2];
        for (int k = 0; k < 2; ++k) {
            const double a = vals[2 * k], b = vals[2 * k + 1], g = gamma[k];
            const double diff = b - a;
            double r = a + diff * g;                       // numpy _lerp
            if (g >= 0.5) r = b - diff * (1.0 - g);
            pr[k] = r;
        }
        out[0] = pr[0];
        out[1] = pr[1];
        out[2] = (pr[1] - pr[0] > 0.001) ? 1.0 : 2.0;
    }
}

__global__ __launch_bounds__(256) void pre_rescale_kernel(PreParams p) {
    const int64_t n = (int64_t)p.h * p.w;
    unsigned int local_max = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const double* pc = p.pct + ch * 4;
            const unsigned int v = load_px(p, i, ch);
            unsigned char u = 0;
            if (pc[2] == 1.0) {
                double x = (double)v;
                x = x < pc[0] ? pc[0] : (x > pc[1] ? pc[1] : x);          // np.clip
                x = (x - pc[0]) / (pc[1] - pc[0]);
                x = x * 255.0 + 0.0;
                u = (unsigned char)x;                                      // truncation, as numpy's cast
            } else if (pc[2] == 2.0) {
                u = (unsigned char)(v & 0xffu);                            // astype(uint8) wraps
            }
            p.u8[i * 3 + ch] = u;
            local_max = local_max > u ? local_max : u;
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned int o = __shfl_xor(local_max, off);
        local_max = local_max > o ? local_max : o;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(p.vmax, local_max);
}

__global__ __launch_bounds__(256) void pre_unit_kernel(PreParams p) {
    const int64_t n = (int64_t)p.h * p.w * 3;
    const double m = (double)*p.vmax;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p.out[i] = (float)((double)p.u8[i] / m);               // float64 division, then the float32 cast (predict.py:107-108)
}

}  // namespace

size_t normalize_workspace_bytes(int h, int w) {
    return (size_t)3 * 65536 * 4 + 256 + 3 * 4 * 8 + 256 + ((size_t)h * w * 3 + 255) / 256 * 256 + 256;
}

int normalize_image(hipStream_t s, const void* img, int is_u16, int h, int w, int c_in, float* out_hwc, unsigned char* u8_out,
                    void* ws, size_t ws_bytes) {
    CS_REQUIRE(c_in >= 1 && h > 0 && w > 0, "normalize: bad image shape %d x %d x %d", h, w, c_in);
    CS_REQUIRE(c_in == 1 || c_in >= 3, "normalize: 2-channel images are not defined by the reference");
    if (ws_bytes < normalize_workspace_bytes(h, w)) {
        set_error("normalize: workspace %zu < %zu bytes", ws_bytes, normalize_workspace_bytes(h, w));
        return CS_ERR_WORKSPACE;
    }
    PreParams p{};
    char* b = (char*)ws;
    p.hist = (unsigned int*)b;           b += (size_t)3 * 65536 * 4;
    p.vmax = (unsigned int*)b;           b += 256;
    p.pct = (double*)b;                  b += 256;
    p.u8 = u8_out ? u8_out : (unsigned char*)b;
    p.img = img; p.is_u16 = is_u16; p.h = h; p.w = w; p.c_in = c_in; p.out = out_hwc;
    const int64_t n = (int64_t)h * w;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    CS_HIP(hipMemsetAsync(p.hist, 0, (size_t)3 * 65536 * 4 + 256, s));
    {
        ProfScope prof(K_PRE, s, 0, (double)n * c_in * (is_u16 ? 2 : 1));
        hipLaunchKernelGGL(pre_hist_kernel, dim3(blocks), dim3(256), 0, s, p);
        hipLaunchKernelGGL(pre_percentile_kernel, dim3(3), dim3(1024), 0, s, p, 1.0, 99.0);
    }
    {
        ProfScope prof(K_PRE, s, 0, (double)n * (c_in * (is_u16 ? 2 : 1) + 3));
        hipLaunchKernelGGL(pre_rescale_kernel, dim3(blocks), dim3(256), 0, s, p);
    }
    {
        ProfScope prof(K_PRE, s, 0, (double)n * 3 * 5);
        hipLaunchKernelGGL(pre_unit_kernel, dim3(blocks), dim3(256), 0, s, p);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
