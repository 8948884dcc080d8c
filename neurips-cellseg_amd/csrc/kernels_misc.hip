// gfx950 kernels around the convolutions: normalisation tables, residual add, the two 3-channel layers at the
// top of the decoder (VALU: N = 3 is too narrow for MFMA tiles) and the sliding-window finalize.
//
// Reference rows: M (instance_norm / prelu / residual add inside monai UNet),
// T step 6 (out / count map, crop), S (softmax(dim=1)[0,1], predict.py:110-111), P1 (> 0.5, predict.py:113).
#include <type_traits>

#include "kernels.h"

namespace cs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float prelu1(float v, float slope) { return v >= 0.f ? v : v * slope; }

// ------------------------------------------------------------------------------------------------
// InstanceNorm statistics -> (scale, shift) table.  biased variance, eps inside the sqrt (torch semantics).
// ------------------------------------------------------------------------------------------------
// snap (inference): a channel that is CONSTANT over the image - every conv output behind an all-zero window is its bias - normalises to
// exactly 0 in exact arithmetic ((x - mean) * rstd with x == mean).  The consumers apply fma(x, rstd, -mean * rstd) with an fp32 table,
// which leaves the rounding residual of mean * rstd (up to 6e-8 x 316 x |mean|) instead of 0; two layers on, that residual is the only
// variance of the map and the net's output on blank windows is amplified rounding noise (0.03 ... 0.3 logits off an fp64 evaluation,
// tests/diagnostics/gpu_blank_windows.py).  The sums are exact fp64 sums of fp32 values, so a constant channel shows var <= 4e-16 mean^2:
// below 1e-13 mean^2 (a spread of 3e-7 of the value: a few ulp of jitter at most) the table entry is (0, 0).  Training keeps (rstd, -mean rstd):
// the backward pass needs rstd = 1 / sqrt(eps) there.
__global__ void norm_finalize_kernel(const double* stats, float2* tab, int total, double inv_count, double eps, int snap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double mean = stats[2 * i] * inv_count;
    double var = stats[2 * i + 1] * inv_count - mean * mean;
    if (snap && var <= 1e-13 * mean * mean) {
        tab[i] = make_float2(0.f, 0.f);
        return;
    }
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + eps);
    tab[i] = make_float2((float)rstd, (float)(-mean * rstd));
}

int launch_norm_finalize(hipStream_t s, const double* stats, float2* tab, int n, int c, int count, float eps, bool snap_constant) {
    const int total = n * c;
    ProfScope prof(K_NORM_FINALIZE, s, 0, 24.0 * total);
    hipLaunchKernelGGL(norm_finalize_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, stats, tab, total,
                       1.0 / (double)count, (double)eps, snap_constant ? 1 : 0);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

__global__ void bn_table_kernel(const float* weight, const float* bias, const float* mean, const float* var, float2* tab,
                                int n, int c, int c_total, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * c) return;
    const int ch = i % c, img = i / c;
    const float rstd = 1.0f / sqrtf(var[ch] + eps);
    const float alpha = rstd * (weight ? weight[ch] : 1.f);
    tab[(size_t)img * c_total + ch] = make_float2(alpha, (bias ? bias[ch] : 0.f) - mean[ch] * alpha);
}

int launch_bn_table(hipStream_t s, const float* weight, const float* bias, const float* mean, const float* var, float2* tab,
                    int n, int c, int c_total, float eps) {
    ProfScope prof(K_NORM_FINALIZE, s, 0, 8.0 * n * c);
    hipLaunchKernelGGL(bn_table_kernel, dim3((unsigned)ceil_div(n * c, 256)), dim3(256), 0, s, weight, bias, mean, var, tab, n,
                       c, c_total, eps);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// BatchNorm with BATCH statistics (training mode; torch.nn.BatchNorm2d.forward with self.training): the conv epilogues
// leave per-(image, channel) fp64 sums, one thread per channel reduces them over the batch - mean / biased variance over
// n*count elements, "wavefront reductions for BatchNorm statistics" of the north star happen in the producers - and writes
//   tab  [n][c_total]  (gamma*rstd, beta - mean*gamma*rstd)   what every consumer applies lazily (same for all images)
//   xtab [n][c_total]  (rstd, -mean*rstd)                     x_hat = y*rstd - mean*rstd, needed by the backward pass
// and updates the running statistics in place: momentum 0.1, UNBIASED variance (torch semantics).
__global__ void bn_train_finalize_kernel(const double* stats, const float* weight, const float* bias, float* run_mean, float* run_var,
                                         float2* tab, float2* xtab, int n, int c, int c_total, double inv_m, double unbias, double eps,
                                         float momentum) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < n; ++i) {
        s += stats[((size_t)i * c_total + ch) * 2];
        q += stats[((size_t)i * c_total + ch) * 2 + 1];
    }
    const double mean = s * inv_m;
    double var = q * inv_m - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + eps);
    const double g = weight ? (double)weight[ch] : 1.0, b = bias ? (double)bias[ch] : 0.0;
    const float2 t = make_float2((float)(rstd * g), (float)(b - mean * rstd * g));
    const float2 x = make_float2((float)rstd, (float)(-mean * rstd));
    for (int i = 0; i < n; ++i) {
        tab[(size_t)i * c_total + ch] = t;
        if (xtab) xtab[(size_t)i * c_total + ch] = x;
    }
    if (run_mean) {
        run_mean[ch] = (1.f - momentum) * run_mean[ch] + momentum * (float)mean;
        run_var[ch] = (1.f - momentum) * run_var[ch] + momentum * (float)(var * unbias);
    }
}

int launch_bn_train_finalize(hipStream_t s, const double* stats, const float* weight, const float* bias, float* run_mean,
                             float* run_var, float2* tab, float2* xtab, int n, int c, int c_total, int count, float eps,
                             float momentum) {
    const double m = (double)n * (double)count;
    ProfScope prof(K_NORM_FINALIZE, s, 0, 16.0 * n * c + 16.0 * n * c);
    hipLaunchKernelGGL(bn_train_finalize_kernel, dim3((unsigned)ceil_div(c, 64)), dim3(64), 0, s, stats, weight, bias, run_mean, run_var,
                       tab, xtab, n, c, c_total, 1.0 / m, m > 1.0 ? m / (m - 1.0) : 1.0, (double)eps, momentum);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// out = T_y(y) + T_r(r): materialises a ResidualUnit output (cx + res).  float4 per thread.
// ------------------------------------------------------------------------------------------------
struct AddSrc {
    const float* ptr;
    const float2* tab;
    const float* slope;
    int pitch, tab_c, tab_off;
};

template <bool A16>
__device__ __forceinline__ f32x4 load_lazy4(const AddSrc& s, int img, int64_t pix, int c4) {
    f32x4 v;
    if constexpr (A16) v = bf16x4_to_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(s.ptr) + pix * s.pitch + 4 * c4));
    else v = *reinterpret_cast<const f32x4*>(s.ptr + pix * s.pitch + 4 * c4);
    if (s.tab) {
        const f32x4* t = reinterpret_cast<const f32x4*>(s.tab + (size_t)img * s.tab_c + s.tab_off + 4 * c4);
        const f32x4 t0 = t[0], t1 = t[1];
        const float slope = s.slope ? *s.slope : 0.f;
        v.x = prelu1(v.x * t0.x + t0.y, slope);
        v.y = prelu1(v.y * t0.z + t0.w, slope);
        v.z = prelu1(v.z * t1.x + t1.y, slope);
        v.w = prelu1(v.w * t1.z + t1.w, slope);
    }
    return v;
}

template <bool A16>
__global__ void norm_act_add_kernel(AddSrc y, AddSrc r, float* out, int pitch_out, int c4n, int64_t npix, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / c4n;
        const int c4 = (int)(i - pix * c4n);
        const int img = (int)(pix / npix);
        const f32x4 a = load_lazy4<A16>(y, img, pix, c4);
        const f32x4 b = load_lazy4<A16>(r, img, pix, c4);
        if constexpr (A16) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out) + pix * pitch_out + 4 * c4) = f32x4_to_bf16(a + b);
        else *reinterpret_cast<f32x4*>(out + pix * pitch_out + 4 * c4) = a + b;
    }
}

// bf16 datapath: 8 channels (16 bytes) per thread and tensor - 8-byte accesses left a third of the HBM bandwidth unused
typedef unsigned u32x4m __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_lazy8(const AddSrc& s, int img, int64_t pix, int c8, float (&v)[8]) {
    const u32x4m raw = *reinterpret_cast<const u32x4m*>(reinterpret_cast<const unsigned short*>(s.ptr) + pix * s.pitch + 8 * c8);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[2 * k] = __builtin_bit_cast(float, raw[k] << 16);
        v[2 * k + 1] = __builtin_bit_cast(float, raw[k] & 0xffff0000u);
    }
    if (s.tab) {
        const f32x4* t = reinterpret_cast<const f32x4*>(s.tab + (size_t)img * s.tab_c + s.tab_off + 8 * c8);
        const float slope = s.slope ? *s.slope : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 tk = t[k];
            v[2 * k] = prelu1(v[2 * k] * tk.x + tk.y, slope);
            v[2 * k + 1] = prelu1(v[2 * k + 1] * tk.z + tk.w, slope);
        }
    }
}
__global__ void norm_act_add16x8_kernel(AddSrc y, AddSrc r, float* out, int pitch_out, int c8n, int64_t npix, int64_t total,
                                        unsigned char* out8, int pitch8) {
    if (out8) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);     // saturating e4m3 conversion (MODE.FP16_OVFL)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / c8n;
        const int c8 = (int)(i - pix * c8n);
        const int img = (int)(pix / npix);
        float a[8], b[8];
        load_lazy8(y, img, pix, c8, a);
        load_lazy8(r, img, pix, c8, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += b[k];
        if (out) {
            u32x4m o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = f32_to_bf16_bits(a[2 * k]) | (f32_to_bf16_bits(a[2 * k + 1]) << 16);
            *reinterpret_cast<u32x4m*>(reinterpret_cast<unsigned short*>(out) + pix * pitch_out + 8 * c8) = o;
        }
        if (out8) *reinterpret_cast<uint2*>(out8 + pix * pitch8 + 8 * c8) = f32x8_to_fp8(a);
    }
}

// fp8 mode: one pass that applies the producer's norm table + activation and stores e4m3 (2 bytes in, 1 byte out per element)
__global__ void act_quant_kernel(AddSrc src, unsigned char* dst8, int pitch8, int c8n, int64_t npix, int64_t total) {
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / c8n;
        const int c8 = (int)(i - pix * c8n);
        const int img = (int)(pix / npix);
        float a[8];
        load_lazy8(src, img, pix, c8, a);
        *reinterpret_cast<uint2*>(dst8 + pix * pitch8 + 8 * c8) = f32x8_to_fp8(a);
    }
}

static AddSrc to_add(const ConvSrc& s) { return AddSrc{s.ptr, s.tab, s.slope, s.pitch, s.tab_c ? s.tab_c : s.c, s.tab_off}; }

int launch_act_quant(hipStream_t s, const ConvSrc& src, unsigned char* dst8, int pitch8, int64_t npix_per_image, int n) {
    CS_REQUIRE(src.h16 && src.c % 8 == 0 && src.pitch % 8 == 0 && pitch8 % 8 == 0 && src.tab_off % 8 == 0,
               "act_quant: a bf16 source with channels / pitches in multiples of 8 is required");
    const int c8n = src.c / 8;
    const int64_t total = (int64_t)n * npix_per_image * c8n, blocks = (total + 255) / 256;
    ProfScope prof(K_NORM_ACT_ADD, s, 0, 24.0 * total);
    hipLaunchKernelGGL(act_quant_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, to_add(src), dst8, pitch8, c8n,
                       npix_per_image, total);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int launch_norm_act_add(hipStream_t s, const ConvSrc& y, const ConvSrc& r, float* out, int pitch_out, int64_t npix_per_image,
                        int n, bool act16, unsigned char* out8, int pitch8) {
    CS_REQUIRE(y.c == r.c && y.c % 4 == 0, "norm_act_add: channel mismatch %d vs %d", y.c, r.c);
    CS_REQUIRE(out || out8, "norm_act_add: no output");
    CS_REQUIRE(y.h16 == act16 && r.h16 == act16, "norm_act_add: mixed activation element types");
    const int c4n = y.c / 4;
    const int64_t total = (int64_t)n * npix_per_image * c4n;
    const int64_t blocks = (total + 255) / 256;
    ProfScope prof(K_NORM_ACT_ADD, s, 0, (act16 ? (16.0 + (out ? 8.0 : 0.0) + (out8 ? 4.0 : 0.0)) : 48.0) * total);
    const bool wide = act16 && y.c % 8 == 0 && y.pitch % 8 == 0 && r.pitch % 8 == 0 && pitch_out % 8 == 0 && y.tab_off % 8 == 0 && r.tab_off % 8 == 0;
    CS_REQUIRE(!out8 || (wide && pitch8 % 8 == 0), "norm_act_add: the e4m3 output needs the bf16 datapath with channels in multiples of 8");
    if (wide) {
        const int64_t total8 = total / 2, blocks8 = (total8 + 255) / 256;
        hipLaunchKernelGGL(norm_act_add16x8_kernel, dim3((unsigned)(blocks8 < 65536 ? blocks8 : 65536)), dim3(256), 0, s, to_add(y),
                           to_add(r), out, pitch_out, c4n / 2, npix_per_image, total8, out8, pitch8);
    } else if (act16)
        hipLaunchKernelGGL(norm_act_add_kernel<true>, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, to_add(y),
                           to_add(r), out, pitch_out, c4n, npix_per_image, total);
    else
        hipLaunchKernelGGL(norm_act_add_kernel<false>, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, to_add(y),
                       to_add(r), out, pitch_out, c4n, npix_per_image, total);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// top transposed conv: cin -> COUT (3) k3 s2 p1 op1, VALU.  One thread per INPUT pixel computes the 2x2 output
// quad.  Weights W[cin][COUT][3][3] are re-laid in LDS as [cin][COUT*9 (+pad to 28)] and read as broadcasts.
// ------------------------------------------------------------------------------------------------
struct SmallSrc {
    const float* ptr;
    int c, pitch;
};

template <bool IN16>
__device__ __forceinline__ f32x4 load4_act(const float* base, size_t elem) {     // 4 channels of an fp32 or a bf16 activation tensor
    if constexpr (IN16) return bf16x4_to_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem));
    else return *reinterpret_cast<const f32x4*>(base + elem);
}

template <int COUT, bool IN16>
__global__ __launch_bounds__(256) void convt_small_kernel(SmallSrc a, SmallSrc b, const float* w, const float* bias, float* out,
                                                          double* stats, int hin, int win) {
    constexpr int WROW = ((COUT * 9 + 3) / 4) * 4;
    extern __shared__ __attribute__((aligned(16))) float w_s[];   // [cin][WROW]
    const int cin = a.c + b.c;
    for (int i = threadIdx.x; i < cin * WROW; i += blockDim.x) {
        const int ci = i / WROW, r = i - ci * WROW;
        w_s[i] = r < COUT * 9 ? w[(size_t)ci * COUT * 9 + r] : 0.f;
    }
    __syncthreads();
    const int img = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = pix < hin * win;
    const int iy = valid ? pix / win : 0, ix = valid ? pix - iy * win : 0;
    float acc[4][COUT];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[ph][co] = 0.f;

    const bool has_r = ix + 1 < win, has_d = iy + 1 < hin;
    for (int part = 0; part < 2; ++part) {
        const SmallSrc& s = part == 0 ? a : b;
        if (s.c == 0) continue;
        const int cbase = part == 0 ? 0 : a.c;
        const size_t e00 = (((size_t)img * hin + iy) * win + ix) * s.pitch;          // element offsets (fp32 or bf16 elements)
        const size_t e01 = e00 + s.pitch, e10 = e00 + (size_t)win * s.pitch, e11 = e10 + s.pitch;
        for (int c = 0; c < s.c; c += 4) {
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 x00 = load4_act<IN16>(s.ptr, e00 + c);
            const f32x4 x01 = has_r ? load4_act<IN16>(s.ptr, e01 + c) : z;
            const f32x4 x10 = has_d ? load4_act<IN16>(s.ptr, e10 + c) : z;
            const f32x4 x11 = (has_r && has_d) ? load4_act<IN16>(s.ptr, e11 + c) : z;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* wr = &w_s[(cbase + c + j) * WROW];
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float* wk = wr + co * 9;   // [ky][kx]
                    acc[0][co] += x00[j] * wk[4];                                                      // (even, even)
                    acc[1][co] += x01[j] * wk[3] + x00[j] * wk[5];                                     // (even, odd)
                    acc[2][co] += x10[j] * wk[1] + x00[j] * wk[7];                                     // (odd, even)
                    acc[3][co] += x11[j] * wk[0] + x10[j] * wk[2] + x01[j] * wk[6] + x00[j] * wk[8];   // (odd, odd)
                }
            }
        }
    }
    const int hout = 2 * hin, wout = 2 * win;
    double s[COUT], q[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        s[co] = 0.0;
        q[co] = 0.0;
    }
    if (valid) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            float* o = out + (((size_t)img * hout + 2 * iy + (ph >> 1)) * wout + 2 * ix + (ph & 1)) * COUT;
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float v = acc[ph][co] + (bias ? bias[co] : 0.f);
                o[co] = v;
                s[co] += (double)v;
                q[co] += (double)v * (double)v;
            }
        }
    }
    if (stats) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                s[co] += __shfl_xor(s[co], off);
                q[co] += __shfl_xor(q[co], off);
            }
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(stats + ((size_t)img * COUT + co) * 2, s[co]);
                atomicAdd(stats + ((size_t)img * COUT + co) * 2 + 1, q[co]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// top transposed conv as GEMM + gather (the MFMA form of the kernel above, used when cin % 16 == 0):
//   P[pixel][co*9 + tap] = sum_cin x[pixel][cin] * W[cin][co][tap]        (M = pixels, N = 27 -> 32, K = cin; fp32 MFMA)
//   out[2i+a][2j+b][co]  = bias + sum over the 1/2/2/4 (ky,kx) of that phase of P[i+di][j+dj][co*9 + ky*3 + kx]
// A 16-wide MFMA tile cannot hold N = 3 output channels, but it holds the 27 (channel, tap) products: the block computes
// P for its 8 x 16 input pixels plus a one-pixel halo (9 x 17 = 153 rows of the GEMM), keeps it in LDS and every output
// pixel sums its contributions from there.  4.5x fewer MFMAs than padding N = 3 to 16, and no VALU inner loop.
// ------------------------------------------------------------------------------------------------
typedef float f32x4m __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) Px3 {             // one 3-channel pixel: a single 12-byte store (global_store_dwordx3)
    float c0, c1, c2;
};

// SUM2 (fp32 inference): the `b` source is the decoder's residual sum D = act(norm(V)) + act(norm(U)) of the level below the top, read as its
// two raw tensors with their lazy InstanceNorm tables + PReLU slopes - the norm_act_add pass that materialised D (two reads and one write of
// a 32-channel 128 x 128 tensor per window) is not launched.
struct SumExtra {
    const float* ptr2;             // U (same pitch / channel count as b.ptr = V)
    const float2* tab;             // table of V: [n][tab_c] (scale, shift), + tab_off
    const float2* tab2;            // table of U
    const float* slope;            // PReLU slopes (device scalars; null = ReLU)
    const float* slope2;
    int tab_c, tab_off;
};

template <int CIN, bool IN16, bool SUM2 = false>
__global__ __launch_bounds__(256) void convt_top_gemm_kernel(SmallSrc a, SmallSrc b, const float* __restrict__ w, const float* bias,
                                                             float* out, double* stats, int hin, int win, int tiles_x, int tiles_y,
                                                             int n_tiles, SumExtra sx) {
    static_assert(!(SUM2 && IN16), "sum sources are implemented for fp32 activations");
    static_assert(!SUM2 || CIN == 64 || CIN == 128, "sum sources: the items of each tensor must divide evenly over the four waves");
    constexpr int TH = 8, TW = 16, IH = TH + 1, IW = TW + 1, NPIX = IH * IW, MPAD = 160;   // 10 M-tiles of 16 rows
    constexpr int PP = 33;                                                              // P pitch (27 used, 32 computed)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int cin = CIN, XP = CIN + 4;                   // x pitch in floats
    float* x_s = sm;                                         // [MPAD][XP]
    float* w_s = sm + MPAD * XP;                             // [cin/16][4 kk][32 n][4 j] = W[16 c16 + 4 kk + j][n]
    float* p_s = sm;                                         // [MPAD][PP] (reuses x_s after the GEMM)
    __shared__ double red[4][6];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kk = lane >> 4;

    // PERSISTENT block: a contiguous run of tiles, the weights staged once, the NEXT tile's input patch prefetched into
    // registers while the current tile is multiplied and gathered, and all tile-invariant index arithmetic hoisted out of
    // the tile loop: the layer is bound by instruction issue (VALU + MFMA share a SIMD), not by HBM or LDS.
    const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(n_tiles, t_begin + per);
    if (t_begin >= t_end) return;

    // weights [cin][27] -> fragment order: one ds_read_b128 yields the B values of 4 consecutive MFMA k-steps
    for (int i = tid; i < cin * 32; i += 256) {
        const int j = i & 3, nn = (i >> 2) & 31, kq = (i >> 7) & 3, c16 = i >> 9;
        const int k = c16 * 16 + 4 * kq + j;
        w_s[i] = nn < 27 ? w[(size_t)k * 27 + nn] : 0.f;
    }
    constexpr int c4n = CIN / 4, ITERS = MPAD * c4n / 256;   // unconditional, fully unrolled loads: all in flight at once
    typedef typename std::conditional<IN16, uint2, f32x4m>::type raw4;   // 4 channels as loaded (bf16 bits or fp32)
    raw4 rx[ITERS];
    // tile-invariant part of the staging addresses (element offsets; bf16 sources are addressed through a 16-bit pointer)
    const float* sptr[ITERS];
    int rel[ITERS], spitch[ITERS], liy[ITERS], lix[ITERS];
    // item -> (pixel, channel quad).  Plain launches: item = tid + 256 it, quads fastest.  SUM2 launches (a.c == b.c): a wave instruction
    // covers 8 pixels x the 8 quads of ONE tensor, and every wave takes the skip tensor's quads in its first ITERS / 2 items and the sum's
    // in the rest - which items pay for the second tensor's loads and the two lazy activations is then known at compile time, and the four
    // waves carry equal shares of that work (all lanes predicated: 350 instructions per wave and tile; a / b waves: 60 / 350; this: 205)
    constexpr int HALF = ITERS / 2;
    auto item_of = [&](int it, int& pix, int& c4) {
        if constexpr (SUM2) {
            const int sec = it >= HALF ? 1 : 0, k = it - sec * HALF;
            c4 = sec * (c4n / 2) + (lane % (c4n / 2));
            pix = lane / (c4n / 2) + (64 / (c4n / 2)) * (wave + 4 * k);
        } else {
            const int i = tid + 256 * it;
            pix = i / c4n;
            c4 = i - pix * c4n;
        }
    };
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        int pix, c4;
        item_of(it, pix, c4);
        const int iy = pix / IW, ix = pix - iy * IW;
        const int c = 4 * c4;
        const bool in_a = c < a.c;
        sptr[it] = IN16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(in_a ? a.ptr : b.ptr) + (in_a ? c : c - a.c))
                        : (in_a ? a.ptr + c : b.ptr + (c - a.c));
        spitch[it] = in_a ? a.pitch : b.pitch;
        rel[it] = (iy * win + ix) * spitch[it];
        liy[it] = pix < NPIX ? iy : (1 << 20);               // rows beyond the patch are never valid
        lix[it] = ix;
    }
    unsigned okmask = 0;                                     // padding is zeroed at LDS-store time: consuming a loaded value
                                                             // earlier would make hipcc wait for the prefetch before the GEMM
    // SUM2: a thread's channel quad is the same for all its items (256 % c4n == 0), so it is a `b` thread or an `a` thread throughout;
    // `b` threads also fetch the second tensor and, per tile, the two lazy-norm tables of the tile's image (4 channels each)
    const int cb = SUM2 ? 4 * (lane % (c4n / 2)) : 0;                     // channel of this lane's sum-source items inside b
    const ptrdiff_t delta2 = SUM2 ? sx.ptr2 - b.ptr : 0;                  // second tensor: same offsets, other base
    f32x4m rx2[SUM2 ? ITERS - HALF : 1];
    f32x4m tA0, tA1, tB0, tB1;                               // (scale, shift) x 4 channels of V's and of U's table
    float slA = 0.f, slB = 0.f;
    if constexpr (SUM2) {
        slA = sx.slope ? *sx.slope : 0.f;
        slB = sx.slope2 ? *sx.slope2 : 0.f;
    }
    auto prefetch = [&](int img, int iy0, int ix0) {
        okmask = 0;
        const int base_px = (img * hin + iy0) * win + ix0;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const bool ok = iy0 + liy[it] < hin && ix0 + lix[it] < win;
            const int off = ok ? base_px * spitch[it] + rel[it] : 0;
            if constexpr (IN16) rx[it] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(sptr[it]) + off);
            else rx[it] = *reinterpret_cast<const f32x4m*>(sptr[it] + off);
            if constexpr (SUM2)
                if (it >= HALF) rx2[it - HALF] = *reinterpret_cast<const f32x4m*>(sptr[it] + delta2 + off);
            okmask |= ok ? (1u << it) : 0u;
        }
        if constexpr (SUM2) {
            const f32x4m* ta = reinterpret_cast<const f32x4m*>(sx.tab + (size_t)img * sx.tab_c + sx.tab_off + cb);
            const f32x4m* tb = reinterpret_cast<const f32x4m*>(sx.tab2 + (size_t)img * sx.tab_c + sx.tab_off + cb);
            tA0 = ta[0];
            tA1 = ta[1];
            tB0 = tb[0];
            tB1 = tb[1];
        }
    };
    auto act4 = [](f32x4m v, f32x4m t0, f32x4m t1, float slope) {      // lazy InstanceNorm + PReLU on 4 channels (conv_common.h: Lazy4::apply_on)
        f32x4m r;
        const float x0 = v.x * t0.x + t0.y, x1 = v.y * t0.z + t0.w, x2 = v.z * t1.x + t1.y, x3 = v.w * t1.z + t1.w;
        r.x = __builtin_fmaf(slope, __builtin_fminf(x0, 0.f), __builtin_fmaxf(x0, 0.f));
        r.y = __builtin_fmaf(slope, __builtin_fminf(x1, 0.f), __builtin_fmaxf(x1, 0.f));
        r.z = __builtin_fmaf(slope, __builtin_fminf(x2, 0.f), __builtin_fmaxf(x2, 0.f));
        r.w = __builtin_fmaf(slope, __builtin_fminf(x3, 0.f), __builtin_fmaxf(x3, 0.f));
        return r;
    };
    int xo[5], wo[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int t = wave + 4 * i, m = t >> 1, nn = t & 1;
        xo[i] = (m * 16 + li) * XP + 4 * kk;
        wo[i] = (kk * 32 + nn * 16 + li) * 4;
    }
    const int hout = 2 * hin, wout = 2 * win;
    float bv[3];                                             // loaded once: a load inside the tile loop would be the youngest
#pragma unroll                                               // outstanding one and its wait would also drain the prefetch
    for (int co = 0; co < 3; ++co) bv[co] = bias ? bias[co] : 0.f;
    // gather geometry of this thread's two output pixels (tile-invariant)
    int g_oyl[2], g_oxl[2], g_p[2][4], g_n[2];
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int o = tid + 256 * rep;                       // 0 .. 511
        const int oyl = o >> 5, oxl = o & 31;
        const int i = oyl >> 1, pa = oyl & 1, j = oxl >> 1, pb = oxl & 1;
        g_oyl[rep] = oyl;
        g_oxl[rep] = oxl;
        int cnt = 0;
        // rows: a = 0 -> (ky 1, di 0); a = 1 -> (ky 0, di 1), (ky 2, di 0).  columns alike.
#pragma unroll
        for (int ry = 0; ry < 2; ++ry) {
#pragma unroll
            for (int rxx = 0; rxx < 2; ++rxx) {
                const bool use = !(pa == 0 && ry == 1) && !(pb == 0 && rxx == 1);
                const int ky = pa == 0 ? 1 : (ry == 0 ? 0 : 2), di = (pa == 1 && ry == 0) ? 1 : 0;
                const int kx = pb == 0 ? 1 : (rxx == 0 ? 0 : 2), dj = (pb == 1 && rxx == 0) ? 1 : 0;
                g_p[rep][ry * 2 + rxx] = use ? ((i + di) * IW + j + dj) * PP + ky * 3 + kx : -1;
                cnt += use;
            }
        }
        g_n[rep] = cnt;
    }

    int tx = t_begin % tiles_x, ty = (t_begin / tiles_x) % tiles_y, img = t_begin / (tiles_x * tiles_y);
    prefetch(img, ty * TH, tx * TW);
    double s[3] = {0, 0, 0}, q[3] = {0, 0, 0};               // statistics of the current image, flushed when the image changes
    auto flush_stats = [&](int im) {
#pragma unroll
        for (int co = 0; co < 3; ++co) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                s[co] += __shfl_xor(s[co], off);
                q[co] += __shfl_xor(q[co], off);
            }
            if (lane == 0) {
                red[wave][2 * co] = s[co];
                red[wave][2 * co + 1] = q[co];
            }
            s[co] = 0.0;
            q[co] = 0.0;
        }
        __syncthreads();
        if (tid < 6) atomicAdd(stats + (size_t)im * 6 + tid, red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
        __syncthreads();
    };
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int iy0 = ty * TH, ix0 = tx * TW, cur_img = img;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            int pix, c4;
            item_of(it, pix, c4);
            f32x4m xv;
            if constexpr (IN16) xv = bf16x4_to_f32(rx[it]);
            else xv = rx[it];
            if constexpr (SUM2)
                if (it >= HALF) xv = act4(xv, tA0, tA1, slA) + act4(rx2[it - HALF], tB0, tB1, slB);    // same operations, same order as norm_act_add
            *reinterpret_cast<f32x4m*>(&x_s[pix * XP + 4 * c4]) = ((okmask >> it) & 1u) ? xv : f32x4m{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        // advance to the next tile (the last trip re-reads its own tile from L2: the loads stay unconditional)
        const bool more = tile + 1 < t_end;
        if (more && ++tx == tiles_x) {
            tx = 0;
            if (++ty == tiles_y) {
                ty = 0;
                ++img;
            }
        }
        prefetch(img, ty * TH, tx * TW);

        // GEMM: 10 M-tiles x 2 N-tiles = 20 output tiles, 5 per wave; 16 channels (4 k-steps) per pair of b128 reads
        f32x4m acc[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c16 = 0; c16 < cin / 16; ++c16) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const f32x4m xa = *reinterpret_cast<const f32x4m*>(&x_s[xo[i] + c16 * 16]);
                const f32x4m wb = *reinterpret_cast<const f32x4m*>(&w_s[wo[i] + c16 * 512]);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, wb.x, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, wb.y, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, wb.z, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, wb.w, acc[i], 0, 0, 0);
            }
        }
        __syncthreads();                                     // everyone is done with x_s: P may overwrite it
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int t = wave + 4 * i, m = t >> 1, nn = t & 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) p_s[(m * 16 + kk * 4 + r) * PP + nn * 16 + li] = acc[i][r];
        }
        __syncthreads();

        // gather: 16 x 32 output pixels x 3 channels per tile, two pixels per thread
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int oy = 2 * iy0 + g_oyl[rep], ox = 2 * ix0 + g_oxl[rep];
            if (oy < hout && ox < wout) {
                float v[3];
#pragma unroll
                for (int co = 0; co < 3; ++co) v[co] = bv[co];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (g_p[rep][t] >= 0) {
                        const float* pp = &p_s[g_p[rep][t]];
#pragma unroll
                        for (int co = 0; co < 3; ++co) v[co] += pp[co * 9];
                    }
                }
                *reinterpret_cast<Px3*>(out + (((size_t)cur_img * hout + oy) * wout + ox) * 3) = Px3{v[0], v[1], v[2]};
#pragma unroll
                for (int co = 0; co < 3; ++co) {
                    s[co] += (double)v[co];
                    q[co] += (double)v[co] * (double)v[co];
                }
            }
        }
        __syncthreads();                                     // gather reads of p_s are done before the next patch lands in x_s
        if (stats && (img != cur_img || !more)) flush_stats(cur_img);
    }
}

// ------------------------------------------------------------------------------------------------
// the same layer for the bf16 / fp8 datapaths: the inputs ARE bf16, so the patch is staged as a plain copy and the 27-column
// GEMM runs on v_mfma_f32_16x16x32_bf16 (CIN / 32 instructions per tile instead of CIN / 4 fp32 ones); weights are converted to
// bf16 fragments once per (persistent) block.  Gather, bias, statistics and the fp32 3-channel output as above.
// ------------------------------------------------------------------------------------------------
// SUM2 (bf16 datapath, round 5): the `b` source is the residual sum D = act(norm(V)) + act(norm(U)) of the level below, read as its two raw
// bf16 tensors with their lazy tables: the staging computes the sum in fp32 and rounds it to bf16 once - the very value norm_act_add16x8
// would have written - so the pass that materialised D (two reads and one write of a 32-channel 128 x 128 tensor per window) is not
// launched and the results are bitwise those of the unfused sequence.  Items are dealt so that a wave instruction covers ONE tensor: every
// wave takes the skip tensor's 16-byte items first and the sum's after them (which items pay for the second load and the two activations
// is known at compile time; the four waves carry equal shares).
typedef __bf16 bf16x8m __attribute__((ext_vector_type(8)));
template <int CIN, bool SUM2 = false>
__global__ __launch_bounds__(256) void convt_top_gemm16_kernel(SmallSrc a, SmallSrc b, const float* __restrict__ w, const float* bias,
                                                               float* out, double* stats, int hin, int win, int tiles_x, int tiles_y,
                                                               int n_tiles, SumExtra sx) {
    constexpr int TH = 8, TW = 16, IH = TH + 1, IW = TW + 1, NPIX = IH * IW, MPAD = 160;
    constexpr int PP = 33;                                   // P pitch in floats
    constexpr int XP = CIN + 8;                              // x pitch in bf16 elements: 16-byte A-fragment reads are conflict-free
    constexpr int X_BYTES = MPAD * XP * 2, P_BYTES = MPAD * PP * 4;
    constexpr int XP_BYTES = X_BYTES > P_BYTES ? X_BYTES : P_BYTES;
    extern __shared__ __attribute__((aligned(16))) char sm16[];
    unsigned short* x_s = reinterpret_cast<unsigned short*>(sm16);             // [MPAD][XP] bf16
    float* p_s = reinterpret_cast<float*>(sm16);                               // [MPAD][PP] (reuses x_s after the GEMM)
    unsigned short* w_s = reinterpret_cast<unsigned short*>(sm16 + XP_BYTES);  // [CIN/32][2][64 lanes][8] bf16 fragments
    __shared__ double red[4][6];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kk = lane >> 4;
    const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(n_tiles, t_begin + per);
    if (t_begin >= t_end) return;

    // weights W[cin][27] fp32 -> bf16 fragments: element (c32, nn, lane, j) = W[c32*32 + 8 (lane >> 4) + j][nn*16 + (lane & 15)]
    for (int i = tid; i < CIN * 32; i += 256) {
        const int j = i & 7, ln = (i >> 3) & 63, nn = (i >> 9) & 1, c32 = i >> 10;
        const int k = c32 * 32 + 8 * (ln >> 4) + j, col = nn * 16 + (ln & 15);
        w_s[i] = (unsigned short)f32_to_bf16_bits(col < 27 ? w[(size_t)k * 27 + col] : 0.f);
    }
    static_assert(!SUM2 || CIN == 64 || CIN == 128, "sum sources: 64 / 128 input channels (half of them summed)");
    constexpr int c8n = CIN / 8;                             // 16-byte items per pixel
    // SUM2: G item columns per tensor, PXW pixels per wave instruction, KS items per wave and tensor (pixels >= MPAD are dropped)
    constexpr int G = c8n / 2, PXW = 64 / G, KS = (MPAD + 4 * PXW - 1) / (4 * PXW);
    constexpr int ITERS = SUM2 ? 2 * KS : (MPAD * c8n + 255) / 256;
    typedef unsigned u32x4t __attribute__((ext_vector_type(4)));
    u32x4t rx[ITERS];
    u32x4t rx2[SUM2 ? KS : 1];
    const unsigned short* sptr[ITERS];
    int rel[ITERS], spitch[ITERS], liy[ITERS], lix[ITERS];
    auto item_of = [&](int it, int& pix, int& c8, bool& live) {
        if constexpr (SUM2) {
            const int sec = it >= KS ? 1 : 0, k = it - sec * KS;
            c8 = sec * G + (lane % G);
            pix = lane / G + PXW * (wave + 4 * k);
            live = pix < MPAD;
        } else {
            const int i = tid + 256 * it;
            pix = i / c8n;
            c8 = i - pix * c8n;
            live = i < MPAD * c8n;
        }
    };
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        int pix, c8;
        bool live;
        item_of(it, pix, c8, live);
        const int iy = pix / IW, ix = pix - iy * IW;
        const int c = 8 * c8;
        const bool in_a = c < a.c;
        sptr[it] = reinterpret_cast<const unsigned short*>(in_a ? a.ptr : b.ptr) + (in_a ? c : c - a.c);
        spitch[it] = in_a ? a.pitch : b.pitch;
        rel[it] = (iy * win + ix) * spitch[it];
        liy[it] = (pix < NPIX && live) ? iy : (1 << 20);
        lix[it] = ix;
    }
    unsigned okmask = 0;
    const int cb = SUM2 ? 8 * (lane % G) : 0;                // channel of this lane's sum-source items inside b
    const ptrdiff_t delta2 = SUM2 ? reinterpret_cast<const unsigned short*>(sx.ptr2) - reinterpret_cast<const unsigned short*>(b.ptr) : 0;
    f32x4m tA[4], tB[4];                                     // (scale, shift) x 8 channels of V's and of U's table
    float slA = 0.f, slB = 0.f;
    if constexpr (SUM2) {
        slA = sx.slope ? *sx.slope : 0.f;
        slB = sx.slope2 ? *sx.slope2 : 0.f;
    }
    auto prefetch = [&](int img, int iy0, int ix0) {
        okmask = 0;
        const int base_px = (img * hin + iy0) * win + ix0;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const bool ok = iy0 + liy[it] < hin && ix0 + lix[it] < win;
            const int off = ok ? base_px * spitch[it] + rel[it] : 0;
            rx[it] = *reinterpret_cast<const u32x4t*>(sptr[it] + off);
            if constexpr (SUM2)
                if (it >= KS) rx2[it - KS] = *reinterpret_cast<const u32x4t*>(sptr[it] + delta2 + off);
            okmask |= ok ? (1u << it) : 0u;
        }
        if constexpr (SUM2) {
            const f32x4m* ta = reinterpret_cast<const f32x4m*>(sx.tab + (size_t)img * sx.tab_c + sx.tab_off + cb);
            const f32x4m* tb = reinterpret_cast<const f32x4m*>(sx.tab2 + (size_t)img * sx.tab_c + sx.tab_off + cb);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tA[k] = ta[k];
                tB[k] = tb[k];
            }
        }
    };
    // act(norm(V)) + act(norm(U)) on 8 channels, the operations and their order as in norm_act_add16x8_kernel (load_lazy8), one bf16 rounding
    auto sum8 = [&](const u32x4t& v, const u32x4t& u) {
        u32x4t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v0 = __builtin_bit_cast(float, v[k] << 16), v1 = __builtin_bit_cast(float, v[k] & 0xffff0000u);
            const float u0 = __builtin_bit_cast(float, u[k] << 16), u1 = __builtin_bit_cast(float, u[k] & 0xffff0000u);
            const float s0 = prelu1(v0 * tA[k].x + tA[k].y, slA) + prelu1(u0 * tB[k].x + tB[k].y, slB);
            const float s1 = prelu1(v1 * tA[k].z + tA[k].w, slA) + prelu1(u1 * tB[k].z + tB[k].w, slB);
            o[k] = f32_to_bf16_bits(s0) | (f32_to_bf16_bits(s1) << 16);
        }
        return o;
    };
    int xo[5], wo[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int t = wave + 4 * i, m = t >> 1, nn = t & 1;
        xo[i] = (m * 16 + li) * XP + 8 * kk;                 // bf16 elements
        wo[i] = (nn * 64 + lane) * 8;
    }
    const int hout = 2 * hin, wout = 2 * win;
    float bv[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) bv[co] = bias ? bias[co] : 0.f;
    int g_oyl[2], g_oxl[2], g_p[2][4];
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int o = tid + 256 * rep;
        const int oyl = o >> 5, oxl = o & 31;
        const int i = oyl >> 1, pa = oyl & 1, j = oxl >> 1, pb = oxl & 1;
        g_oyl[rep] = oyl;
        g_oxl[rep] = oxl;
#pragma unroll
        for (int ry = 0; ry < 2; ++ry) {
#pragma unroll
            for (int rxx = 0; rxx < 2; ++rxx) {
                const bool use = !(pa == 0 && ry == 1) && !(pb == 0 && rxx == 1);
                const int ky = pa == 0 ? 1 : (ry == 0 ? 0 : 2), di = (pa == 1 && ry == 0) ? 1 : 0;
                const int kx = pb == 0 ? 1 : (rxx == 0 ? 0 : 2), dj = (pb == 1 && rxx == 0) ? 1 : 0;
                g_p[rep][ry * 2 + rxx] = use ? ((i + di) * IW + j + dj) * PP + ky * 3 + kx : -1;
            }
        }
    }

    int tx = t_begin % tiles_x, ty = (t_begin / tiles_x) % tiles_y, img = t_begin / (tiles_x * tiles_y);
    prefetch(img, ty * TH, tx * TW);
    double s[3] = {0, 0, 0}, q[3] = {0, 0, 0};
    auto flush_stats = [&](int im) {
#pragma unroll
        for (int co = 0; co < 3; ++co) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                s[co] += __shfl_xor(s[co], off);
                q[co] += __shfl_xor(q[co], off);
            }
            if (lane == 0) {
                red[wave][2 * co] = s[co];
                red[wave][2 * co + 1] = q[co];
            }
            s[co] = 0.0;
            q[co] = 0.0;
        }
        __syncthreads();
        if (tid < 6) atomicAdd(stats + (size_t)im * 6 + tid, red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
        __syncthreads();
    };
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int iy0 = ty * TH, ix0 = tx * TW, cur_img = img;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            int pix, c8;
            bool live;
            item_of(it, pix, c8, live);
            u32x4t xv = rx[it];
            if constexpr (SUM2)
                if (it >= KS) xv = sum8(rx[it], rx2[it - KS]);
            if (live) *reinterpret_cast<u32x4t*>(&x_s[pix * XP + 8 * c8]) = ((okmask >> it) & 1u) ? xv : u32x4t{0u, 0u, 0u, 0u};
        }
        __syncthreads();
        const bool more = tile + 1 < t_end;
        if (more && ++tx == tiles_x) {
            tx = 0;
            if (++ty == tiles_y) {
                ty = 0;
                ++img;
            }
        }
        prefetch(img, ty * TH, tx * TW);

        f32x4m acc[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c32 = 0; c32 < CIN / 32; ++c32) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const bf16x8m xa = *reinterpret_cast<const bf16x8m*>(&x_s[xo[i] + c32 * 32]);
                const bf16x8m wb = *reinterpret_cast<const bf16x8m*>(&w_s[wo[i] + c32 * 1024]);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, wb, acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int t = wave + 4 * i, m = t >> 1, nn = t & 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) p_s[(m * 16 + kk * 4 + r) * PP + nn * 16 + li] = acc[i][r];
        }
        __syncthreads();
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int oy = 2 * iy0 + g_oyl[rep], ox = 2 * ix0 + g_oxl[rep];
            if (oy < hout && ox < wout) {
                float v[3];
#pragma unroll
                for (int co = 0; co < 3; ++co) v[co] = bv[co];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (g_p[rep][t] >= 0) {
                        const float* pp = &p_s[g_p[rep][t]];
#pragma unroll
                        for (int co = 0; co < 3; ++co) v[co] += pp[co * 9];
                    }
                }
                *reinterpret_cast<Px3*>(out + (((size_t)cur_img * hout + oy) * wout + ox) * 3) = Px3{v[0], v[1], v[2]};
#pragma unroll
                for (int co = 0; co < 3; ++co) {
                    s[co] += (double)v[co];
                    q[co] += (double)v[co] * (double)v[co];
                }
            }
        }
        __syncthreads();
        if (stats && (img != cur_img || !more)) flush_stats(cur_img);
    }
}

int launch_convt_small(hipStream_t s, const ConvSrc& a, const ConvSrc& b, const float* w_torch, const float* bias, float* out,
                       int cout, double* stats, int n, int hin, int win) {
    CS_REQUIRE(cout == 3, "convT(small): cout %d unsupported (3-class contract)", cout);
    const bool sum2 = b.ptr2 != nullptr;
    CS_REQUIRE(a.tab == nullptr && (b.tab == nullptr || sum2), "convT(small): lazy inputs only as a sum source (second operand)");
    CS_REQUIRE(!sum2 || (b.tab && b.tab2 && a.h16 == b.h16 && a.c == b.c && (a.c + b.c == 64 || a.c + b.c == 128)),
               "convT(small): a sum source needs both lazy tables, 64 / 128 input channels and as many skip channels as summed ones");
    CS_REQUIRE(a.c % 4 == 0 && b.c % 4 == 0, "convT(small): channels must be multiples of 4");
    CS_REQUIRE(a.h16 == b.h16 || b.c == 0, "convT(small): mixed activation element types");
    const int cin = a.c + b.c;
    const size_t lds = (size_t)cin * 28 * sizeof(float);
    const double flops = 2.0 * n * hin * win * (double)cout * cin * 9;
    const double bytes = n * ((a.h16 ? 2.0 : 4.0) * hin * win * cin + 4.0 * 4.0 * hin * win * cout);
    ProfScope prof(K_CONVT_TOP, s, flops, bytes);
    if ((cin == 32 || cin == 64 || cin == 128) && a.c % 4 == 0) {      // GEMM + gather on the matrix cores
        const int tiles_x = ceil_div(win, 16), tiles_y = ceil_div(hin, 8);
        const size_t sm = ((size_t)160 * (cin + 4) + (size_t)cin * 32) * sizeof(float);
        const int n_tiles = tiles_x * tiles_y * n;
        const int resident = 256 * (int)((160 * 1024) / (sm + 64));           // persistent blocks: as many as the LDS lets live at once
        const dim3 grid((unsigned)(n_tiles < resident ? n_tiles : resident));
        const SmallSrc sa{a.ptr, a.c, a.pitch}, sb{b.ptr, b.c, b.pitch};
        const SumExtra sx{b.ptr2, b.tab, b.tab2, b.slope, b.slope2, b.tab_c ? b.tab_c : b.c, b.tab_off};
#define CS_TOP(CIN_, H_) hipLaunchKernelGGL((convt_top_gemm_kernel<CIN_, H_>), grid, dim3(256), sm, s, sa, sb, w_torch, bias, out, stats, hin, win, tiles_x, tiles_y, n_tiles, sx)
#define CS_TOPS(CIN_) hipLaunchKernelGGL((convt_top_gemm_kernel<CIN_, false, true>), grid, dim3(256), sm, s, sa, sb, w_torch, bias, out, stats, hin, win, tiles_x, tiles_y, n_tiles, sx)
        if (sum2 && !a.h16) {
            if (cin == 64) CS_TOPS(64); else CS_TOPS(128);
            CS_HIP(hipGetLastError());
            return CS_OK;
        }
        CS_REQUIRE(!sum2 || (a.c % 16 == 0 && a.pitch % 8 == 0 && b.pitch % 8 == 0 && sx.tab_off % 8 == 0), "convT(small): bf16 sum source: channels / pitches in multiples of 8");
        if (a.h16 && a.c % 8 == 0 && b.c % 8 == 0) {           // bf16 inputs: bf16 matrix pipe, plain-copy staging
            const size_t xb = (size_t)160 * (cin + 8) * 2, pb = (size_t)160 * 33 * 4;
            const size_t sm16 = (xb > pb ? xb : pb) + (size_t)cin * 32 * 2;
            const int resident16 = 256 * (int)((160 * 1024) / (sm16 + 64));
            const dim3 grid16((unsigned)(n_tiles < resident16 ? n_tiles : resident16));
#define CS_TOP16(CIN_, S_) hipLaunchKernelGGL((convt_top_gemm16_kernel<CIN_, S_>), grid16, dim3(256), sm16, s, sa, sb, w_torch, bias, out, stats, hin, win, tiles_x, tiles_y, n_tiles, sx)
            if (sum2) { if (cin == 64) CS_TOP16(64, true); else CS_TOP16(128, true); }
            else if (cin == 32) CS_TOP16(32, false); else if (cin == 64) CS_TOP16(64, false); else CS_TOP16(128, false);
#undef CS_TOP16
        } else if (a.h16) {
            if (cin == 32) CS_TOP(32, true); else if (cin == 64) CS_TOP(64, true); else CS_TOP(128, true);
        } else {
            if (cin == 32) CS_TOP(32, false); else if (cin == 64) CS_TOP(64, false); else CS_TOP(128, false);
        }
#undef CS_TOP
#undef CS_TOPS
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
    CS_REQUIRE(!sum2, "convT(small): the VALU form takes plain tensors");
    dim3 grid((unsigned)ceil_div(hin * win, 256), (unsigned)n);
    if (a.h16)
        hipLaunchKernelGGL((convt_small_kernel<3, true>), grid, dim3(256), lds, s, SmallSrc{a.ptr, a.c, a.pitch},
                           SmallSrc{b.ptr, b.c, b.pitch}, w_torch, bias, out, stats, hin, win);
    else
        hipLaunchKernelGGL((convt_small_kernel<3, false>), grid, dim3(256), lds, s, SmallSrc{a.ptr, a.c, a.pitch},
                       SmallSrc{b.ptr, b.c, b.pitch}, w_torch, bias, out, stats, hin, win);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// last layer: logits = conv3x3(T(u)) + bias + T(u)   (ResidualUnit(3->3, subunits=1, last_conv_only) with an
// identity residual).  T = InstanceNorm + PReLU of the top transposed conv, applied lazily.  One thread per pixel.
// ------------------------------------------------------------------------------------------------
// Round 5: the patch is staged with 16-byte loads and read back with 16-byte LDS reads, the 81 weights sit in scalar registers and the
// 12 output floats of a thread leave as three 16-byte stores (the first form moved every element as a dword - 14 loads, 14 LDS stores and
// 135 LDS reads per thread - and ran at 44 % of the HBM rate).  Window widths are multiples of 16, so every row of u / out starts on a
// 16-byte boundary; the LDS row mirrors global floats [3 x0 - 4, 3 x0 + 200) of the image row, i.e. patch pixel x0 - 1 starts at float 1.
template <int C>
__global__ __launch_bounds__(256) void conv_last_kernel(const float* __restrict__ u, const float2* __restrict__ tab, const float* __restrict__ slope_p,
                                                        const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out,
                                                        int h, int wd, int tiles_x, int tiles_y) {
    static_assert(C == 3, "3-class contract");
    constexpr int TH = 16, TW = 64, PH = TH + 2;
    constexpr int ROW4 = (TW * C + 8 + 3) / 4 + 1;           // 51 float4 per patch row: floats [-4, 200) around the tile's first pixel
    constexpr int PITCH4 = ROW4 + 2;                         // 53 float4: rows start 16 banks apart (the 16 threads of a row read 48-byte strides)
    __shared__ f32x4m p_s[PH * PITCH4];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int img = bid / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    float sc[C], sh[C];
    const float slope = (tab && slope_p) ? *slope_p : 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float2 t = tab ? tab[img * C + c] : make_float2(1.f, 0.f);
        sc[c] = t.x;
        sh[c] = t.y;
    }
    float wr[C * C * 9], br[C];                              // uniform addresses: scalar loads, the FMAs below take them as SGPR operands
#pragma unroll
    for (int i = 0; i < C * C * 9; ++i) wr[i] = w[i];
#pragma unroll
    for (int c = 0; c < C; ++c) br[c] = bias ? bias[c] : 0.f;
    const float* base = u + (size_t)img * h * wd * C;
    const int row_floats = wd * C;
    constexpr int N4 = PH * ROW4, ITERS = (N4 + 255) / 256;
    f32x4m rv[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {                     // unconditional, clamped 16-byte loads: all in flight together
        const int i = tid + 256 * it;
        const int r = i / ROW4, q4 = i - r * ROW4;
        const int yy = y0 - 1 + r, f0 = x0 * C - 4 + 4 * q4;  // first float of the quad inside the image row
        const bool ok = i < N4 && (unsigned)yy < (unsigned)h && f0 >= 0 && f0 + 3 < row_floats;
        rv[it] = *reinterpret_cast<const f32x4m*>(base + (ok ? (size_t)yy * row_floats + f0 : 0));
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = tid + 256 * it;
        const int r = i / ROW4, q4 = i - r * ROW4;
        const int yy = y0 - 1 + r, f0 = x0 * C - 4 + 4 * q4;
        const bool ok = (unsigned)yy < (unsigned)h && f0 >= 0 && f0 + 3 < row_floats;      // quads never straddle the image edge (wd % 4 == 0)
        f32x4m v = rv[it];
        if (tab) {
            // channel of float f0 + k: (f0 + k) % 3 with f0 = 3 x0 - 4 + 4 q4, x0 % 4 == 0  ->  (2 + q4 + k) % 3
            const int c0 = (2 + q4) % 3;
            const float s0 = c0 == 0 ? sc[0] : c0 == 1 ? sc[1] : sc[2], s1 = c0 == 0 ? sc[1] : c0 == 1 ? sc[2] : sc[0], s2 = c0 == 0 ? sc[2] : c0 == 1 ? sc[0] : sc[1];
            const float h0 = c0 == 0 ? sh[0] : c0 == 1 ? sh[1] : sh[2], h1 = c0 == 0 ? sh[1] : c0 == 1 ? sh[2] : sh[0], h2 = c0 == 0 ? sh[2] : c0 == 1 ? sh[0] : sh[1];
            v.x = prelu1(v.x * s0 + h0, slope);
            v.y = prelu1(v.y * s1 + h1, slope);
            v.z = prelu1(v.z * s2 + h2, slope);
            v.w = prelu1(v.w * s0 + h0, slope);
        }
        if (i < N4) p_s[r * PITCH4 + q4] = ok ? v : f32x4m{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int ly = tid >> 4, lx = (tid & 15) * 4;            // 16 rows x 16 groups of 4 pixels
    // the thread's 6 x 3 floats of a row start at float 1 + 3 lx = 1 + 12 m of the LDS row: five 16-byte reads [12 m, 12 m + 20)
    float nb[3][20];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const f32x4m q = p_s[(ly + r) * PITCH4 + (lx * C) / 4 + k];
            nb[r][4 * k] = q.x; nb[r][4 * k + 1] = q.y; nb[r][4 * k + 2] = q.z; nb[r][4 * k + 3] = q.w;
        }
    const int y = y0 + ly;
    if (y >= h || x0 + lx >= wd) return;                     // (wd % 4 == 0: a thread's four pixels are inside or outside together)
    float o[4 * C];
#pragma unroll
    for (int px = 0; px < 4; ++px) {
        float acc[C];
#pragma unroll
        for (int co = 0; co < C; ++co) acc[co] = br[co];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ci = 0; ci < C; ++ci) {
                    const float v = nb[ky][1 + (px + kx) * C + ci];
#pragma unroll
                    for (int co = 0; co < C; ++co) acc[co] += v * wr[(co * C + ci) * 9 + ky * 3 + kx];
                }
        // identity residual T(u), added after the conv as the reference does
#pragma unroll
        for (int co = 0; co < C; ++co) o[px * C + co] = acc[co] + nb[1][1 + (px + 1) * C + co];
    }
    f32x4m* dst = reinterpret_cast<f32x4m*>(out + (((size_t)img * h + y) * wd + x0 + lx) * C);
    dst[0] = f32x4m{o[0], o[1], o[2], o[3]};
    dst[1] = f32x4m{o[4], o[5], o[6], o[7]};
    dst[2] = f32x4m{o[8], o[9], o[10], o[11]};
}

int launch_conv_last(hipStream_t s, const ConvSrc& u, const float* w_torch, const float* bias, float* out, int c, int n, int h,
                     int w) {
    CS_REQUIRE(c == 3 && u.pitch == 3, "conv(last): %d channels unsupported (3-class contract)", c);
    CS_REQUIRE(w % 4 == 0, "conv(last): the width %d must be a multiple of 4 (windows are multiples of 16)", w);
    const double flops = 2.0 * n * h * w * 81.0;
    const double bytes = 4.0 * n * (double)h * w * 6;
    ProfScope prof(K_CONV_LAST, s, flops, bytes);
    const int tiles_x = ceil_div(w, 64), tiles_y = ceil_div(h, 16);
    dim3 grid((unsigned)(tiles_x * tiles_y * n));
    hipLaunchKernelGGL((conv_last_kernel<3>), grid, dim3(256), 0, s, u.ptr, u.tab, u.slope, w_torch, bias, out, h, w, tiles_x, tiles_y);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// sliding-window finalize: for each image pixel sum the (<= 2 x 2) window predictions that cover it IN WINDOW
// INDEX ORDER (the reference's `out[slice] += pred` order, so the fp32 sum is bit-identical), divide by the
// count, then softmax over classes, class-1 probability and the strict > 0.5 threshold.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int win_start(int i, int interval, int size, int roi) {
    const int s = i * interval;
    return s < size - roi ? s : size - roi;
}

template <int C>
__global__ __launch_bounds__(256) void swi_finalize_kernel(SwiFinalizeArgs a) {
    const cs_swi_plan& pl = a.plan;
    const int x = a.col_begin + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = a.row_begin + blockIdx.y;
    if (x >= a.col_end || y >= a.row_end) return;
    const int py = y + pl.pad_top, px = x + pl.pad_left;   // padded coordinates
    float sum[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sum[c] = 0.f;
    float cnt = 0.f;
    int ih_lo = (py - pl.roi_h) / pl.interval_h;
    ih_lo = ih_lo < 0 ? 0 : ih_lo;
    int iw_lo = (px - pl.roi_w) / pl.interval_w;
    iw_lo = iw_lo < 0 ? 0 : iw_lo;
    for (int ih = ih_lo; ih < pl.n_h; ++ih) {
        const int hs = win_start(ih, pl.interval_h, pl.pad_h, pl.roi_h);
        if (hs > py) break;
        if (py >= hs + pl.roi_h) continue;
        for (int iw = iw_lo; iw < pl.n_w; ++iw) {
            const int ws = win_start(iw, pl.interval_w, pl.pad_w, pl.roi_w);
            if (ws > px) break;
            if (px >= ws + pl.roi_w) continue;
            const float* p = a.win_logits + ((((size_t)(ih - a.ih_base) * pl.n_w + iw) * pl.roi_h + (py - hs)) * pl.roi_w + (px - ws)) * C;
#pragma unroll
            for (int c = 0; c < C; ++c) sum[c] += p[c];
            cnt += 1.f;
        }
    }
    float l[C], m = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        l[c] = sum[c] / cnt;
        m = fmaxf(m, l[c]);
    }
    const size_t o = (size_t)y * pl.img_w + x;
    if (a.logits_nchw) {
#pragma unroll
        for (int c = 0; c < C; ++c) a.logits_nchw[(size_t)c * pl.img_h * pl.img_w + o] = l[c];
    }
    if (a.prob1 || a.mask) {
        float den = 0.f, e1 = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float e = expf(l[c] - m);
            den += e;
            if (c == 1) e1 = e;
        }
        const float p1 = e1 / den;
        if (a.prob1) a.prob1[o] = p1;
        if (a.mask) a.mask[o] = p1 > 0.5f ? 1 : 0;
    }
}

int launch_swi_finalize(hipStream_t s, const SwiFinalizeArgs& a0) {
    SwiFinalizeArgs a = a0;
    if (a.col_end == 0 && a.col_begin == 0) a.col_end = a.plan.img_w;     // callers that finalise whole rows
    CS_REQUIRE(a.c == 3, "swi finalize: %d classes unsupported (3-class contract)", a.c);
    CS_REQUIRE(a.row_begin >= 0 && a.row_end <= a.plan.img_h && a.row_begin <= a.row_end, "swi finalize: bad row range");
    CS_REQUIRE(a.col_begin >= 0 && a.col_end <= a.plan.img_w && a.col_begin <= a.col_end, "swi finalize: bad column range");
    if (a.row_end == a.row_begin || a.col_end == a.col_begin) return CS_OK;
    const double px = (double)(a.row_end - a.row_begin) * (a.col_end - a.col_begin);
    ProfScope prof(K_SWI_FINALIZE, s, 0, px * (12.0 * 4 + 17.0));
    dim3 grid((unsigned)ceil_div(a.col_end - a.col_begin, 256), (unsigned)(a.row_end - a.row_begin));
    hipLaunchKernelGGL((swi_finalize_kernel<3>), grid, dim3(256), 0, s, a);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// Zero-fill of up to two regions in ONE launch (statistics accumulators, gradient buckets).  hipMemsetAsync goes through the runtime's blit
// path: a 2-5 us fill kernel with 8-20 us of dispatch gap around it on a stream of back-to-back 5-15 us kernels (training step timeline,
// profiles/train_timeline_r05_*.txt); a plain kernel of the stream's own code object queues like any other launch.
__global__ __launch_bounds__(256) void zero_kernel(uint4* a, size_t a16, uint4* b, size_t b16) {
    const size_t stride = (size_t)gridDim.x * 256;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a16 + b16; i += stride) {
        if (i < a16) a[i] = z;
        else b[i - a16] = z;
    }
}
int launch_zero(hipStream_t s, void* a, size_t a_bytes, void* b, size_t b_bytes) {
    if (!b) b_bytes = 0;
    if (a_bytes + b_bytes == 0) return CS_OK;
    // regions carved out of the callers' workspaces are 16-byte aligned and sized; anything else takes the runtime's fill
    if (((uintptr_t)a | (uintptr_t)b | a_bytes | b_bytes) & 15) {
        if (a_bytes) CS_HIP(hipMemsetAsync(a, 0, a_bytes, s));
        if (b_bytes) CS_HIP(hipMemsetAsync(b, 0, b_bytes, s));
        return CS_OK;
    }
    const size_t n16 = (a_bytes + b_bytes) >> 4;
    size_t blocks = (n16 + 256 * 4 - 1) / (256 * 4);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint4*)a, a_bytes >> 4, (uint4*)b, b_bytes >> 4);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
