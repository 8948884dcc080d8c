// gfx950 backward-pass kernels for the 3-class U-Net (row M-bwd of SURVEY.md 8(a): what `loss.backward()` at
// baseline/model_training_3class.py:260 dispatches through autograd for conv2d / conv_transpose2d /
// instance_norm / prelu / add / cat).
//
// Data gradients reuse the forward MFMA kernels (kernels_conv.hip) with re-packed weights:
//   dgrad(conv 3x3 s1)  = conv 3x3 s1 with W'[ci][co][8-tap]           (pack: transposed + flipped)
//   dgrad(conv 3x3 s2)  = transposed conv k3 s2 p1 op1 with Wt = W      (pack: transposed)
//   dgrad(convT k3 s2)  = conv 3x3 s2 with W' = Wt                      (pack: plain)
// This file adds what has no forward counterpart:
//   * weight gradients: MFMA GEMM with M = 16 "dy" channels, N = 16 "x" channels, K = pixels (4 per MFMA),
//     9 taps x MT accumulator tiles per wave, split-K over images / row bands, wave partials merged with LDS
//     float atomics and committed with one global float atomic per element and block
//   * InstanceNorm + PReLU backward: per-(image, channel) reductions sum(dz), sum(dz*z) and the PReLU slope gradient
//     in one pass, then dy = rstd * (dz - mean(dz) - z * mean(dz*z)) in a second pass
//   * per-channel column sums (bias gradients) and the VALU kernels for the two 3-channel layers at the top.
#include "kernels.h"

namespace cs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4b(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float prelu_b(float v, float slope) { return v >= 0.f ? v : v * slope; }
// four consecutive channels of an activation tensor at ELEMENT offset `off`: fp32, or - the bf16 tape of the bf16 training mode (the
// forward kept its activations as bf16 in HBM) - four bf16 values widened to fp32.  Compile-time: a branch inside the load groups of
// these kernels would split the groups (all loads of a trip are issued before the first use).
template <bool H16>
__device__ __forceinline__ f32x4 ld_act4(const float* base, size_t off) {
    if constexpr (H16) return bf16x4_to_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off));
    else return *reinterpret_cast<const f32x4*>(base + off);
}

// ------------------------------------------------------------------------------------------------
// InstanceNorm + PReLU backward
//   z = y*scale + shift ; a = prelu(z) ; dz = da * (z >= 0 ? 1 : slope) ; dslope += sum(da * z [z < 0])
//   red[n][c] = (sum dz, sum dz*z) ;  dy = scale * (dz - red0/m - z * red1/m)
// ------------------------------------------------------------------------------------------------
struct NormBwdParams {
    const float* da;   int da_pitch;
    const float* y;    int y_pitch;   int y16;   // y16: y holds bf16 elements (pitch still counts elements)
    const float2* tab; int tab_c, tab_off;
    const float* slope;
    double* red;       // [n][c][2]
    double* dslope;    // NORM_DSLOPE_SLOTS partial accumulators (or null for ReLU): blocks spread their atomics over the slots -
                       // thousands of fp64 atomics on ONE address serialised at ~35 ns each (70 us of an 80 us pass at 8 x 128^2 x 16)
    float* dy;         int dy_pitch;
    int c, npix;       // channels, pixels per image
    int pix_per_block;
    const float2* xtab;   // BatchNorm: (rstd, -mean*rstd) so that x_hat = y*rstd - mean*rstd; null: x_hat == z (InstanceNorm, no affine)
    double inv_m;         // 1 / (elements behind each pair of sums): npix (InstanceNorm) or n*npix (BatchNorm, sums reduced over n)
    unsigned* amax;       // bits of max |dy| (atomicMax; non-negative floats order like unsigned integers) or null
    double* dasum;        // [n][c] sums of da itself over the image's pixels, or null: `da` is also the gradient of a conv bias (the raw
                          // residual branch) - its column sums ride on this pass instead of a pass of their own (zero on entry)
};

// wide: c % 4 == 0 and 256 % (c/4) == 0
template <bool Y16>
__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(NormBwdParams p) {
    __shared__ float sh[256 * 13];
    const int c4n = p.c >> 2, lanes = 256 / c4n;
    const int quad = threadIdx.x % c4n, pl = threadIdx.x / c4n;
    const int img = blockIdx.y;
    const int64_t pix0 = (int64_t)blockIdx.x * p.pix_per_block;
    const int64_t pix1 = pix0 + p.pix_per_block < p.npix ? pix0 + p.pix_per_block : p.npix;
    const f32x4* t = reinterpret_cast<const f32x4*>(p.tab + (size_t)img * p.tab_c + p.tab_off + 4 * quad);
    const f32x4 t0 = t[0], t1 = t[1];
    const float sc[4] = {t0.x, t0.z, t1.x, t1.z}, sf[4] = {t0.y, t0.w, t1.y, t1.w};
    float xs[4] = {sc[0], sc[1], sc[2], sc[3]}, xf[4] = {sf[0], sf[1], sf[2], sf[3]};     // x_hat table (== the z table without affine)
    if (p.xtab) {
        const f32x4* xt = reinterpret_cast<const f32x4*>(p.xtab + (size_t)img * p.tab_c + p.tab_off + 4 * quad);
        const f32x4 x0 = xt[0], x1 = xt[1];
        xs[0] = x0.x; xs[1] = x0.z; xs[2] = x1.x; xs[3] = x1.z;
        xf[0] = x0.y; xf[1] = x0.w; xf[2] = x1.y; xf[3] = x1.w;
    }
    const float slope = p.slope ? *p.slope : 0.f;
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, s0[4] = {0, 0, 0, 0}, ds = 0.f;
    // 4 pixels per trip, all 8 loads issued before the first use: the pass is bound by loads in flight, not by arithmetic
    for (int64_t px = pix0 + pl; px < pix1; px += 4 * lanes) {
        f32x4 da[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t pu = px + u * lanes < pix1 ? px + u * lanes : px;       // clamped: the tail re-reads a valid pixel
            const int64_t g = (int64_t)img * p.npix + pu;
            da[u] = *reinterpret_cast<const f32x4*>(p.da + g * p.da_pitch + 4 * quad);
            y[u] = ld_act4<Y16>(p.y, (size_t)(g * p.y_pitch + 4 * quad));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (px + u * lanes >= pix1) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = y[u][j] * sc[j] + sf[j];
                const float xh = y[u][j] * xs[j] + xf[j];
                const float dz = z >= 0.f ? da[u][j] : da[u][j] * slope;
                if (z < 0.f) ds += da[u][j] * z;
                s1[j] += dz;
                s2[j] += dz * xh;
                s0[j] += da[u][j];
            }
        }
    }
    // reduce over the `lanes` threads that share a channel quad
    float* mine = sh + threadIdx.x * 13;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mine[j] = s1[j];
        mine[4 + j] = s2[j];
        mine[9 + j] = s0[j];
    }
    mine[8] = ds;
    __syncthreads();
    if (threadIdx.x < c4n) {
        float r1[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0}, r0[4] = {0, 0, 0, 0};
        for (int l = 0; l < lanes; ++l) {
            const float* o = sh + (l * c4n + threadIdx.x) * 13;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r1[j] += o[j];
                r2[j] += o[4 + j];
                r0[j] += o[9 + j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double* d = p.red + ((size_t)img * p.c + 4 * threadIdx.x + j) * 2;
            atomicAdd(d, (double)r1[j]);
            atomicAdd(d + 1, (double)r2[j]);
            if (p.dasum) atomicAdd(p.dasum + (size_t)img * p.c + 4 * threadIdx.x + j, (double)r0[j]);
        }
    }
    if (p.dslope) {
        float tot = 0.f;
        if (threadIdx.x < 64) {
            for (int l = threadIdx.x; l < 256; l += 64) tot += sh[l * 13 + 8];
            for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off);
            if (threadIdx.x == 0) atomicAdd(p.dslope + ((blockIdx.x + 13 * blockIdx.y) & (NORM_DSLOPE_SLOTS - 1)), (double)tot);
        }
    }
}

// apply pass: dy = sc * (dz - mean(dz) - x_hat * mean(dz * x_hat)).  Same thread layout as the reduce pass (a thread keeps its
// channel quad for all its pixels, blockIdx.y = image), so the tables and the two means are loaded and converted ONCE per
// thread; the round-1 version was a flat grid-stride loop that paid two 64-bit divisions and eight fp64 loads per element quad.
template <bool Y16>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(NormBwdParams p) {
    const int c4n = p.c >> 2, lanes = 256 / c4n;
    const int quad = threadIdx.x % c4n, pl = threadIdx.x / c4n;
    const int img = blockIdx.y;
    const int64_t pix0 = (int64_t)blockIdx.x * p.pix_per_block;
    const int64_t pix1 = pix0 + p.pix_per_block < p.npix ? pix0 + p.pix_per_block : p.npix;
    const f32x4* t = reinterpret_cast<const f32x4*>(p.tab + (size_t)img * p.tab_c + p.tab_off + 4 * quad);
    const f32x4 t0 = t[0], t1 = t[1];
    const float sc[4] = {t0.x, t0.z, t1.x, t1.z}, sf[4] = {t0.y, t0.w, t1.y, t1.w};
    float xs[4] = {sc[0], sc[1], sc[2], sc[3]}, xf[4] = {sf[0], sf[1], sf[2], sf[3]};
    if (p.xtab) {
        const f32x4* xt = reinterpret_cast<const f32x4*>(p.xtab + (size_t)img * p.tab_c + p.tab_off + 4 * quad);
        const f32x4 x0 = xt[0], x1 = xt[1];
        xs[0] = x0.x; xs[1] = x0.z; xs[2] = x1.x; xs[3] = x1.z;
        xf[0] = x0.y; xf[1] = x0.w; xf[2] = x1.y; xf[3] = x1.w;
    }
    const double* red = p.red + ((size_t)img * p.c + 4 * quad) * 2;
    float m1[4], m2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m1[j] = (float)(red[2 * j] * p.inv_m);
        m2[j] = (float)(red[2 * j + 1] * p.inv_m);
    }
    const float slope = p.slope ? *p.slope : 0.f;
    float amax = 0.f;
    for (int64_t px = pix0 + pl; pl < lanes && px < pix1; px += 4 * lanes) {
        f32x4 da[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t pu = px + u * lanes < pix1 ? px + u * lanes : px;
            const int64_t g = (int64_t)img * p.npix + pu;
            da[u] = *reinterpret_cast<const f32x4*>(p.da + g * p.da_pitch + 4 * quad);
            y[u] = ld_act4<Y16>(p.y, (size_t)(g * p.y_pitch + 4 * quad));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (px + u * lanes >= pix1) continue;
            const int64_t g = (int64_t)img * p.npix + px + u * lanes;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = y[u][j] * sc[j] + sf[j];
                const float xh = y[u][j] * xs[j] + xf[j];
                const float dz = z >= 0.f ? da[u][j] : da[u][j] * slope;
                o[j] = sc[j] * (dz - m1[j] - xh * m2[j]);
            }
            *reinterpret_cast<f32x4*>(p.dy + g * p.dy_pitch + 4 * quad) = o;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (p.amax) amax_commit(p.amax, amax);           // (block-uniform branch: every thread of the block gets here)
}

// One-launch form for the small tensors of the deep levels (InstanceNorm, npix <= NORM_BWD_FUSED_MAX_PIX): a block OWNS (image,
// QB channel quads), so the sums of the reduce pass need no atomics and no second launch - pass 1 sums over the image's pixels, the
// waves meet through LDS, pass 2 re-reads the same (L2-resident) elements and writes dy.  The two-kernel form spends ~19 us per norm
// site on such tensors (launch + atomics + a dependent launch); 10 of the 16 sites of the reference-width batch-8 step qualify.
template <int QB, bool Y16>
__global__ __launch_bounds__(256) void norm_bwd_fused_kernel(NormBwdParams p) {
    constexpr int LANES = 256 / QB;                      // pixel lanes per quad
    __shared__ double sh[4][QB][13];
    const int quad_l = threadIdx.x % QB, pl = threadIdx.x / QB;
    const int quad = blockIdx.x * QB + quad_l, img = blockIdx.y;
    const int wave = threadIdx.x >> 6;
    const f32x4* t = reinterpret_cast<const f32x4*>(p.tab + (size_t)img * p.tab_c + p.tab_off + 4 * quad);
    const f32x4 t0 = t[0], t1 = t[1];
    const float sc[4] = {t0.x, t0.z, t1.x, t1.z}, sf[4] = {t0.y, t0.w, t1.y, t1.w};
    const float slope = p.slope ? *p.slope : 0.f;
    const float* const da0 = p.da + (size_t)img * p.npix * p.da_pitch + 4 * quad;
    const size_t y0 = (size_t)img * p.npix * p.y_pitch + 4 * quad;      // element offset (fp32 or bf16 elements)
    float v[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};            // s1[4], s2[4], ds, sum of da[4]
    // four pixels per trip, their eight loads issued before the first use (one load pair in flight per thread ran these small tensors
    // at 40 % of the rate the two-kernel form reaches: 42 us per launch for 67 MB at 32 x 32^2 x 128)
    for (int px = pl; px < p.npix; px += 4 * LANES) {
        f32x4 da[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pu = px + u * LANES < p.npix ? px + u * LANES : px;       // clamped: the tail re-reads a valid pixel
            da[u] = *reinterpret_cast<const f32x4*>(da0 + (size_t)pu * p.da_pitch);
            y[u] = ld_act4<Y16>(p.y, y0 + (size_t)pu * p.y_pitch);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (px + u * LANES >= p.npix) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = y[u][j] * sc[j] + sf[j];
                const float dz = z >= 0.f ? da[u][j] : da[u][j] * slope;
                if (z < 0.f) v[8] += da[u][j] * z;
                v[j] += dz;
                v[4 + j] += dz * z;
                v[9 + j] += da[u][j];
            }
        }
    }
    // lanes of a wave that share a quad sit QB apart
#pragma unroll
    for (int k = 0; k < 13; ++k)
#pragma unroll
        for (int off = QB; off < 64; off <<= 1) v[k] += __shfl_xor(v[k], off);
    if ((threadIdx.x & 63) < QB)
#pragma unroll
        for (int k = 0; k < 13; ++k) sh[wave][quad_l][k] = (double)v[k];
    __syncthreads();
    if (p.dasum && threadIdx.x < QB)                     // the block owns (image, quad): plain stores
#pragma unroll
        for (int j = 0; j < 4; ++j)
            p.dasum[(size_t)img * p.c + 4 * quad + j] = (sh[0][quad_l][9 + j] + sh[1][quad_l][9 + j]) + (sh[2][quad_l][9 + j] + sh[3][quad_l][9 + j]);
    float m1[4], m2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m1[j] = (float)(((sh[0][quad_l][j] + sh[1][quad_l][j]) + (sh[2][quad_l][j] + sh[3][quad_l][j])) * p.inv_m);
        m2[j] = (float)(((sh[0][quad_l][4 + j] + sh[1][quad_l][4 + j]) + (sh[2][quad_l][4 + j] + sh[3][quad_l][4 + j])) * p.inv_m);
    }
    if (p.dslope && threadIdx.x < QB) {
        const double ds = (sh[0][quad_l][8] + sh[1][quad_l][8]) + (sh[2][quad_l][8] + sh[3][quad_l][8]);
        atomicAdd(p.dslope + ((blockIdx.x * QB + quad_l + 13 * blockIdx.y) & (NORM_DSLOPE_SLOTS - 1)), ds);
    }
    float amax = 0.f;
    float* const dy0 = p.dy + (size_t)img * p.npix * p.dy_pitch + 4 * quad;
    for (int px = pl; px < p.npix; px += 4 * LANES) {
        f32x4 da[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pu = px + u * LANES < p.npix ? px + u * LANES : px;
            da[u] = *reinterpret_cast<const f32x4*>(da0 + (size_t)pu * p.da_pitch);
            y[u] = ld_act4<Y16>(p.y, y0 + (size_t)pu * p.y_pitch);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (px + u * LANES >= p.npix) continue;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = y[u][j] * sc[j] + sf[j];
                const float dz = z >= 0.f ? da[u][j] : da[u][j] * slope;
                o[j] = sc[j] * (dz - m1[j] - z * m2[j]);
            }
            *reinterpret_cast<f32x4*>(dy0 + (size_t)(px + u * LANES) * p.dy_pitch) = o;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (p.amax) amax_commit(p.amax, amax);           // (block-uniform branch)
}

// narrow (c <= 4, any pitch): one thread per pixel
template <int C>
__global__ __launch_bounds__(256) void norm_bwd_reduce_small_kernel(NormBwdParams p) {
    const int img = blockIdx.y;
    const float slope = p.slope ? *p.slope : 0.f;
    float s1[C], s2[C], ds = 0.f, sc[C], sf[C], xs[C], xf[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float2 t = p.tab[(size_t)img * p.tab_c + p.tab_off + c];
        const float2 x = p.xtab ? p.xtab[(size_t)img * p.tab_c + p.tab_off + c] : t;
        sc[c] = t.x; sf[c] = t.y; xs[c] = x.x; xf[c] = x.y; s1[c] = 0.f; s2[c] = 0.f;
    }
    for (int64_t px = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; px < p.npix; px += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = (int64_t)img * p.npix + px;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float da = p.da[g * p.da_pitch + c];
            const float yv = p.y[g * p.y_pitch + c];
            const float z = yv * sc[c] + sf[c];
            const float dz = z >= 0.f ? da : da * slope;
            if (z < 0.f) ds += da * z;
            s1[c] += dz;
            s2[c] += dz * (yv * xs[c] + xf[c]);
        }
    }
    // wave shuffle -> block LDS -> ONE fp64 atomic per block and sum (per-wave atomics on the same 2C words of an image serialise
    // at the memory side: 8192 waves x 6 atomics made this pass 107 us)
    __shared__ float blk[4][2 * C + 1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        for (int off = 32; off >= 1; off >>= 1) {
            s1[c] += __shfl_xor(s1[c], off);
            s2[c] += __shfl_xor(s2[c], off);
        }
        if ((threadIdx.x & 63) == 0) {
            blk[threadIdx.x >> 6][2 * c] = s1[c];
            blk[threadIdx.x >> 6][2 * c + 1] = s2[c];
        }
    }
    for (int off = 32; off >= 1; off >>= 1) ds += __shfl_xor(ds, off);
    if ((threadIdx.x & 63) == 0) blk[threadIdx.x >> 6][2 * C] = ds;
    __syncthreads();
    if (threadIdx.x < 2 * C)
        atomicAdd(p.red + (size_t)img * C * 2 + threadIdx.x,
                  (double)blk[0][threadIdx.x] + (double)blk[1][threadIdx.x] + (double)blk[2][threadIdx.x] + (double)blk[3][threadIdx.x]);
    if (p.dslope && threadIdx.x == 2 * C)
        atomicAdd(p.dslope + ((blockIdx.x + 13 * blockIdx.y) & (NORM_DSLOPE_SLOTS - 1)),
                  (double)blk[0][2 * C] + (double)blk[1][2 * C] + (double)blk[2][2 * C] + (double)blk[3][2 * C]);
}

template <int C>
__global__ __launch_bounds__(256) void norm_bwd_apply_small_kernel(NormBwdParams p) {
    const int img = blockIdx.y;
    const int64_t px = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= p.npix) return;
    const float slope = p.slope ? *p.slope : 0.f;
    const double inv_m = p.inv_m;
    const int64_t g = (int64_t)img * p.npix + px;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float2 t = p.tab[(size_t)img * p.tab_c + p.tab_off + c];
        const float2 x = p.xtab ? p.xtab[(size_t)img * p.tab_c + p.tab_off + c] : t;
        const float da = p.da[g * p.da_pitch + c];
        const float yv = p.y[g * p.y_pitch + c];
        const float z = yv * t.x + t.y;
        const float dz = z >= 0.f ? da : da * slope;
        const double* red = p.red + ((size_t)img * C + c) * 2;
        p.dy[g * p.dy_pitch + c] = t.x * (dz - (float)(red[0] * inv_m) - (yv * x.x + x.y) * (float)(red[1] * inv_m));
    }
}

// BatchNorm: the per-image sums are reduced over the batch (every image's slot receives the batch total, which is what the
// apply pass then reads with inv_m = 1 / (n * npix)); d(gamma) = sum dz * x_hat, d(beta) = sum dz fall out of the same sums.
__global__ void bn_bwd_finalize_kernel(double* red, int n, int c, float* dgamma, float* dbeta) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < n; ++i) {
        s1 += red[((size_t)i * c + ch) * 2];
        s2 += red[((size_t)i * c + ch) * 2 + 1];
    }
    for (int i = 0; i < n; ++i) {
        red[((size_t)i * c + ch) * 2] = s1;
        red[((size_t)i * c + ch) * 2 + 1] = s2;
    }
    if (dbeta) dbeta[ch] = (float)s1;
    if (dgamma) dgamma[ch] = (float)s2;
}

int launch_norm_bwd(hipStream_t s, const ConvSrc& y, const float* da, int da_pitch, float* dy, int dy_pitch, double* red,
                    double* dslope, int64_t npix, int n, const float2* xtab, bool batch, float* dgamma, float* dbeta, float* amax,
                    double* dasum) {
    CS_REQUIRE(y.tab != nullptr, "norm_bwd: the forward table is required");
    CS_REQUIRE(!batch || xtab, "norm_bwd: BatchNorm needs the forward's x_hat table");
    NormBwdParams p{};
    CS_REQUIRE(!y.q8 && !y.ptr2, "norm_bwd: fp32 or bf16 activations only");
    p.da = da; p.da_pitch = da_pitch; p.y = y.ptr; p.y_pitch = y.pitch; p.y16 = y.h16 ? 1 : 0;
    p.tab = y.tab; p.tab_c = y.tab_c ? y.tab_c : y.c; p.tab_off = y.tab_off; p.slope = y.slope;
    p.red = red; p.dslope = y.slope ? dslope : nullptr; p.dy = dy; p.dy_pitch = dy_pitch; p.c = y.c; p.npix = (int)npix;
    p.xtab = xtab;
    p.amax = reinterpret_cast<unsigned*>(amax);
    p.dasum = dasum;
    CS_REQUIRE(!dasum || y.c != 3, "norm_bwd: the column sums of da ride on the wide kernels only");
    p.inv_m = 1.0 / ((double)npix * (batch ? (double)n : 1.0));
    ProfScope prof(K_NORM_BWD, s, 0, (y.h16 ? 16.0 : 20.0) * n * npix * y.c);
    auto bn_finalize = [&]() {
        if (batch) hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)ceil_div(y.c, 64)), dim3(64), 0, s, red, n, y.c, dgamma, dbeta);
    };
    if (y.c == 3) {
        CS_REQUIRE(!y.h16, "norm_bwd: the 3-channel tensors are fp32 in every mode");
        dim3 grid((unsigned)(ceil_div((int)npix, 256) < 64 ? ceil_div((int)npix, 256) : 64), (unsigned)n);
        hipLaunchKernelGGL((norm_bwd_reduce_small_kernel<3>), grid, dim3(256), 0, s, p);
        bn_finalize();
        hipLaunchKernelGGL((norm_bwd_apply_small_kernel<3>), dim3((unsigned)ceil_div((int)npix, 256), (unsigned)n), dim3(256), 0, s, p);
    } else {
        const int c4n = y.c / 4;
        CS_REQUIRE(y.c % 4 == 0 && c4n <= 256 && 256 % c4n == 0, "norm_bwd: %d channels unsupported (need a power of two in 16..1024)", y.c);
        // enough blocks to keep ~30 KB of loads in flight per CU: ~4096 blocks over the batch (the round-1 choice of 1024 pixels
        // per block left 2 blocks per CU on the large tensors and ran the pass at 1 TB/s)
        // ... but never so many that a block's share shrinks to a handful of elements per thread: at 8 x 128^2 x 16 channels 2048 blocks
        // held 4 elements per thread and the pass was nothing but 65 k fp64 atomics on 256 addresses (70 us for 17 MB; 16 of the 172
        // launches of the reference-width training step were a fifth of its time) - at least `min_elems` elements per thread
        if (!batch && npix <= NORM_BWD_FUSED_MAX_PIX) {
            // widest channel group that still gives the launch >= 128 blocks
            // (round 5, 32-base x 32: 8 quads per block 2.863 ms per bf16 step, 4: 2.900, 16: 2.903; MORE blocks of fewer quads lose - 2.93 / 3.10 /
            // 3.27 ms at >= 512 / 1024 / 4096 blocks: a pixel's channels are contiguous, narrow groups waste the lines they touch)
            const int qb = (c4n % 8 == 0 && (c4n / 8) * n >= 128) ? 8 :
                           (c4n % 4 == 0 && (c4n / 4) * n >= 128) ? 4 : (c4n % 2 == 0 && (c4n / 2) * n >= 128) ? 2 : 1;
            const dim3 grid((unsigned)(c4n / qb), (unsigned)n);
            if (y.h16) {
                if (qb == 8) hipLaunchKernelGGL((norm_bwd_fused_kernel<8, true>), grid, dim3(256), 0, s, p);
                else if (qb == 4) hipLaunchKernelGGL((norm_bwd_fused_kernel<4, true>), grid, dim3(256), 0, s, p);
                else if (qb == 2) hipLaunchKernelGGL((norm_bwd_fused_kernel<2, true>), grid, dim3(256), 0, s, p);
                else hipLaunchKernelGGL((norm_bwd_fused_kernel<1, true>), grid, dim3(256), 0, s, p);
            } else {
                if (qb == 8) hipLaunchKernelGGL((norm_bwd_fused_kernel<8, false>), grid, dim3(256), 0, s, p);
                else if (qb == 4) hipLaunchKernelGGL((norm_bwd_fused_kernel<4, false>), grid, dim3(256), 0, s, p);
                else if (qb == 2) hipLaunchKernelGGL((norm_bwd_fused_kernel<2, false>), grid, dim3(256), 0, s, p);
                else hipLaunchKernelGGL((norm_bwd_fused_kernel<1, false>), grid, dim3(256), 0, s, p);
            }
            CS_HIP(hipGetLastError());
            return CS_OK;
        }
        auto chunk = [&](int64_t target_blocks, int min_elems) {
            int64_t ppb = (npix * n + target_blocks - 1) / target_blocks;
            const int64_t ppb_min = (int64_t)min_elems * 256 / y.c;
            ppb = ppb < ppb_min ? ppb_min : ppb;
            ppb = ppb < 64 ? 64 : ppb;
            ppb = (ppb + 63) / 64 * 64;
            return (int)(ppb < npix ? ppb : npix);
        };
        p.pix_per_block = chunk(2048, 64);
        dim3 grid((unsigned)ceil_div((int)npix, p.pix_per_block), (unsigned)n);
        if (y.h16) hipLaunchKernelGGL(norm_bwd_reduce_kernel<true>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(norm_bwd_reduce_kernel<false>, grid, dim3(256), 0, s, p);
        bn_finalize();
        NormBwdParams pa = p;                                // finer pixel chunks: the apply pass has no per-block reduction to amortise
        pa.pix_per_block = chunk(4096, 16);
        const dim3 ga((unsigned)ceil_div((int)npix, pa.pix_per_block), (unsigned)n);
        if (y.h16) hipLaunchKernelGGL(norm_bwd_apply_kernel<true>, ga, dim3(256), 0, s, pa);
        else hipLaunchKernelGGL(norm_bwd_apply_kernel<false>, ga, dim3(256), 0, s, pa);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// column sums: dst[c] = sum over pixels of src[pix][c]   (bias gradients of the convs that do not feed a norm).
// Two stages without atomics: every block reduces its pixel range to one fp32 row of `partial`, a second kernel sums the
// rows in fp64.  (Round 1 added every block's row into the same c doubles with global atomics: 512 blocks serialised on
// 32..512 addresses and the pass ran at ~110 GB/s.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* src, int pitch, int c, int64_t npix, int64_t pix_per_block,
                                                             float* partial) {
    __shared__ f32x4 sh[256];
    const int c4n = c >> 2;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
    for (int q0 = 0; q0 < c4n; q0 += 256) {                   // channel quads handled 256 at a time (c <= 1024: one trip)
        const int cq = c4n - q0 < 256 ? c4n - q0 : 256;       // quads in this trip
        const int lanes = 256 / cq;
        const int quad = threadIdx.x % cq, pl = threadIdx.x / cq;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pl < lanes)
            for (int64_t px = p0 + pl; px < p1; px += lanes) acc += *reinterpret_cast<const f32x4*>(src + px * pitch + 4 * (q0 + quad));
        sh[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < cq) {
            f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int l = 0; l < lanes; ++l) t += sh[l * cq + threadIdx.x];
            *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.x * c + 4 * (q0 + threadIdx.x)) = t;
        }
        __syncthreads();
    }
}

// block = 8 channels x 32 row lanes: the rows of `partial` are walked by 32 lanes in parallel (a single thread per channel
// walking 1024 rows was latency-bound: 0.1 ms per launch), LDS tree over the lanes, fp64 accumulation
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* partial, int rows, int c, float* dst) {
    __shared__ double sh[256];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int ch = blockIdx.x * 8 + cl;
    double acc = 0.0;
    if (ch < c)
        for (int r = rl; r < rows; r += 32) acc += (double)partial[(size_t)r * c + ch];
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0 && ch < c) {
        double t = 0.0;
#pragma unroll
        for (int l = 0; l < 32; ++l) t += sh[l * 8 + cl];
        dst[ch] = (float)t;
    }
}

int colsum_blocks(int64_t npix_total) {
    const int64_t b = (npix_total + 255) / 256;
    return (int)(b < 1024 ? (b < 1 ? 1 : b) : 1024);
}

int launch_colsum(hipStream_t s, const float* src, int pitch, int c, int64_t npix_total, float* partial, float* dst) {
    CS_REQUIRE(c % 4 == 0 && pitch % 4 == 0, "colsum: %d channels / pitch %d must be multiples of 4", c, pitch);
    const int blocks = colsum_blocks(npix_total);
    const int64_t ppb = (npix_total + blocks - 1) / blocks;
    ProfScope prof(K_COLSUM, s, 0, 4.0 * npix_total * c);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, pitch, c, npix_total, ppb, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)ceil_div(c, 8)), dim3(256), 0, s, partial, blocks, c, dst);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// weight gradient (MFMA).  dw[tap][co_off + i][ci_off + j] += sum_pixels A[pix][i] * B[pix*S + tap - pad][j]
//   A ("dy-like"): NHWC raw, 16*MT channels per block.  B ("x-like"): one 16-channel chunk (lazy norm/act applied
//   while staging) or a narrow / strided source with <= 4 channels (first layer input, 3-channel tensors).
// ------------------------------------------------------------------------------------------------
struct WgradParams {
    const float* a;
    int a_pitch;
    int a16, b16;                // the operand holds bf16 elements (an activation tensor of the bf16 tape; pitches count elements)
    // B, mode 0: NHWC 16-channel chunks
    const float* b_ptr;
    const float2* b_tab;
    const float* b_slope;
    int b_pitch, b_tab_c, b_tab_off;
    // B, mode 1: narrow / strided (element strides), b_c <= 4 channels
    int b_mode, b_c;
    int64_t b_sn, b_sc, b_sh, b_sw;
    const float* dummy;          // valid readable memory for the branch-free table loads
    float* part;                 // [split_k][taps][co_total][ci_total] partial sums (plain stores, no atomics)
    int co_total, ci_total, co_off, ci_off;
    int n, ho, wo, hi, wi;
    int tiles_x, tiles_y, split_k, total_tiles;
};

// Split-K over pixels WITHOUT atomics: block k reduces its contiguous range of (image, tile) pairs and stores one partial
// [taps][16*MT][16] slab; the unpack kernel sums the split_k partials while it converts to the PyTorch layout.  (A first
// version committed every block with global float atomics: hundreds of blocks adding to the same few KB serialised at
// the memory side and every launch took 110-260 us whatever its size.)
template <int S, int KS, int MT, bool NARROW, int TH = 4, bool B16 = false>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WgradParams p) {
    constexpr int TW = 16;
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int PAD = KS / 2, TAPS = KS * KS;
    constexpr int COP = (MT & 1) ? 16 * MT : 16 * MT + 16;   // dy pixel pitch: == 16 (mod 32) -> conflict-free A reads
    constexpr int CSX = (S == 1) ? 16 : 24;                   // x pixel pitch: S*CSX == 16 (mod 32)
    constexpr int A_ITERS = (TH * TW * 4 * MT + 255) / 256;   // float4 per thread for the dy tile
    constexpr int B_ITERS = (IH * IW + 63) / 64;
    constexpr int A_FLOATS = TH * TW * COP, B_FLOATS = (IH * IW * CSX + 3) / 4 * 4;
    constexpr int NACC = (TAPS * MT + 3) / 4;              // accumulator tiles (tap, co-tile) owned by one wave
    __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + B_FLOATS];
    float* const a_s = smem;
    float* const b_s = smem + A_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int ks_id = blockIdx.x;               // split-K slice
    const int co0 = blockIdx.y * 16 * MT;       // A channel base of this block
    const int cb0 = blockIdx.z * 16;            // B channel chunk
    const int q = tid & 3;
    const int tiles_per_img = p.tiles_x * p.tiles_y;

    // The 9*MT (tap, co-tile) output tiles of the block are dealt round-robin to the 4 waves; every wave walks ALL pixel
    // rows of the tile for its own outputs, so no cross-wave reduction (and no LDS atomics) is needed at the end.
    f32x4 acc[NACC];
    int a_off[NACC], b_off[NACC];                          // LDS offsets of this wave's tiles: co-tile column / tap shift
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int t = wave + 4 * i;                        // tile id = tap * MT + m
        const int tap = (t < TAPS * MT) ? t / MT : 0, m = (t < TAPS * MT) ? t - tap * MT : 0;
        const int ky = tap / KS, kx = tap - ky * KS;
        a_off[i] = m * 16 + li;
        b_off[i] = (ky * IW + kx) * CSX + li;
    }

    const bool lazy = !NARROW && p.b_tab != nullptr;
    f32x4 ra[A_ITERS], rb[B_ITERS];
    f32x4 t0, t1;                 // lazy norm/act table of the PREFETCHED tile's image; applied when the tile is stored to LDS
    float slope = 0.f;
    unsigned ok_a = 0, ok_b = 0;  // validity bits of the prefetched tile
    // global -> registers only: consuming a loaded value here would put the s_waitcnt in front of the MFMAs and
    // serialise every tile behind its own load latency
    auto load_tile = [&](int tile) {
        const int img = tile / tiles_per_img, tl = tile - img * tiles_per_img;
        const int tile_y = tl / p.tiles_x, tile_x = tl - tile_y * p.tiles_x;
        const int oy0 = tile_y * TH, ox0 = tile_x * TW;
        const f32x4* t = lazy ? reinterpret_cast<const f32x4*>(p.b_tab + (size_t)img * p.b_tab_c + p.b_tab_off + cb0 + 4 * q)
                              : reinterpret_cast<const f32x4*>(p.dummy);
        t0 = t[0];
        t1 = t[1];
        const float* sp = (lazy && p.b_slope) ? p.b_slope : p.dummy;
        slope = *sp;
        ok_a = 0;
        ok_b = 0;
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
            const int i = tid + 256 * it;
            const int px = i / (4 * MT), c4 = i - px * (4 * MT);
            const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
            const bool ok = (A_ITERS * 256 == TH * TW * 4 * MT || px < TH * TW) && oy < p.ho && ox < p.wo;
            const size_t off = ok ? (((size_t)img * p.ho + oy) * p.wo + ox) * p.a_pitch + co0 + 4 * c4 : 0;
            ra[it] = *reinterpret_cast<const f32x4*>(p.a + off);
            ok_a |= (ok ? 1u : 0u) << it;
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int pix = (tid >> 2) + 64 * it;
            const int iy = pix / IW, ix = pix - iy * IW;
            const int gy = oy0 * S - PAD + iy, gx = ox0 * S - PAD + ix;
            const bool ok = pix < IH * IW && gy >= 0 && gy < p.hi && gx >= 0 && gx < p.wi;
            if (!NARROW) {
                const size_t off = ok ? (((size_t)img * p.hi + gy) * p.wi + gx) * p.b_pitch + cb0 + 4 * q : 0;
                rb[it] = ld_act4<B16>(p.b_ptr, off);
                ok_b |= (ok ? 1u : 0u) << it;
            } else {
                const bool use = q == 0 && ok;
                const float* b = p.b_ptr + (use ? img * p.b_sn + gy * p.b_sh + gx * p.b_sw : 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) rb[it][c] = b[(c < p.b_c ? c : 0) * p.b_sc];
                ok_b |= (use ? 1u : 0u) << it;
            }
        }
    };
    auto store_tile = [&]() {
        const float sl = (lazy && p.b_slope) ? slope : 0.f;
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
            const int i = tid + 256 * it;
            const int px = i / (4 * MT), c4 = i - px * (4 * MT);
            if (A_ITERS * 256 == TH * TW * 4 * MT || px < TH * TW)
                *reinterpret_cast<f32x4*>(&a_s[px * COP + 4 * c4]) = ((ok_a >> it) & 1u) ? ra[it] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int pix = (tid >> 2) + 64 * it;
            if (pix < IH * IW) {
                f32x4 v = rb[it];
                if (!NARROW) {
                    f32x4 r;
                    r.x = prelu_b(v.x * t0.x + t0.y, sl);
                    r.y = prelu_b(v.y * t0.z + t0.w, sl);
                    r.z = prelu_b(v.z * t1.x + t1.y, sl);
                    r.w = prelu_b(v.w * t1.z + t1.w, sl);
                    v = lazy ? r : v;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c >= p.b_c) v[c] = 0.f;
                }
                *reinterpret_cast<f32x4*>(&b_s[pix * CSX + 4 * q]) = ((ok_b >> it) & 1u) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // contiguous slice of the (image, tile) list for this split-K block
    const int t_begin = (int)((int64_t)ks_id * p.total_tiles / p.split_k);
    const int t_end = (int)((int64_t)(ks_id + 1) * p.total_tiles / p.split_k);
    if (t_begin < t_end) {
        load_tile(t_begin);
        store_tile();
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        const bool more = tile + 1 < t_end;
        if (more) load_tile(tile + 1);
        // 16 K-steps (4 pixel rows x 4 groups of 4 pixels).  All tiles of a wave share ONE dy channel tile ((wave + 4 i) % MT does
        // not depend on i), so a step needs one A value and NACC B values; they are read from LDS one step AHEAD of their MFMAs
        // (left to itself hipcc put every operand pair and an lgkmcnt(0) wait directly in front of its MFMA: each MFMA then paid a
        // full LDS round trip and a wave reached about a quarter of the matrix rate)
        float a_op[2], b_op[2][NACC];
        auto read_ops = [&](int step, float& av, float (&bv)[NACC]) {
            const int row = step >> 2, ks = step & 3;
            av = a_s[(row * 16 + 4 * ks + kk) * COP + a_off[0]];
            const int pb = (row * S * IW + (4 * ks + kk) * S) * CSX;
#pragma unroll
            for (int i = 0; i < NACC; ++i) bv[i] = b_s[pb + b_off[i]];
        };
        read_ops(0, a_op[0], b_op[0]);
#pragma unroll
        for (int step = 0; step < 4 * TH; ++step) {
            if (step + 1 < 4 * TH) read_ops(step + 1, a_op[(step + 1) & 1], b_op[(step + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);       // the reads stay in front of this step's MFMAs
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                if ((TAPS * MT) % 4 == 0 || wave + 4 * i < TAPS * MT) acc[i] = mfma4b(a_op[step & 1], b_op[step & 1][i], acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }
    // every wave stores its own tiles of this split-K slice's partial (C layout: row = 4*kk + r, column = li)
    const int ci_valid = NARROW ? p.b_c : 16;
    float* part = p.part + (size_t)ks_id * TAPS * p.co_total * p.ci_total;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        const int t = wave + 4 * i;
        if (t < TAPS * MT && li < ci_valid) {
            const int tap = t / MT, m = t - tap * MT;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                part[((size_t)tap * p.co_total + p.co_off + co0 + m * 16 + kk * 4 + r) * p.ci_total + p.ci_off + cb0 + li] = acc[i][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient against a NARROW operand (<= 3 channels: the image at the first layer, dU at the top transposed conv), 3x3 stride 2.
// The general kernel gives every tap its own 16-column tile and uses b_c of the 16 columns (3 of 16: 81 % of its MFMAs multiply
// zeros, 69 us per launch for 0.9 GFLOP).  Here the 9 taps x b_c channels ARE the columns: column j = tap * b_c + c, 27 of the 32
// columns of two tiles for three channels, 4.5 x fewer MFMAs.  A lane's B value is an LDS read at its own (tap, channel) offset -
// LDS reads are per-lane addressed anyway.  K tiles of 8 x 16 pixels; the narrow operand is staged as scalars, one (pixel, channel)
// item per lane and pass, so that every lane carries a load (the general kernel's narrow mode used one lane in four).
// Same split-K partial layout as wgrad_mfma_kernel: [split_k][tap][co_total][ci_total].
// ------------------------------------------------------------------------------------------------
template <int MT, bool A16 = false>
__global__ __launch_bounds__(256, 2) void wgrad_narrow_kernel(WgradParams p) {
    constexpr int TH = 8, TW = 16, S = 2;
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int IWP = IW + 2;                                // padded row: tap rows land 12 banks apart
    constexpr int COP = (MT & 1) ? 16 * MT : 16 * MT + 16;     // dy pixel pitch == 16 (mod 32): conflict-free A reads
    constexpr int A_ITERS = TH * TW * 4 * MT / 256;
    constexpr int B_ITERS = (IH * IW * 3 + 255) / 256;
    constexpr int A_FLOATS = TH * TW * COP, B_FLOATS = IH * IWP * 4;
    constexpr int NT = 2 * MT;                                 // (column tile, co tile) pairs of the block
    constexpr int NACC = (NT + 3) / 4;
    __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + B_FLOATS];
    float* const a_s = smem;
    float* const b_s = smem + A_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int ks_id = blockIdx.x;
    const int co0 = blockIdx.y * 16 * MT;
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ncol = 9 * p.b_c;

    f32x4 acc[NACC];
    int b_off[NACC], col[NACC];
    const int m_w = (MT == 4) ? wave : (MT == 2) ? (wave & 1) : 0;      // all tiles of a wave share one co tile
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int t = wave + 4 * i;
        const int ct = (t < NT) ? t / MT : 0;
        const int j = ct * 16 + li;
        const int jj = j < ncol ? j : 0;
        const int tap = jj / p.b_c, c = jj - tap * p.b_c;
        const int ky = tap / 3, kx = tap - ky * 3;
        col[i] = (t < NT && j < ncol) ? tap * 4 + c : -1;             // (tap, channel) of this lane's column, -1: no such column
        b_off[i] = (ky * IWP + kx) * 4 + c;
    }

    f32x4 ra[A_ITERS];
    float rb[B_ITERS];
    unsigned ok_a = 0, ok_b = 0;
    auto load_tile = [&](int tile) {
        const int img = tile / tiles_per_img, tl = tile - img * tiles_per_img;
        const int tile_y = tl / p.tiles_x, tile_x = tl - tile_y * p.tiles_x;
        const int oy0 = tile_y * TH, ox0 = tile_x * TW;
        ok_a = 0;
        ok_b = 0;
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
            const int i = tid + 256 * it;
            const int px = i / (4 * MT), c4 = i - px * (4 * MT);
            const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
            const bool ok = oy < p.ho && ox < p.wo;
            const size_t off = ok ? (((size_t)img * p.ho + oy) * p.wo + ox) * p.a_pitch + co0 + 4 * c4 : 0;
            ra[it] = ld_act4<A16>(p.a, off);
            ok_a |= (ok ? 1u : 0u) << it;
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int i = tid + 256 * it;
            const int c = i / (IH * IW), pix = i - c * (IH * IW);       // channel-major items: a wave reads a run of one row
            const int iy = pix / IW, ix = pix - iy * IW;
            const int gy = oy0 * S - 1 + iy, gx = ox0 * S - 1 + ix;
            const bool ok = c < p.b_c && gy >= 0 && gy < p.hi && gx >= 0 && gx < p.wi;
            rb[it] = p.b_ptr[ok ? img * p.b_sn + c * p.b_sc + gy * p.b_sh + gx * p.b_sw : 0];
            ok_b |= (ok ? 1u : 0u) << it;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
            const int i = tid + 256 * it;
            const int px = i / (4 * MT), c4 = i - px * (4 * MT);
            *reinterpret_cast<f32x4*>(&a_s[px * COP + 4 * c4]) = ((ok_a >> it) & 1u) ? ra[it] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int i = tid + 256 * it;
            const int c = i / (IH * IW), pix = i - c * (IH * IW);
            const int iy = pix / IW, ix = pix - iy * IW;
            if (c < 3) b_s[(iy * IWP + ix) * 4 + c] = ((ok_b >> it) & 1u) ? rb[it] : 0.f;
        }
    };

    const int t_begin = (int)((int64_t)ks_id * p.total_tiles / p.split_k);
    const int t_end = (int)((int64_t)(ks_id + 1) * p.total_tiles / p.split_k);
    if (t_begin < t_end) {
        load_tile(t_begin);
        store_tile();
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        const bool more = tile + 1 < t_end;
        if (more) load_tile(tile + 1);
        float a_op[2], b_op[2][NACC];
        auto read_ops = [&](int step, float& av, float (&bv)[NACC]) {
            const int row = step >> 2, ks = step & 3;
            av = a_s[(row * 16 + 4 * ks + kk) * COP + m_w * 16 + li];
            const int pb = (row * S * IWP + (4 * ks + kk) * S) * 4;
#pragma unroll
            for (int i = 0; i < NACC; ++i) bv[i] = b_s[pb + b_off[i]];
        };
        read_ops(0, a_op[0], b_op[0]);
#pragma unroll
        for (int step = 0; step < 4 * TH; ++step) {
            if (step + 1 < 4 * TH) read_ops(step + 1, a_op[(step + 1) & 1], b_op[(step + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                if (NT % 4 == 0 || wave + 4 * i < NT) acc[i] = mfma4b(a_op[step & 1], b_op[step & 1][i], acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }
    float* part = p.part + (size_t)ks_id * 9 * p.co_total * p.ci_total;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
        if (col[i] >= 0) {
            const int tap = col[i] >> 2, c = col[i] & 3;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                part[((size_t)tap * p.co_total + p.co_off + co0 + m_w * 16 + kk * 4 + r) * p.ci_total + p.ci_off + c] = acc[i][r];
        }
}

// how many split-K slices a weight-gradient launch should use: enough blocks to fill the chip, never more than tiles
// pixel rows per K tile: 8 for the 3x3 stride-1 layers with 64-channel blocks (twice the MFMAs per barrier pair), else 4
static int g_wgrad_narrow = getenv("CS_WGRAD_NARROW") ? atoi(getenv("CS_WGRAD_NARROW")) : 1;   // 0: narrow operands through the general kernel
static inline bool wgrad_narrow_cols(int ksize, int stride, int b_c) { return g_wgrad_narrow && ksize == 3 && stride == 2 && b_c <= 3; }
static inline int wgrad_th(int ksize, int stride, int a_c, int b_c, int ho) {
    if (wgrad_narrow_cols(ksize, stride, b_c)) return 8;
    return (ksize == 3 && stride == 1 && a_c % 64 == 0 && b_c >= 16 && ho >= 8) ? 8 : 4;
}

int wgrad_split_k(int a_c, int b_c, int n, int ho, int wo, int ksize, int stride) {
    const int mt = (a_c % 64 == 0) ? 4 : (a_c % 32 == 0) ? 2 : 1;
    const int combos = (a_c / (16 * mt)) * (b_c >= 16 ? b_c / 16 : 1);
    const int total_tiles = n * ceil_div(wo, 16) * ceil_div(ho, wgrad_th(ksize, stride, a_c, b_c, ho));
    // blocks per launch: measured on the 32-base step (wgrad + unpack ms): 1024 -> 2.46 + 0.50, 768 -> 2.67 + 0.40, 512 -> 2.53 + 0.30,
    // 384 -> 3.17 + 0.25: fewer slices halve the partial slabs the unpack pass has to sum
    const int target_blocks = 512;
    // (four times the blocks for the narrow launches - 3-channel operand, 70 us each, four per step - was measured: the launches got
    // 5 us shorter and the unpack pass 15 us longer each)
    int k = ceil_div(target_blocks, combos);
    if (k > total_tiles) k = total_tiles;
    return k < 1 ? 1 : k;
}

int launch_wgrad(hipStream_t s, int ksize, int stride, const float* a, int a_pitch, int a_c, const ConvSrc& b, const WgradNarrow* nb,
                 float* part, int split_k, int co_total, int ci_total, int co_off, int ci_off, int n, int ho, int wo, int hi, int wi, bool a_h16) {
    CS_REQUIRE(a_c % 16 == 0 && a_c > 0, "wgrad: %d dy channels (must be a multiple of 16)", a_c);
    CS_REQUIRE((ksize == 3 && (stride == 1 || stride == 2)) || (ksize == 1 && stride == 1), "wgrad: k%d s%d unsupported", ksize, stride);
    WgradParams p{};
    p.a = a; p.a_pitch = a_pitch; p.a16 = a_h16 ? 1 : 0;
    p.b16 = (!nb && b.h16) ? 1 : 0;
    CS_REQUIRE(nb || (!b.q8 && !b.ptr2), "wgrad: fp32 or bf16 activations only");
    int b_chunks;
    if (nb) {
        CS_REQUIRE(nb->c >= 1 && nb->c <= 4, "wgrad: narrow source with %d channels", nb->c);
        p.b_mode = 1; p.b_c = nb->c; p.b_ptr = nb->ptr; p.b_sn = nb->sn; p.b_sc = nb->sc; p.b_sh = nb->sh; p.b_sw = nb->sw;
        b_chunks = 1;
    } else {
        CS_REQUIRE(b.c % 16 == 0 && b.c > 0, "wgrad: %d x channels (must be a multiple of 16)", b.c);
        p.b_mode = 0; p.b_ptr = b.ptr; p.b_tab = b.tab; p.b_slope = b.slope; p.b_pitch = b.pitch;
        p.b_tab_c = b.tab_c ? b.tab_c : b.c; p.b_tab_off = b.tab_off;
        b_chunks = b.c / 16;
    }
    p.dummy = a;
    p.part = part; p.co_total = co_total; p.ci_total = ci_total; p.co_off = co_off; p.ci_off = ci_off;
    p.n = n; p.ho = ho; p.wo = wo; p.hi = hi; p.wi = wi;
    const int th = wgrad_th(ksize, stride, a_c, nb ? nb->c : b.c, ho);
    p.tiles_x = ceil_div(wo, 16);
    p.tiles_y = ceil_div(ho, th);
    p.total_tiles = n * p.tiles_x * p.tiles_y;
    CS_REQUIRE(split_k >= 1 && split_k <= p.total_tiles, "wgrad: split_k %d out of range", split_k);
    p.split_k = split_k;
    const int mt = (a_c % 64 == 0) ? 4 : (a_c % 32 == 0) ? 2 : 1;
    dim3 grid((unsigned)split_k, (unsigned)(a_c / (16 * mt)), (unsigned)b_chunks);
    const double flops = 2.0 * n * ho * wo * (double)a_c * (nb ? nb->c : b.c) * ksize * ksize;
    ProfScope prof(K_WGRAD, s, flops, 4.0 * n * ((double)ho * wo * a_c + (double)hi * wi * (nb ? nb->c : b.c)));
#define CS_WG(S_, KS_, NB_)                                                                                   \
    switch (mt) {                                                                                            \
        case 4: hipLaunchKernelGGL((wgrad_mfma_kernel<S_, KS_, 4, NB_>), grid, dim3(256), 0, s, p); break;   \
        case 2: hipLaunchKernelGGL((wgrad_mfma_kernel<S_, KS_, 2, NB_>), grid, dim3(256), 0, s, p); break;   \
        default: hipLaunchKernelGGL((wgrad_mfma_kernel<S_, KS_, 1, NB_>), grid, dim3(256), 0, s, p); break;  \
    }
    // operands off the bf16 tape: the top transposed conv's activations (narrow kernel, A) and the 1x1 residual conv's input (B); the
    // 3x3 layers of the bf16 mode run on kernels_bwd16.hip
    CS_REQUIRE(!(a_h16 && !(nb && wgrad_narrow_cols(ksize, stride, nb->c))) && !(p.b16 && ksize != 1),
               "wgrad(fp32 kernels): bf16 activations only as the A operand of the narrow kernel / the B operand of a 1x1 layer");
    if (nb) {
        CS_REQUIRE(ksize == 3 && stride == 2, "wgrad: the narrow operand is only used by stride-2 3x3 layers");
        if (wgrad_narrow_cols(ksize, stride, nb->c)) {
            if (a_h16) {
                switch (mt) {
                    case 4: hipLaunchKernelGGL((wgrad_narrow_kernel<4, true>), grid, dim3(256), 0, s, p); break;
                    case 2: hipLaunchKernelGGL((wgrad_narrow_kernel<2, true>), grid, dim3(256), 0, s, p); break;
                    default: hipLaunchKernelGGL((wgrad_narrow_kernel<1, true>), grid, dim3(256), 0, s, p); break;
                }
            } else {
                switch (mt) {
                    case 4: hipLaunchKernelGGL((wgrad_narrow_kernel<4>), grid, dim3(256), 0, s, p); break;
                    case 2: hipLaunchKernelGGL((wgrad_narrow_kernel<2>), grid, dim3(256), 0, s, p); break;
                    default: hipLaunchKernelGGL((wgrad_narrow_kernel<1>), grid, dim3(256), 0, s, p); break;
                }
            }
        } else {
            CS_WG(2, 3, true)
        }
    } else if (ksize == 1 && p.b16) {
        switch (mt) {
            case 4: hipLaunchKernelGGL((wgrad_mfma_kernel<1, 1, 4, false, 4, true>), grid, dim3(256), 0, s, p); break;
            case 2: hipLaunchKernelGGL((wgrad_mfma_kernel<1, 1, 2, false, 4, true>), grid, dim3(256), 0, s, p); break;
            default: hipLaunchKernelGGL((wgrad_mfma_kernel<1, 1, 1, false, 4, true>), grid, dim3(256), 0, s, p); break;
        }
    } else if (ksize == 1) { CS_WG(1, 1, false) } else if (stride == 1 && th == 8) {
        hipLaunchKernelGGL((wgrad_mfma_kernel<1, 3, 4, false, 8>), grid, dim3(256), 0, s, p);
    } else if (stride == 1) { CS_WG(1, 3, false) } else { CS_WG(2, 3, false) }
#undef CS_WG
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// sum of the split_k partials [k][taps][rows_total][cols_total] (rows row_off.., cols col_off..) -> PyTorch layout
// dst[row][col][tap].  Block = EL consecutive elements x KL split-K lanes (KL = 256 / EL): every thread sums ALL taps of its
// element over its share of the split-K slices (loads coalesced over the elements), the lanes are merged through LDS and the
// block writes its EL * taps outputs as one contiguous run.  KL = 4 for short split-K, 32 for the high-resolution layers whose
// split-K is in the hundreds (round 1: 32 elements x 8 lanes, one tap per block, 22 us per launch on average).
// blockIdx.y = part: a launch may unpack two row ranges of one partial buffer into two tensors (the fused [unit0 || residual] pair:
// rows [row_off, row_off + rows) -> dst, rows [row_off + rows, row_off + 2 rows) -> dst2 with its own scales)
template <int EL, int TAPS>
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* part, int split_k, int rows_total, int cols_total, int row_off,
                                                           int col_off, int rows, int cols, float* dst, WgradUnscale us, int unscale,
                                                           float* dst2, WgradUnscale us2) {
    constexpr int KL = 256 / EL;
    __shared__ float sh[KL][EL * TAPS + 1];
    const int el = threadIdx.x % EL, kl = threadIdx.x / EL;
    const int per_tap = rows * cols;
    const int e = blockIdx.x * EL + el;
    const bool second = blockIdx.y != 0;
    if (second) {
        row_off += rows;
        dst = dst2;
        us = us2;
    }
    float acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = 0.f;
    if (e < per_tap) {
        const int row = e / cols, col = e - row * cols;
        const size_t off = ((size_t)row_off + row) * cols_total + col_off + col;
        const size_t tap_stride = (size_t)rows_total * cols_total, k_stride = (size_t)TAPS * tap_stride;
        for (int k = kl; k < split_k; k += KL) {
            const float* pk = part + (size_t)k * k_stride + off;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[t] += pk[(size_t)t * tap_stride];
        }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) sh[kl][el * TAPS + t] = acc[t];
    __syncthreads();
    const int e0 = blockIdx.x * EL;
    const int n_out = (per_tap - e0 < EL ? per_tap - e0 : EL) * TAPS;
    // partials of the split-fp16 kernels carry both operands' power-of-two scales (exact to divide out)
    const float inv = unscale ? 1.f / (f16_dyn_scale(us.a_amax) * f16_dyn_scale(us.b_amax)) : 1.f;
    for (int i = threadIdx.x; i < n_out; i += 256) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < KL; ++l) t += sh[l][i];
        dst[(size_t)e0 * TAPS + i] = t * inv;
    }
}

int launch_unpack_wgrad(hipStream_t s, const float* part, int split_k, int rows_total, int cols_total, int row_off, int col_off,
                        int rows, int cols, int taps, float* dst, const WgradUnscale* unscale, float* dst2, const WgradUnscale* unscale2) {
    const int total = rows * cols * taps;
    const WgradUnscale us = unscale ? *unscale : WgradUnscale();
    const WgradUnscale us2 = unscale2 ? *unscale2 : WgradUnscale();
    const int do_unscale = unscale ? 1 : 0;
    CS_REQUIRE(taps == 9 || taps == 1, "unpack_wgrad: %d taps", taps);
    CS_REQUIRE(!dst2 || ((unscale != nullptr) == (unscale2 != nullptr) && row_off + 2 * rows <= rows_total), "unpack_wgrad: bad second part");
    const unsigned parts = dst2 ? 2u : 1u;
    ProfScope prof(K_UNPACK, s, 0, 4.0 * total * (split_k + 1) * parts);
#define CS_UNPACK(EL_, T_)                                                                                                              \
    hipLaunchKernelGGL((unpack_wgrad_kernel<EL_, T_>), dim3((unsigned)ceil_div(rows * cols, EL_), parts), dim3(256), 0, s, part, split_k, \
                       rows_total, cols_total, row_off, col_off, rows, cols, dst, us, do_unscale, dst2, us2)
    if (split_k > 64 && rows * cols <= 1024 && taps == 9) {
        CS_UNPACK(2, 9);            // small tensor, hundreds of slices (16- and 32-channel layers at 128^2): 8 elements per block left 32
                                    // blocks summing 512 slabs (65 us); 2 elements x 128 split-K lanes put the same loads on 4x the CUs
    } else if (split_k > 16) {
        if (taps == 9) CS_UNPACK(8, 9); else CS_UNPACK(8, 1);
    } else {
        if (taps == 9) CS_UNPACK(64, 9); else CS_UNPACK(64, 1);
    }
#undef CS_UNPACK
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// All fp64 -> fp32 hand-overs of one backward pass (PReLU slope gradients: NORM_DSLOPE_SLOTS partial sums each; the last conv's
// 84 weight / bias sums) in ONE launch at the end of the pass: block b does job b, dst[i] = sum over slots of src[slot * count + i].
// (One small launch behind every norm backward was 19 launches of ~5 us on the critical chain of a 2 ms step.)
__global__ __launch_bounds__(256) void f64_jobs_kernel(F64Jobs jobs) {
    const F64Job jb = jobs.j[blockIdx.x];
    for (int i = threadIdx.x; i < jb.count; i += blockDim.x) {
        double t = 0.0;
        // eight independent loads in flight per thread (a dependent chain of 32 x 8 loads took 130 us for a 512-channel, 32-image job)
        int sl = 0;
        for (; sl + 8 <= jb.slots; sl += 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = jb.src[(size_t)(sl + k) * jb.count + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) t += v[k];
        }
        for (; sl < jb.slots; ++sl) t += jb.src[(size_t)sl * jb.count + i];
        jb.dst[i] = (float)t;
    }
}
int launch_f64_jobs(hipStream_t s, const F64Jobs& jobs) {
    if (jobs.n == 0) return CS_OK;
    hipLaunchKernelGGL(f64_jobs_kernel, dim3((unsigned)jobs.n), dim3(256), 0, s, jobs);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// top of the decoder (3 channels), VALU
// ------------------------------------------------------------------------------------------------
// last layer backward: logits = conv3x3(T(u); W) + b + T(u)
//   d_au[p][ci] = dl[p][ci] + sum_{co,k} dl[p - (k - 1)][co] * W[co][ci][k]
//   dW[co][ci][k] = sum_p dl[p][co] * T(u)[p + k - 1][ci] ;  db[co] = sum_p dl[p][co]
// Block = a 16 x 64 pixel tile.  The tile's gradient rows and ACTIVATED inputs go to LDS once, with a one-pixel halo that holds zeros
// outside the image (zero padding comes after the activation, and a neighbour outside the image contributes nothing) - so the
// 18 neighbour reads of a pixel are plain 16-byte LDS loads without bounds checks, and the norm + PReLU of an input pixel is
// evaluated once instead of once per tap.  (The first form read every neighbour from global memory - 54 scalar loads per pixel - and
// took 0.1 ms at batch 32.)  A thread keeps the 81 + 3 sums of its four pixels in registers; the block's sums meet below.
template <int C>
__global__ __launch_bounds__(256) void conv_last_bwd_kernel(const float* dl, const float* u, const float2* tab, const float* slope_p,
                                                            const float* __restrict__ w, float* d_au, float* partial, int h, int wd,
                                                            int tiles_x) {
    static_assert(C == 3, "pixel records below are float4 = 3 channels + pad");
    constexpr int TH = 16, TW = 64, IH = TH + 2, IW = TW + 2;
    __shared__ f32x4 dl_s[IH * IW], a_s[IH * IW];
    const int img = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const float slope = slope_p ? *slope_p : 0.f;
    float sc[C], sf[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float2 t = tab[img * C + c];
        sc[c] = t.x; sf[c] = t.y;
    }
    const float* dlb = dl + (size_t)img * h * wd * C;
    const float* ub = u + (size_t)img * h * wd * C;
    for (int i = threadIdx.x; i < IH * IW; i += 256) {
        const int iy = i / IW, ix = i - iy * IW;
        const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f}, a = f32x4{0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < h && gx >= 0 && gx < wd) {
            const size_t o = ((size_t)gy * wd + gx) * C;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                d[c] = dlb[o + c];
                a[c] = prelu_b(ub[o + c] * sc[c] + sf[c], slope);
            }
        }
        dl_s[i] = d;
        a_s[i] = a;
    }
    float wr[C * C * 9];                       // uniform: scalar registers
#pragma unroll
    for (int i = 0; i < C * C * 9; ++i) wr[i] = w[i];
    __syncthreads();
    float gw[C * C * 9], gb[C];
#pragma unroll
    for (int i = 0; i < C * C * 9; ++i) gw[i] = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) gb[c] = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
        const int ly = (threadIdx.x >> 6) * 4 + k4, lx = threadIdx.x & 63;       // a wave walks four rows of the tile
        const int y = y0 + ly, x = x0 + lx;
        if (y >= h || x >= wd) continue;
        const int ctr = (ly + 1) * IW + lx + 1;
        const f32x4 g = dl_s[ctr];
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            acc[c] = g[c];
            gb[c] += g[c];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int k = ky * 3 + kx;
                const f32x4 dq = dl_s[ctr - (ky - 1) * IW - (kx - 1)];      // data gradient: the neighbour this pixel's activation fed
                const f32x4 av = a_s[ctr + (ky - 1) * IW + (kx - 1)];       // weight gradient: the activation under tap k
#pragma unroll
                for (int co = 0; co < C; ++co)
#pragma unroll
                    for (int ci = 0; ci < C; ++ci) {
                        acc[ci] += dq[co] * wr[(co * C + ci) * 9 + k];
                        gw[(co * C + ci) * 9 + k] += g[co] * av[ci];
                    }
            }
#pragma unroll
        for (int c = 0; c < C; ++c) d_au[((size_t)img * h * wd + (size_t)y * wd + x) * C + c] = acc[c];
    }
    // Wave reduction as a halving butterfly: at every step a lane keeps one half of its list and hands the other half to its partner, so
    // the 84 sums cost 64 + 32 + ... + 2 = 126 cross-lane moves instead of 84 x 6 = 504.  After the six steps lane l holds the wave
    // totals of entries 2 l and 2 l + 1 of the list (padded with zeros to 128).
    constexpr int NV = C * C * 9 + C;
    static_assert(NV <= 128, "the butterfly below reduces at most 128 values per wave");
    __shared__ float red_s[4][128];
    {
        float v[128];
#pragma unroll
        for (int i = 0; i < 128; ++i) v[i] = i < C * C * 9 ? gw[i] : (i < NV ? gb[i - C * C * 9] : 0.f);
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const int m = 32 >> step, half = 64 >> step;
            const bool up = (lane & m) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const float lo = v[i], hi = v[i + half];
                const float send = up ? lo : hi, keep = up ? hi : lo;
                v[i] = keep + __shfl_xor(send, m);
            }
        }
        red_s[threadIdx.x >> 6][2 * lane] = v[0];
        red_s[threadIdx.x >> 6][2 * lane + 1] = v[1];
    }
    __syncthreads();
    if (threadIdx.x < C * C * 9 + C)
        partial[((size_t)img * gridDim.x + blockIdx.x) * (C * C * 9 + C) + threadIdx.x] =
            red_s[0][threadIdx.x] + red_s[1][threadIdx.x] + red_s[2][threadIdx.x] + red_s[3][threadIdx.x];
}

// sum of the per-block partial rows in fp64: one block per column, 256 row lanes, LDS tree (8 columns x 32 row lanes per block
// walked 2048 rows in 64 dependent trips: 21 us on the serial stretch at the top of the backward pass)
__global__ __launch_bounds__(256) void rows_sum_kernel(const float* partial, int rows, int cols, double* dst) {
    __shared__ double sh[256];
    const int c = blockIdx.x;
    double acc = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) acc += (double)partial[(size_t)r * cols + c];
    sh[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) dst[c] = sh[0];
}

// one partial row of 84 sums per 16 x 64 pixel tile
static inline int last_bwd_tiles(int h, int w) { return ceil_div(h, 16) * ceil_div(w, 64); }
size_t conv_last_bwd_partial_floats(int n, int h, int w) { return (size_t)n * last_bwd_tiles(h, w) * 84; }

int launch_conv_last_bwd(hipStream_t s, const float* dlogits, const ConvSrc& u, const float* w_torch, float* d_au, double* dw_db,
                         float* partial, int n, int h, int w) {
    CS_REQUIRE(u.c == 3 && u.pitch == 3 && u.tab, "conv_last_bwd: 3-class contract");
    ProfScope prof(K_TOP_BWD, s, 4.0 * n * h * w * 81.0, 4.0 * n * (double)h * w * 9);
    const int blocks = last_bwd_tiles(h, w);
    dim3 grid((unsigned)blocks, (unsigned)n);
    hipLaunchKernelGGL((conv_last_bwd_kernel<3>), grid, dim3(256), 0, s, dlogits, u.ptr, u.tab, u.slope, w_torch, d_au, partial, h, w,
                       ceil_div(w, 64));
    hipLaunchKernelGGL(rows_sum_kernel, dim3(84), dim3(256), 0, s, partial, n * blocks, 84, dw_db);   // [81 weights | 3 biases]
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// top transposed conv backward (data): dx[i][j][ci] = sum_{co,ky,kx} du[2i-1+ky][2j-1+kx][co] * Wt[ci][co][ky][kx]
// one thread per input pixel, channels [c_begin, c_begin + c_count) of the concatenated input go to `dst` (+= if accumulate)
template <int COUT>
__global__ __launch_bounds__(256) void convt_small_bwd_data_kernel(const float* du, const float* wt, int cin_total, int c_begin,
                                                                   int c_count, float* dst, int dst_pitch, int accumulate, int hin,
                                                                   int win) {
    extern __shared__ __attribute__((aligned(16))) float w_s[];   // [c_count][COUT*9 (+1)]
    constexpr int WROW = COUT * 9 + 1;
    for (int i = threadIdx.x; i < c_count * COUT * 9; i += blockDim.x) {
        const int ci = i / (COUT * 9), r = i - ci * COUT * 9;
        w_s[ci * WROW + r] = wt[(size_t)(c_begin + ci) * COUT * 9 + r];
    }
    __syncthreads();
    const int img = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= hin * win) return;
    const int iy = pix / win, ix = pix - iy * win;
    const int hout = 2 * hin, wout = 2 * win;
    float g[9][COUT];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int oy = 2 * iy - 1 + ky, ox = 2 * ix - 1 + kx;
            const bool ok = oy >= 0 && oy < hout && ox >= 0 && ox < wout;
#pragma unroll
            for (int co = 0; co < COUT; ++co) g[ky * 3 + kx][co] = ok ? du[(((size_t)img * hout + oy) * wout + ox) * COUT + co] : 0.f;
        }
    float* o = dst + ((size_t)img * hin * win + pix) * dst_pitch;
    if ((c_count & 3) == 0 && (dst_pitch & 3) == 0) {          // 16-byte stores: 4 channels at a time
        static_assert(WROW % 4 == 0, "a weight row is read as 16-byte LDS words");
        for (int ci = 0; ci < c_count; ci += 4) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // the row's 27 weights as seven 16-byte broadcast reads (27 scalar LDS reads per channel made the kernel LDS-issue bound)
                float wr[WROW];
                const f32x4* w4 = reinterpret_cast<const f32x4*>(&w_s[(ci + j) * WROW]);
#pragma unroll
                for (int i = 0; i < WROW / 4; ++i) {
                    const f32x4 t = w4[i];
                    wr[4 * i] = t.x; wr[4 * i + 1] = t.y; wr[4 * i + 2] = t.z; wr[4 * i + 3] = t.w;
                }
#pragma unroll
                for (int co = 0; co < COUT; ++co)
#pragma unroll
                    for (int k = 0; k < 9; ++k) acc[j] += g[k][co] * wr[co * 9 + k];
            }
            if (accumulate) acc += *reinterpret_cast<const f32x4*>(o + ci);
            *reinterpret_cast<f32x4*>(o + ci) = acc;
        }
        return;
    }
    for (int ci = 0; ci < c_count; ++ci) {
        const float* wr = &w_s[ci * WROW];
        float acc = 0.f;
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc += g[k][co] * wr[co * 9 + k];
        o[ci] = accumulate ? o[ci] + acc : acc;
    }
}

int launch_convt_small_bwd_data(hipStream_t s, const float* du, const float* w_torch, int cin_total, int c_begin, int c_count,
                                float* dst, int dst_pitch, bool accumulate, int n, int hin, int win) {
    ProfScope prof(K_TOP_BWD, s, 2.0 * n * hin * win * 27.0 * c_count, 4.0 * n * (double)hin * win * (12 + c_count));
    dim3 grid((unsigned)ceil_div(hin * win, 256), (unsigned)n);
    hipLaunchKernelGGL((convt_small_bwd_data_kernel<3>), grid, dim3(256), (size_t)c_count * 28 * sizeof(float), s, du, w_torch,
                       cin_total, c_begin, c_count, dst, dst_pitch, accumulate ? 1 : 0, hin, win);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
