// gfx950 (MI355X) fp32 convolution kernels for the 3-class U-Net hot path.
//
// Replaces the implicit torch conv2d / conv_transpose2d / instance_norm / prelu dispatches behind
// monai.networks.nets.UNet.forward (reference call sites baseline/predict.py:109 -> :45-52,
// baseline/model_training_3class.py:255).
//
// Design (DESIGN.md "Kernels"):
//   * NHWC fp32 activations; implicit GEMM with M = 16 output pixels of one image row, N = 16 output
//     channels, K = taps x input channels, on v_mfma_f32_16x16x4_f32 (exact fp32 fma chain, 64-lane wave).
//   * the K loop walks 16-channel chunks: the halo'd input patch of the chunk is staged through LDS with the
//     PRODUCER's InstanceNorm + PReLU applied on the fly (lazy normalisation: normalised tensors are never
//     written to HBM), weights arrive pre-packed in MFMA fragment order so a B fragment is one ds_read_b128.
//   * epilogue adds the bias, stores the raw conv output and accumulates per-(image, channel) sum / sum of
//     squares in fp64 (wavefront shuffle reduction + one fp64 atomic per wave and channel) for the next norm.
#include "conv_common.h"
#include "pack_jobs.h"

#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

namespace cs {

// Epilogue shared by the MFMA kernels: the wave's accumulators (C layout: channel = lane & 15, pixel = 4*(lane>>4)+r)
// go through a wave-private LDS slab so that the global stores are 16 B per lane with consecutive lanes on
// consecutive channels of one pixel (whole 64*NT-byte pixel records, full cache lines) instead of 64-byte fragments.
// `ep` points at this wave's slab of ROWS*16 pixels x (16*NT + 4) floats.  Returns nothing; stats are separate.
template <int NT, int ROWS, bool OUT16 = false>
__device__ __forceinline__ void epilogue_store(float* ep, const f32x4 (&vals)[ROWS][NT], int lane, float* out_row0, int64_t row_stride,
                                               int pitch, int n_rows_valid, int n_cols_valid, const float* add_row0 = nullptr,
                                               int64_t add_row_stride = 0, int add_pitch = 0, int64_t out_elem0 = 0) {
    constexpr int EPS = 16 * NT + 4;
    const int pl = lane & 15, g = lane >> 4;
#pragma unroll
    for (int m = 0; m < ROWS; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ep[(m * 16 + g * 4 + r) * EPS + nt * 16 + pl] = vals[m][nt][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the slab is wave-private, no barrier needed
    __builtin_amdgcn_wave_barrier();
    constexpr int F4_PER_PIX = NT * 4;
    static_assert((ROWS * 16 * F4_PER_PIX) % 64 == 0, "whole passes of the wave");
#pragma unroll
    for (int it = 0; it < ROWS * 16 * F4_PER_PIX / 64; ++it) {   // (trip count visible to the unroller: started at `lane`, hipcc kept
        const int i = lane + 64 * it;                            // the loop rolled - one LDS round trip per pass - in the first-layer kernels)
        const int pix = i / F4_PER_PIX, c4 = i - pix * F4_PER_PIX;
        const int m = pix >> 4, x = pix & 15;
        if (m < n_rows_valid && x < n_cols_valid) {
            f32x4 v = *reinterpret_cast<const f32x4*>(&ep[pix * EPS + c4 * 4]);
            if (add_row0) v += *reinterpret_cast<const f32x4*>(add_row0 + m * add_row_stride + (int64_t)x * add_pitch + c4 * 4);
            if constexpr (OUT16)      // out_row0 = tensor base, out_elem0 = element offset of the tile (bf16 elements)
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out_row0) + out_elem0 + m * row_stride + (int64_t)x * pitch + c4 * 4) = f32x4_to_bf16(v);
            else
                *reinterpret_cast<f32x4*>(out_row0 + out_elem0 + m * row_stride + (int64_t)x * pitch + c4 * 4) = v;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// generic conv: KS in {3,1}, stride S in {1,2}, NT output-channel tiles of 16 per block, MT pixel rows per wave.
// block = 256 threads = 4 waves; block tile = (4*MT) rows x 16 columns of output pixels x 16*NT channels.
// ------------------------------------------------------------------------------------------------
template <int S, int KS, int NT, int MT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvParams p) {
    constexpr int TH = 4 * MT, TW = 16;
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int CS = (S == 1) ? 24 : 20;   // floats per staged pixel: 16 channels + pad (bank-conflict-free A reads)
    constexpr int PAD = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int IN_ITERS = (IH * IW + 63) / 64;          // staged pixels per thread (4 threads per pixel)
    constexpr int W_ITERS = (TAPS * NT * 64 + 255) / 256;  // packed-weight float4s per thread
    constexpr int IN_FLOATS = (IH * IW * CS + 3) / 4 * 4;
    constexpr int EP_FLOATS = 4 * MT * 16 * (16 * NT + 4);     // epilogue slabs reuse the staging memory
    constexpr int EPST_FLOATS = EP_FLOATS + 4 * 16 * NT * 2 * 2;   // + fp64 statistics of the 4 waves
    constexpr int SM_FLOATS = (IN_FLOATS + TAPS * NT * 256) > EPST_FLOATS ? (IN_FLOATS + TAPS * NT * 256) : EPST_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];
    float* const in_s = smem;
    float* const w_s = smem + IN_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, g = lane >> 4;
    const WorkId wk = decode_work(p.tiles_x * p.tiles_y * p.n, p.ntile_total / NT);
    if (!wk.valid) return;                                   // whole block exits together (block-uniform)
    int bid = wk.spatial;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int nt0 = wk.ct * NT;
    const int q = tid & 3;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-thread staging geometry (the same for every chunk): pixel offset inside the image, or -1 for padding
    int goff[IN_ITERS];
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (tid >> 2) + 64 * it;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = oy0 * S - PAD + iy, gx = ox0 * S - PAD + ix;
        goff[it] = (pix < IH * IW && gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win) ? gy * p.win + gx : -1;
    }

    const int chunks_a = p.a.c >> 4, chunks = (p.a.c + p.b.c) >> 4;
    f32x4 rin[IN_ITERS], rw[W_ITERS];
    Lazy4 lz;

    // global -> registers for chunk `ch` (raw values; the lazy norm/act is applied when they are written to LDS)
    auto load_chunk = [&](int ch) {
        const bool from_a = ch < chunks_a;
        const SrcDev& src = from_a ? p.a : p.b;
        const int c0 = (from_a ? ch : ch - chunks_a) << 4;
        lz.load(src, img, c0 + 4 * q, p.wpack);
        const float* base = src.ptr + (size_t)img * p.hin * p.win * src.pitch + c0 + 4 * q;
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {   // unconditional (clamped) loads; padding is zeroed at store time
            const int o = goff[it] >= 0 ? goff[it] : 0;
            rin[it] = *reinterpret_cast<const f32x4*>(base + (size_t)o * src.pitch);
        }
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.wpack) + ((size_t)ch * TAPS * p.ntile_total + nt0) * 64;
#pragma unroll
        for (int it = 0; it < W_ITERS; ++it) {
            const int i = tid + 256 * it;
            if ((TAPS * NT * 64) % 256 == 0 || i < TAPS * NT * 64) {
                const int tap = i / (NT * 64), r = i - tap * (NT * 64);
                rw[it] = wsrc[(size_t)tap * p.ntile_total * 64 + r];
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int pix = (tid >> 2) + 64 * it;
            if (IH * IW % 64 == 0 || pix < IH * IW) {
                const f32x4 v = goff[it] >= 0 ? lz.apply(rin[it]) : f32x4{0.f, 0.f, 0.f, 0.f};   // padding stays 0 after the activation
                *reinterpret_cast<f32x4*>(&in_s[pix * CS + 4 * q]) = v;
            }
        }
#pragma unroll
        for (int it = 0; it < W_ITERS; ++it) {
            const int i = tid + 256 * it;
            if ((TAPS * NT * 64) % 256 == 0 || i < TAPS * NT * 64) reinterpret_cast<f32x4*>(w_s)[i] = rw[it];
        }
    };

    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int ch = 0; ch < chunks; ++ch) {
        const bool more = ch + 1 < chunks;
        if (more) load_chunk(ch + 1);   // next chunk's global loads stay in flight under this chunk's MFMAs
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
            f32x4 b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = reinterpret_cast<const f32x4*>(w_s)[(tap * NT + nt) * 64 + lane];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = wave * MT + m;
                const f32x4 a = *reinterpret_cast<const f32x4*>(&in_s[((row * S + ky) * IW + pl * S + kx) * CS + 4 * g]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[m][nt] = mfma4(a.x, b[nt].x, acc[m][nt]);
                    acc[m][nt] = mfma4(a.y, b[nt].y, acc[m][nt]);
                    acc[m][nt] = mfma4(a.z, b[nt].z, acc[m][nt]);
                    acc[m][nt] = mfma4(a.w, b[nt].w, acc[m][nt]);
                }
            }
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    // ---- epilogue: bias, statistics (fp64), then full-line stores through a wave-private LDS slab ----
    // statistics: lane sums -> 4 pixel groups of the wave (shuffles) -> 4 waves of the block (LDS) -> ONE fp64 atomic
    // pair per block and channel (per-wave atomics on the same (image, channel) words cost 3-10 % of the kernel)
    double* st_s = reinterpret_cast<double*>(smem + EP_FLOATS);      // [4 waves][16*NT][2]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (nt0 + nt) * 16 + pl;
        const float bias = p.bias ? p.bias[co] : 0.f;
        double s = 0.0, qq = 0.0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = oy0 + wave * MT + m;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = ox0 + g * 4 + r;
                const float v = acc[m][nt][r] + bias;
                acc[m][nt][r] = v;
                if (oy < p.hout && ox < p.wout) {
                    s += (double)v;
                    qq += (double)v * (double)v;
                }
            }
        }
        if (p.stats) {
            s += __shfl_xor(s, 16);
            qq += __shfl_xor(qq, 16);
            s += __shfl_xor(s, 32);
            qq += __shfl_xor(qq, 32);
            if (lane < 16) {
                st_s[(wave * 16 * NT + nt * 16 + pl) * 2] = s;
                st_s[(wave * 16 * NT + nt * 16 + pl) * 2 + 1] = qq;
            }
        }
    }
    {
        const int oy = oy0 + wave * MT;
        const OutSel o = select_out(p, nt0);
        const size_t pix0 = ((size_t)img * p.hout + oy) * p.wout + ox0;
        epilogue_store<NT, MT>(smem + wave * (MT * 16 * (16 * NT + 4)), acc, lane, o.out + pix0 * o.pitch + o.ch0,
                               (int64_t)p.wout * o.pitch, o.pitch, p.hout - oy, p.wout - ox0,
                               o.add ? o.add + pix0 * o.pitch_add + o.ch0 : nullptr, (int64_t)p.wout * o.pitch_add, o.pitch_add);
    }
    if (p.stats) {
        __syncthreads();
        if (tid < 32 * NT) {
            const int c = tid >> 1, k = tid & 1;
            const double v = st_s[(0 * 16 * NT + c) * 2 + k] + st_s[(1 * 16 * NT + c) * 2 + k] + st_s[(2 * 16 * NT + c) * 2 + k] +
                             st_s[(3 * 16 * NT + c) * 2 + k];
            atomicAdd(p.stats + ((size_t)img * p.cout + nt0 * 16 + c) * 2 + k, v);
        }
    }
}

template <int S, int KS, int NT, int MT>
static int launch_conv_t(hipStream_t s, const ConvParams& p0) {
    ConvParams p = p0;
    constexpr int TH = 4 * MT;
    p.tiles_x = ceil_div(p.wout, 16);
    p.tiles_y = ceil_div(p.hout, TH);
    dim3 grid(xcd_grid(p.tiles_x * p.tiles_y * p.n, p.ntile_total / NT));
    hipLaunchKernelGGL((conv_mfma_kernel<S, KS, NT, MT>), grid, dim3(256), 0, s, p);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// register-weights variant: wave (wn, wm) OWNS output-channel tile wn of the block's WN tiles and MT = TH / (4 / WN) of
// its rows.  The B fragments of a 16-channel chunk (TAPS float4 per lane) go straight from the packed weights (L2)
// into registers ONE TAP ROW ahead of their use - weights never touch LDS, and with WN = 4 every weight is fetched
// once per block - and the staged input is double-buffered in LDS, so a chunk costs ONE barrier.  The LDS footprint is
// the input patch only, which lets four blocks share a CU.
// block = 256 threads = 4 waves; block tile = TH rows x 16 columns x 16*WN channels.
// ------------------------------------------------------------------------------------------------
// LAZY = false: both sources are raw tensors (the fused stride-2 pairs read a materialised ResidualUnit sum; data gradients read dy) -
// no table fetch per chunk and no norm / PReLU / select arithmetic per staged element (on this part vector instructions are taken
// from the fp32 MFMA stream: 16 per float4 x 5 float4 per thread and chunk)
template <int S, int KS, int WN, int TH, int OCC, int NTW, bool LAZY = true>
__global__ __launch_bounds__(256, OCC) void conv_wreg_kernel(ConvParams p) {
    static_assert(KS >= 2, "the one-row-ahead weight ring needs at least two tap rows");
    constexpr int WM = 4 / WN, MT = TH / WM, TW = 16;
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int CS = (S == 1) ? 24 : 20;
    constexpr int PAD = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int IN_ITERS = (IH * IW + 63) / 64;
    constexpr int IN_FLOATS = (IH * IW * CS + 3) / 4 * 4;
    constexpr int NB = WN * NTW;                               // channel tiles per block (NTW per wave)
    constexpr int EPS = 16 * NB + 4;                           // floats per pixel record in the block's epilogue slab
    constexpr int EP_FLOATS = TH * 16 * EPS;
    constexpr int SM_FLOATS = (2 * IN_FLOATS) > EP_FLOATS ? (2 * IN_FLOATS) : EP_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;
    const int pl = lane & 15, g = lane >> 4;
    const WorkId wk = decode_work(p.tiles_x * p.tiles_y * p.n, p.ntile_total / NB);
    if (!wk.valid) return;
    int bid = wk.spatial;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int nt0 = wk.ct * NB;
    const int q = tid & 3;

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int goff[IN_ITERS];
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (tid >> 2) + 64 * it;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = oy0 * S - PAD + iy, gx = ox0 * S - PAD + ix;
        goff[it] = (pix < IH * IW && gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win) ? gy * p.win + gx : -1;
    }

    const int chunks_a = p.a.c >> 4, chunks = (p.a.c + p.b.c) >> 4;
    f32x4 rin[IN_ITERS];
    Lazy4 lz;
    // B fragments of one 16-channel chunk: TAPS float4 per lane, refilled ONE TAP ROW AHEAD: at the start of row ky the
    // slots of row (ky + 1) % KS - consumed one row earlier - receive the next row's (or the next chunk's first row's)
    // fragments, so every load has a full row of MFMAs to return from L2
    f32x4 wc[TAPS][NTW];
    const f32x4* const wbase = reinterpret_cast<const f32x4*>(p.wpack) + (size_t)(nt0 + wn * NTW) * 64 + lane;
    const size_t wtap = (size_t)p.ntile_total * 64;

    auto load_in = [&](int ch) {
        const bool from_a = ch < chunks_a;
        const SrcDev& src = from_a ? p.a : p.b;
        const int c0 = (from_a ? ch : ch - chunks_a) << 4;
        if constexpr (LAZY) lz.load(src, img, c0 + 4 * q, p.wpack);
        const float* base = src.ptr + (size_t)img * p.hin * p.win * src.pitch + c0 + 4 * q;
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int o = goff[it] >= 0 ? goff[it] : 0;
            rin[it] = *reinterpret_cast<const f32x4*>(base + (size_t)o * src.pitch);
        }
    };
    auto store_in = [&](float* in_s) {
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int pix = (tid >> 2) + 64 * it;
            if (IH * IW % 64 == 0 || pix < IH * IW) {
                f32x4 v = rin[it];
                if constexpr (LAZY) v = lz.apply(v);
                if (goff[it] < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(&in_s[pix * CS + 4 * q]) = v;
            }
        }
    };

    load_in(0);
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
        for (int t = 0; t < NTW; ++t) wc[kx][t] = wbase[(size_t)kx * wtap + t * 64];
    store_in(smem);
    __syncthreads();
    const int last_row = chunks * KS - 1;
    for (int ch = 0; ch < chunks; ++ch) {
        const bool more = ch + 1 < chunks;
        // next chunk's input: UNCONDITIONAL loads (the last trip re-reads its own chunk from L2).  A load group skipped on
        // one path would make hipcc size the vmcnt waits of the weight ring for that path and stall on these loads.
        load_in(more ? ch + 1 : ch);
        __builtin_amdgcn_sched_barrier(0);
        const float* in_s = smem + (ch & 1) * IN_FLOATS;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            {
                const int row_next = ch * KS + ky + 1 < last_row ? ch * KS + ky + 1 : last_row;   // clamped: unconditional loads
                const f32x4* wnext = wbase + (size_t)row_next * KS * wtap;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int t = 0; t < NTW; ++t) wc[((ky + 1) % KS) * KS + kx][t] = wnext[(size_t)kx * wtap + t * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            // (hipcc reads each A fragment and waits lgkmcnt(0) directly in front of the 4 * NTW MFMAs it feeds.  Reading it one step
            // ahead by hand - as the weight-gradient kernel does - was measured neutral here: stride-2 group 7.80 -> 7.73 ms on the
            // slide, 0.617 -> 0.642 ms in the training step; the three blocks per CU already cover each other's LDS round trips.)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                f32x4 b[NTW];
#pragma unroll
                for (int t = 0; t < NTW; ++t) b[t] = wc[ky * KS + kx][t];
                f32x4 a[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    a[m] = *reinterpret_cast<const f32x4*>(&in_s[(((wm * MT + m) * S + ky) * IW + pl * S + kx) * CS + 4 * g]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NTW; ++t) {
                        acc[m][t] = mfma4(a[m].x, b[t].x, acc[m][t]);
                        acc[m][t] = mfma4(a[m].y, b[t].y, acc[m][t]);
                        acc[m][t] = mfma4(a[m].z, b[t].z, acc[m][t]);
                        acc[m][t] = mfma4(a[m].w, b[t].w, acc[m][t]);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) asm volatile("" ::"v"(rin[it]));   // keeps the loads above the MFMA section
        if (more) store_in(smem + ((ch + 1) & 1) * IN_FLOATS);
        __syncthreads();
    }

    // ---- epilogue: bias + fp64 statistics of the wave's channels over its rows, block slab, full-line stores ----
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int ct = wn * NTW + t, co = (nt0 + ct) * 16 + pl;
        const float bias = p.bias ? p.bias[co] : 0.f;
        double ssum = 0.0, qsum = 0.0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = wm * MT + m, oy = oy0 + row;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = ox0 + g * 4 + r;
                const float v = acc[m][t][r] + bias;
                smem[(row * 16 + g * 4 + r) * EPS + ct * 16 + pl] = v;
                if (oy < p.hout && ox < p.wout) {
                    ssum += (double)v;
                    qsum += (double)v * (double)v;
                }
            }
        }
        if (p.stats) stats_commit(ssum, qsum, p.stats + ((size_t)img * p.cout + co) * 2, lane);
    }
    __syncthreads();
    {
        constexpr int F4 = NB * 4, PPP = 256 / F4;             // float4s per pixel record, pixels per pass
        const int c4 = tid % F4;
        const OutSel o = select_out(p, nt0 + c4 / 4);          // a block may straddle the (out, out2) split: per channel tile
        const int cw = (c4 & 3) * 4;                           // channel offset inside the 16-channel tile
#pragma unroll
        for (int it = 0; it < TH * 16 / PPP; ++it) {
            const int pix = tid / F4 + PPP * it;
            const int m = pix >> 4, x = pix & 15;
            const int oy = oy0 + m, ox = ox0 + x;
            if (oy < p.hout && ox < p.wout) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&smem[pix * EPS + c4 * 4]);
                const size_t gp = ((size_t)img * p.hout + oy) * p.wout + ox;
                if (o.add) v += *reinterpret_cast<const f32x4*>(o.add + gp * o.pitch_add + o.ch0 + cw);
                *reinterpret_cast<f32x4*>(o.out + gp * o.pitch + o.ch0 + cw) = v;
            }
        }
    }
}

template <int S, int KS, int WN, int TH, int OCC, int NTW>
static int launch_conv_wreg(hipStream_t s, const ConvParams& p0) {
    ConvParams p = p0;
    p.tiles_x = ceil_div(p.wout, 16);
    p.tiles_y = ceil_div(p.hout, TH);
    dim3 grid(xcd_grid(p.tiles_x * p.tiles_y * p.n, p.ntile_total / (WN * NTW)));
    if (S == 2 && !p.a.tab && (p.b.c == 0 || !p.b.tab))          // raw sources (the stride-2 layers' usual case): no lazy-norm arithmetic
        hipLaunchKernelGGL((conv_wreg_kernel<S, KS, WN, TH, OCC, NTW, false>), grid, dim3(256), 0, s, p);
    else
    hipLaunchKernelGGL((conv_wreg_kernel<S, KS, WN, TH, OCC, NTW>), grid, dim3(256), 0, s, p);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

static SrcDev to_dev(const ConvSrc& s) { return SrcDev{s.ptr, s.tab, s.slope, s.c, s.pitch, s.tab_c ? s.tab_c : s.c, s.tab_off, s.ptr2, s.tab2, s.slope2}; }

static int apply_extra(ConvParams& p, const ConvOutExtra* ex, int cout, int* nt) {
    if (!ex) return CS_OK;
    p.addend = ex->addend; p.pitch_add = ex->pitch_add;
    p.addend2 = ex->addend2; p.pitch_add2 = ex->pitch_add2;
    p.out2 = ex->out2; p.pitch_out2 = ex->pitch_out2;
    if (ex->out2) {
        CS_REQUIRE(ex->split % 16 == 0 && ex->split > 0 && ex->split < cout, "conv: bad output split %d of %d", ex->split, cout);
        p.split_nt = ex->split / 16;
        while (*nt > 1 && ((ex->split % (16 * *nt)) != 0 || ((cout - ex->split) % (16 * *nt)) != 0)) *nt /= 2;
    }
    return CS_OK;
}

int launch_conv_mfma(hipStream_t s, int ksize, int stride, const ConvSrc& a, const ConvSrc& b, const float* wpack,
                     const float* bias, float* out, int pitch_out, int cout, double* stats, int n, int hin, int win,
                     const ConvOutExtra* ex) {
    CS_REQUIRE(ksize == 3 || ksize == 1, "conv: ksize %d unsupported", ksize);
    CS_REQUIRE(stride == 1 || stride == 2, "conv: stride %d unsupported", stride);
    CS_REQUIRE(!(ksize == 1 && stride != 1), "conv: 1x1 stride 2 unsupported");
    CS_REQUIRE(a.c % 16 == 0 && b.c % 16 == 0 && a.c > 0, "conv: input channels (%d,%d) must be multiples of 16", a.c, b.c);
    CS_REQUIRE(!a.ptr2 && !b.ptr2, "conv: sum sources are only understood by the transposed convolution");
    CS_REQUIRE(cout % 16 == 0, "conv: cout %d must be a multiple of 16", cout);
    CS_REQUIRE((a.pitch % 4) == 0 && (b.c == 0 || (b.pitch % 4) == 0), "conv: pixel pitch must be a multiple of 4 floats");
    ConvParams p{};
    p.a = to_dev(a);
    p.b = to_dev(b);
    p.wpack = wpack;
    p.bias = bias;
    p.out = out;
    p.stats = stats;
    p.pitch_out = pitch_out;
    p.cout = cout;
    p.ntile_total = cout / 16;
    p.n = n;
    p.hin = hin;
    p.win = win;
    const int pad = ksize / 2;
    p.hout = (hin + 2 * pad - ksize) / stride + 1;
    p.wout = (win + 2 * pad - ksize) / stride + 1;
    int nt = (cout % 64 == 0) ? 4 : (cout % 32 == 0) ? 2 : 1;
    if (int rc = apply_extra(p, ex, cout, &nt)) return rc;
    const double flops = 2.0 * n * p.hout * p.wout * (double)cout * (a.c + b.c) * ksize * ksize;
    const double bytes = 4.0 * n * ((double)hin * win * (a.c + b.c) + (double)p.hout * p.wout * cout);
    ProfScope prof(ksize == 1 ? K_CONV1 : (stride == 1 ? K_CONV3_S1 : K_CONV3_S2), s, flops, bytes);
#define CS_DISPATCH_NT(S_, KS_, MT_)                                           \
    switch (nt) {                                                              \
        case 4: return launch_conv_t<S_, KS_, 4, MT_>(s, p);                   \
        case 2: return launch_conv_t<S_, KS_, 2, MT_>(s, p);                   \
        default: return launch_conv_t<S_, KS_, 1, MT_>(s, p);                  \
    }
    auto blocks_of = [&](int th, int ch_tiles) {
        return (long long)xcd_grid(ceil_div(p.wout, 16) * ceil_div(p.hout, th) * n, p.ntile_total / ch_tiles);
    };
    // 64-channel blocks of the 3x3 layers take the register-weights kernel (measured: s1 31.9 -> 30.3 ms, s2 9.8 -> 8.3 ms per
    // 4096^2 slide; the 32-channel and 1x1 layers were not faster with it and keep the LDS-weights kernel)
    if (nt == 4 && ksize == 3 && stride == 1) return launch_conv_wreg<1, 3, 4, 8, 3, 1>(s, p);
    if (nt == 4 && ksize == 3 && stride == 2) {
        // 128-channel blocks (two channel tiles per wave: every A fragment feeds 8 MFMAs, 288 MFMAs per barrier) when the
        // channel count and the (out, out2) split allow it - and the grid still covers the chip (small training launches do not)
        const bool wide = cout % 128 == 0 && blocks_of(4, 8) >= SMALL_GRID_BLOCKS;
        return wide ? launch_conv_wreg<2, 3, 4, 4, 3, 2>(s, p) : launch_conv_wreg<2, 3, 4, 4, 3, 1>(s, p);
    }
    // LDS-weights kernels: narrower channel blocks for launches that would not cover the chip once (same kernel, same order)
    {
        const int th = ksize == 3 && stride == 2 ? 4 : 8;
        while (nt > 1 && blocks_of(th, nt) < SMALL_GRID_BLOCKS) nt /= 2;
    }
    // (a register-weights form of the 32-channel stride-1 layers with a 16-row tile was measured in round 2: 30.9 vs 30.6 ms per
    // slide for the stride-1 group - not kept)
    if (ksize == 1) { CS_DISPATCH_NT(1, 1, 2) }
    if (stride == 1) { CS_DISPATCH_NT(1, 3, 2) }
    CS_DISPATCH_NT(2, 3, 1)
#undef CS_DISPATCH_NT
}

// ------------------------------------------------------------------------------------------------
// first layer: CIN (3) input channels, 3x3 pad 1, K = 9*CIN flattened as k = tap*CIN + c (padded to 4).
// Reads its windows straight from the image (sliding-window gather fused, row T step 5 "cat slices").
// ------------------------------------------------------------------------------------------------
struct FirstParams {
    const float* image;
    int64_t sn, sc, sh, sw;
    const int2* origins;
    int img_h, img_w, roi_h, roi_w;
    const float* wpack;
    const float* bias;
    float* out;
    double* stats;
    int pitch_out, cout, ntile_total, n, hout, wout, tiles_x, tiles_y;
    int stats_c;       // statistics only for channels [0, stats_c): the residual half of a fused [unit0 || residual] launch feeds no norm
    int nt_pack, nt_off;   // the packed weights hold nt_pack channel tiles per K step; this launch computes tiles [nt_off, nt_off + NT)
    SrcDev add;        // ADD launches: out = conv + bias + act(norm(add)) (the unit's last conv output with its lazy norm table)
};

// epilogue_store with the ResidualUnit's sum folded in: every stored quad gets act(norm(y)) of the same pixel and channels added
// (y raw, its InstanceNorm / BatchNorm table and PReLU slope as in the consumers' lazy staging).  A lane's channel quad is the same
// in every pass (64 % (4 NT) == 0): one table fetch per lane.
template <int NT, int ROWS>
__device__ __forceinline__ void epilogue_store_lazy_add(float* ep, const f32x4 (&vals)[ROWS][NT], int lane, float* out, int64_t elem0,
                                                        int64_t row_stride, int pitch, int n_rows_valid, int n_cols_valid,
                                                        const SrcDev& y, int img, int64_t y_elem0, int64_t y_row_stride, const float* dummy) {
    constexpr int EPS = 16 * NT + 4;
    constexpr int F4_PER_PIX = NT * 4;
    static_assert(64 % F4_PER_PIX == 0 && (ROWS * 16 * F4_PER_PIX) % 64 == 0, "a lane keeps its channel quad; whole passes of the wave");
    const int pl = lane & 15, g = lane >> 4;
    const int c4 = lane % F4_PER_PIX;
    Lazy4 lz;
    lz.load(y, img, c4 * 4, dummy);
    f32x4 yv[ROWS * 16 * F4_PER_PIX / 64];
#pragma unroll
    for (int it = 0; it < ROWS * 16 * F4_PER_PIX / 64; ++it) {       // requested before the slab round trip
        const int pix = (lane + 64 * it) / F4_PER_PIX;
        const int m = pix >> 4, x = pix & 15;
        const bool ok = m < n_rows_valid && x < n_cols_valid;
        yv[it] = *reinterpret_cast<const f32x4*>(y.ptr + (ok ? y_elem0 + m * y_row_stride + (int64_t)x * y.pitch + c4 * 4 : 0));
    }
#pragma unroll
    for (int m = 0; m < ROWS; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ep[(m * 16 + g * 4 + r) * EPS + nt * 16 + pl] = vals[m][nt][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the slab is wave-private
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < ROWS * 16 * F4_PER_PIX / 64; ++it) {
        const int pix = (lane + 64 * it) / F4_PER_PIX;
        const int m = pix >> 4, x = pix & 15;
        if (m < n_rows_valid && x < n_cols_valid) {
            const f32x4 v = lz.apply(yv[it]) + *reinterpret_cast<const f32x4*>(&ep[pix * EPS + c4 * 4]);     // (a + b as norm_act_add forms it)
            *reinterpret_cast<f32x4*>(out + elem0 + m * row_stride + (int64_t)x * pitch + c4 * 4) = v;
        }
    }
}

template <int S, int CIN, int NT, int MT, bool OUT16, bool ADD = false>
__global__ __launch_bounds__(256) void conv_first_kernel(FirstParams p) {
    static_assert(!(ADD && OUT16), "the fused sum is an fp32 epilogue");
    constexpr int TH = 4 * MT, TW = 16;
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int KSTEPS = (9 * CIN + 3) / 4;
    constexpr int IN_FLOATS = (IH * IW * CIN + 4 + 3) / 4 * 4;
    constexpr int EP_FLOATS = 4 * MT * 16 * (16 * NT + 4) + 4 * 16 * NT * 2 * 2;     // 4 wave slabs + fp64 statistics
    constexpr int SM_FLOATS = (IN_FLOATS + KSTEPS * NT * 64) > EP_FLOATS ? (IN_FLOATS + KSTEPS * NT * 64) : EP_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];
    float* const in_s = smem;
    float* const w_s = smem + IN_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, g = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;

    int y0 = 0, x0 = 0;
    const float* image = p.image;
    if (p.origins) {
        const int2 o = p.origins[img];
        y0 = o.x;
        x0 = o.y;
    } else {
        image += (size_t)img * p.sn;
    }
    // the block's weight fragments (7 KB from L2) are requested FIRST and parked in registers: as a plain copy loop behind the
    // patch stores hipcc kept the loop rolled - load, vmcnt(0), LDS store, seven L2 round trips in a row, about half of a block's life
    constexpr int W4 = KSTEPS * NT * 16, W_ITERS = (W4 + 255) / 256;
    f32x4 rwt[W_ITERS];
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it) {
        const int i = tid + 256 * it;
        const int j = (W4 % 256 == 0 || i < W4) ? i : 0;
        // (ks, nt, quad) of the block's slice inside packed[ks][nt_pack][16 quads]; the whole pack when nt_pack == NT
        rwt[it] = reinterpret_cast<const f32x4*>(p.wpack)[((j / (NT * 16)) * p.nt_pack + p.nt_off) * 16 + j % (NT * 16)];
    }
    // unconditional (clamped) loads, fully unrolled: a load under a branch makes hipcc wait for it before issuing the next
    // (a row-per-wave gather with the row arithmetic on the scalar unit - 13 half-empty loads instead of 7 full ones, ~50 fewer VALU
    // instructions per load - was measured: 1.34 vs 1.30 ms per slide, no gain)
    constexpr int ST_ITERS = (IH * IW * CIN + 255) / 256;
    float rv[ST_ITERS];
#pragma unroll
    for (int it = 0; it < ST_ITERS; ++it) {
        const int i = tid + 256 * it;
        const int c = i % CIN, pix = i / CIN;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int wy = oy0 * S - 1 + iy, wx = ox0 * S - 1 + ix;       // window-local coordinates
        const int gy = y0 + wy, gx = x0 + wx;                         // image coordinates
        const bool ok = i < IH * IW * CIN && wy >= 0 && wy < p.roi_h && wx >= 0 && wx < p.roi_w && gy >= 0 && gy < p.img_h &&
                        gx >= 0 && gx < p.img_w;
        const float v = image[ok ? c * p.sc + gy * p.sh + gx * p.sw : 0];
        rv[it] = ok ? v : 0.f;
    }
#pragma unroll
    for (int it = 0; it < ST_ITERS; ++it) {
        const int i = tid + 256 * it;
        if (i < IH * IW * CIN) in_s[i] = rv[it];
    }
    if (tid < 4) in_s[IH * IW * CIN + tid] = 0.f;
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it) {
        const int i = tid + 256 * it;
        if (W4 % 256 == 0 || i < W4) reinterpret_cast<f32x4*>(w_s)[i] = rwt[it];
    }
    __syncthreads();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const int k = 4 * ks + g;
        const int tap = (k < 9 * CIN) ? k / CIN : 0, c = (k < 9 * CIN) ? k - (k / CIN) * CIN : 0;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int koff = (ky * IW + kx) * CIN + c;
        float b[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = w_s[(ks * NT + nt) * 64 + lane];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = wave * MT + m;
            const float a = in_s[((row * S) * IW + pl * S) * CIN + koff];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma4(a, b[nt], acc[m][nt]);
        }
    }

    // statistics: lane sums -> the wave's 4 pixel groups (shuffles) -> the block's 4 waves (LDS) -> ONE fp64 atomic pair per
    // block and channel (per-wave atomics on the same (window, channel) words serialise in L2)
    __syncthreads();   // every wave has finished reading the staged tile before statistics / slabs overwrite it
    constexpr int EPW = MT * 16 * (16 * NT + 4);
    double* st_s = reinterpret_cast<double*>(smem + 4 * EPW);          // [4 waves][16*NT][2], behind the epilogue slabs
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 16 + pl;
        const float bias = p.bias ? p.bias[p.nt_off * 16 + co] : 0.f;
        double s = 0.0, qq = 0.0;
        const bool want = !ADD && p.stats && nt * 16 < p.stats_c;        // block-uniform: the fp64 work is skipped for channels nobody normalises
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = oy0 + wave * MT + m;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = ox0 + g * 4 + r;
                const float v = acc[m][nt][r] + bias;
                acc[m][nt][r] = v;
                if (want && oy < p.hout && ox < p.wout) {
                    s += (double)v;
                    qq += (double)v * (double)v;
                }
            }
        }
        if (want) {
            s += __shfl_xor(s, 16);
            qq += __shfl_xor(qq, 16);
            s += __shfl_xor(s, 32);
            qq += __shfl_xor(qq, 32);
            if (lane < 16) {
                st_s[(wave * 16 * NT + nt * 16 + pl) * 2] = s;
                st_s[(wave * 16 * NT + nt * 16 + pl) * 2 + 1] = qq;
            }
        }
    }
    {
        const int oy = oy0 + wave * MT;
        const int64_t elem0 = (((int64_t)img * p.hout + oy) * p.wout + ox0) * p.pitch_out;
        if constexpr (ADD)
            epilogue_store_lazy_add<NT, MT>(smem + wave * EPW, acc, lane, p.out, elem0, (int64_t)p.wout * p.pitch_out, p.pitch_out, p.hout - oy,
                                            p.wout - ox0, p.add, img, (((int64_t)img * p.hout + oy) * p.wout + ox0) * p.add.pitch,
                                            (int64_t)p.wout * p.add.pitch, p.wpack);
        else
            epilogue_store<NT, MT, OUT16>(smem + wave * EPW, acc, lane, p.out, (int64_t)p.wout * p.pitch_out, p.pitch_out,
                                          p.hout - oy, p.wout - ox0, nullptr, 0, 0, elem0);
    }
    if (!ADD && p.stats) {
        __syncthreads();
        if (tid < 32 * NT && (tid >> 1) < p.stats_c) {
            const int c = tid >> 1, k = tid & 1;
            const double v = st_s[(0 * 16 * NT + c) * 2 + k] + st_s[(1 * 16 * NT + c) * 2 + k] + st_s[(2 * 16 * NT + c) * 2 + k] +
                             st_s[(3 * 16 * NT + c) * 2 + k];
            atomicAdd(p.stats + ((size_t)img * p.cout + c) * 2 + k, v);
        }
    }
}

int launch_conv_first(hipStream_t s, int stride, const WindowSrc& src, int cin, const float* wpack, const float* bias,
                      float* out, int pitch_out, int cout, double* stats, int n, bool out16, int stats_c, int cout_pack, int cout_off,
                      const ConvSrc* add) {
    CS_REQUIRE(cin == 3, "first conv: in_channels %d unsupported (the reference always feeds 3)", cin);
    CS_REQUIRE(stride == 2, "first conv: stride %d unsupported", stride);
    CS_REQUIRE(cout == 16 || cout == 32 || cout == 64 || cout == 128, "first conv: fused cout %d unsupported", cout);
    if (cout_pack <= 0) cout_pack = cout;
    CS_REQUIRE(cout_pack % 16 == 0 && cout_off % 16 == 0 && cout_off >= 0 && cout_off + cout <= cout_pack,
               "first conv: channels [%d, %d) of a %d-channel pack", cout_off, cout_off + cout, cout_pack);
    CS_REQUIRE(!add || (!out16 && !stats && add->ptr && !add->h16 && !add->q8 && !add->ptr2 && add->c == cout && add->pitch % 4 == 0 &&
                        pitch_out % 4 == 0),
               "first conv: the fused residual sum takes a plain fp32 tensor of %d channels and writes fp32", cout);
    FirstParams p{};
    p.image = src.image;
    p.sn = src.sn; p.sc = src.sc; p.sh = src.sh; p.sw = src.sw;
    p.origins = src.origins;
    p.img_h = src.img_h; p.img_w = src.img_w; p.roi_h = src.roi_h; p.roi_w = src.roi_w;
    p.wpack = wpack; p.bias = bias; p.out = out; p.stats = stats;
    p.pitch_out = pitch_out; p.cout = cout; p.ntile_total = cout / 16; p.n = n;
    p.stats_c = stats_c > 0 ? stats_c : cout;
    p.nt_pack = cout_pack / 16;
    p.nt_off = cout_off / 16;
    p.cout = cout_pack;           // (the kernel's only use: the stride of the statistics rows, laid out for the whole pack)
    if (add) p.add = SrcDev{add->ptr, add->tab, add->slope, add->c, add->pitch, add->tab_c ? add->tab_c : add->c, add->tab_off, nullptr, nullptr, nullptr};
    p.hout = (src.roi_h + 2 - 3) / 2 + 1;
    p.wout = (src.roi_w + 2 - 3) / 2 + 1;
    constexpr int MT = 2;
    p.tiles_x = ceil_div(p.wout, 16);
    p.tiles_y = ceil_div(p.hout, 4 * MT);
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * n));
    const double flops = 2.0 * n * p.hout * p.wout * (double)cout * cin * 9;
    const double bytes = n * (4.0 * src.roi_h * src.roi_w * cin + (out16 ? 2.0 : 4.0) * p.hout * p.wout * cout * (add ? 2.0 : 1.0));
    ProfScope prof(K_CONV_FIRST, s, flops, bytes);
#define CS_FIRST(NT_)                                                                                          \
    if (add) hipLaunchKernelGGL((conv_first_kernel<2, 3, NT_, MT, false, true>), grid, dim3(256), 0, s, p);    \
    else if (out16) hipLaunchKernelGGL((conv_first_kernel<2, 3, NT_, MT, true>), grid, dim3(256), 0, s, p);    \
    else hipLaunchKernelGGL((conv_first_kernel<2, 3, NT_, MT, false>), grid, dim3(256), 0, s, p);
    switch (cout / 16) {
        case 1: CS_FIRST(1) break;
        case 2: CS_FIRST(2) break;
        case 4: CS_FIRST(4) break;
        default: CS_FIRST(8) break;
    }
#undef CS_FIRST
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// first layer on the bf16 matrix pipe (bf16 / fp8 datapaths): K = 27 fits ONE v_mfma_f32_16x16x32_bf16 step.
//   phase 1  the 17 x 33 x 3 window patch goes image -> bf16 -> LDS (the tiler's gather, as in conv_first_kernel)
//   phase 2  im2col in LDS: a_s[128 pixels][32 k] bf16 (k = tap * 3 + c), 80-byte rows: A fragments are conflict-free b128 reads
//   phase 3  MT x NT MFMAs per wave, weights (NT fragments of 16 bytes per lane) straight from L2 into registers
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8f __attribute__((ext_vector_type(8)));
template <int NT, int MT>
__global__ __launch_bounds__(256) void conv_first16_kernel(FirstParams p) {
    constexpr int S = 2, CIN = 3, TH = 4 * MT, TW = 16;
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int PATCH = IH * IW * CIN;                     // bf16 elements
    constexpr int PA = 40;                                   // a_s row pitch in bf16 elements (80 bytes)
    constexpr int IN_BYTES = (PATCH * 2 + 15) / 16 * 16, A_BYTES = TH * TW * PA * 2;
    constexpr int EPW = MT * 16 * (16 * NT + 4);             // floats per wave slab
    constexpr int EP_BYTES = (4 * EPW + 4 * 16 * NT * 2 * 2) * 4;
    constexpr int SM_BYTES = (IN_BYTES + A_BYTES) > EP_BYTES ? (IN_BYTES + A_BYTES) : EP_BYTES;
    __shared__ __attribute__((aligned(16))) char smem_c[SM_BYTES];
    unsigned short* const in_s = reinterpret_cast<unsigned short*>(smem_c);
    unsigned short* const a_s = reinterpret_cast<unsigned short*>(smem_c + IN_BYTES);
    float* const smem = reinterpret_cast<float*>(smem_c);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, g = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;

    int y0 = 0, x0 = 0;
    const float* image = p.image;
    if (p.origins) {
        const int2 o = p.origins[img];
        y0 = o.x;
        x0 = o.y;
    } else {
        image += (size_t)img * p.sn;
    }
    // weights: NT fragments, issued first so that they are in flight under the patch gather
    bf16x8f b[NT];
    const bf16x8f* wsrc = reinterpret_cast<const bf16x8f*>(p.wpack) + lane;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = wsrc[nt * 64];
    constexpr int ST_ITERS = (PATCH + 255) / 256;
    float rv[ST_ITERS];
#pragma unroll
    for (int it = 0; it < ST_ITERS; ++it) {                  // unconditional (clamped) loads, all in flight together
        const int i = tid + 256 * it;
        const int c = i % CIN, pix = i / CIN;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int wy = oy0 * S - 1 + iy, wx = ox0 * S - 1 + ix;
        const int gy = y0 + wy, gx = x0 + wx;
        const bool ok = i < PATCH && wy >= 0 && wy < p.roi_h && wx >= 0 && wx < p.roi_w && gy >= 0 && gy < p.img_h && gx >= 0 && gx < p.img_w;
        const float v = image[ok ? c * p.sc + gy * p.sh + gx * p.sw : 0];
        rv[it] = ok ? v : 0.f;
    }
#pragma unroll
    for (int it = 0; it < ST_ITERS; ++it) {
        const int i = tid + 256 * it;
        if (i < PATCH) in_s[i] = (unsigned short)f32_to_bf16_bits(rv[it]);
    }
    __syncthreads();
    // im2col: thread (pixel, half) writes 16 of the pixel's 32 k values; the half is resolved at compile time (two code paths) so
    // that every (tap, channel) offset of the patch is an immediate
    auto build = [&](int px, auto half) {
        constexpr int H = decltype(half)::value;
        const int row = px >> 4, col = px & 15;
        const unsigned short* src = in_s + ((row * S) * IW + col * S) * CIN;
        unsigned w[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            unsigned pair = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                constexpr int dummy = 0;
                const int k = 16 * H + 2 * jj + e;
                const int tap = k / CIN, c = k - tap * CIN, ky = tap / 3, kx = tap - ky * 3;
                const unsigned v = k < 9 * CIN ? src[(ky * IW + kx) * CIN + c] : 0u;
                pair |= v << (16 * e);
                (void)dummy;
            }
            w[jj] = pair;
        }
        uint4* dst = reinterpret_cast<uint4*>(a_s + px * PA + 16 * H);
        dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    };
    for (int item = tid; item < TH * TW * 2; item += 256) {
        if (item & 1) build(item >> 1, std::integral_constant<int, 1>());
        else build(item >> 1, std::integral_constant<int, 0>());
    }
    __syncthreads();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const bf16x8f a = *reinterpret_cast<const bf16x8f*>(a_s + ((wave * MT + m) * 16 + pl) * PA + 8 * g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[nt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    __syncthreads();   // every wave has read its A fragments before statistics / slabs overwrite the staging memory
    double* st_s = reinterpret_cast<double*>(smem + 4 * EPW);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 16 + pl;
        const float bias = p.bias ? p.bias[co] : 0.f;
        double s = 0.0, qq = 0.0;
        const bool want = p.stats && nt * 16 < p.stats_c;        // block-uniform: the fp64 work is skipped for channels nobody normalises
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = oy0 + wave * MT + m;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = ox0 + g * 4 + r;
                const float v = acc[m][nt][r] + bias;
                acc[m][nt][r] = v;
                if (want && oy < p.hout && ox < p.wout) {
                    s += (double)v;
                    qq += (double)v * (double)v;
                }
            }
        }
        if (want) {
            s += __shfl_xor(s, 16);
            qq += __shfl_xor(qq, 16);
            s += __shfl_xor(s, 32);
            qq += __shfl_xor(qq, 32);
            if (lane < 16) {
                st_s[(wave * 16 * NT + nt * 16 + pl) * 2] = s;
                st_s[(wave * 16 * NT + nt * 16 + pl) * 2 + 1] = qq;
            }
        }
    }
    {
        const int oy = oy0 + wave * MT;
        const int64_t elem0 = (((int64_t)img * p.hout + oy) * p.wout + ox0) * p.pitch_out;
        epilogue_store<NT, MT, true>(smem + wave * EPW, acc, lane, p.out, (int64_t)p.wout * p.pitch_out, p.pitch_out,
                                     p.hout - oy, p.wout - ox0, nullptr, 0, 0, elem0);
    }
    if (p.stats) {
        __syncthreads();
        if (tid < 32 * NT && (tid >> 1) < p.stats_c) {
            const int c = tid >> 1, k = tid & 1;
            const double v = st_s[(0 * 16 * NT + c) * 2 + k] + st_s[(1 * 16 * NT + c) * 2 + k] + st_s[(2 * 16 * NT + c) * 2 + k] +
                             st_s[(3 * 16 * NT + c) * 2 + k];
            atomicAdd(p.stats + ((size_t)img * p.cout + c) * 2 + k, v);
        }
    }
}

int launch_conv_first16(hipStream_t s, const WindowSrc& src, int cin, const unsigned short* wpack16, const float* bias, float* out,
                        int pitch_out, int cout, double* stats, int n, int stats_c) {
    CS_REQUIRE(cin == 3, "first conv: in_channels %d unsupported (the reference always feeds 3)", cin);
    CS_REQUIRE(cout == 16 || cout == 32 || cout == 64 || cout == 128, "first conv: fused cout %d unsupported", cout);
    FirstParams p{};
    p.image = src.image;
    p.sn = src.sn; p.sc = src.sc; p.sh = src.sh; p.sw = src.sw;
    p.origins = src.origins;
    p.img_h = src.img_h; p.img_w = src.img_w; p.roi_h = src.roi_h; p.roi_w = src.roi_w;
    p.wpack = reinterpret_cast<const float*>(wpack16); p.bias = bias; p.out = out; p.stats = stats;
    p.pitch_out = pitch_out; p.cout = cout; p.ntile_total = cout / 16; p.n = n;
    p.stats_c = stats_c > 0 ? stats_c : cout;
    p.hout = (src.roi_h + 2 - 3) / 2 + 1;
    p.wout = (src.roi_w + 2 - 3) / 2 + 1;
    constexpr int MT = 2;
    p.tiles_x = ceil_div(p.wout, 16);
    p.tiles_y = ceil_div(p.hout, 4 * MT);
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * n));
    const double flops = 2.0 * n * p.hout * p.wout * (double)cout * cin * 9;
    const double bytes = n * (4.0 * src.roi_h * src.roi_w * cin + 2.0 * p.hout * p.wout * cout);
    ProfScope prof(K_CONV_FIRST, s, flops, bytes);
    switch (cout / 16) {
        case 1: hipLaunchKernelGGL((conv_first16_kernel<1, MT>), grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((conv_first16_kernel<2, MT>), grid, dim3(256), 0, s, p); break;
        case 4: hipLaunchKernelGGL((conv_first16_kernel<4, MT>), grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL((conv_first16_kernel<8, MT>), grid, dim3(256), 0, s, p); break;
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// transposed conv k3 s2 p1 op1 as 4 output phases over a 2x2 input neighbourhood:
//   out[2i+a][2j+b] += in[i+di][j+dj] * W[ky][kx],  ky: 0->(a=1,di=1) 1->(a=0,di=0) 2->(a=1,di=0); same for kx.
// M tile = 16 input pixels of one row; each wave owns one input row of the block's 4-row tile.
// ------------------------------------------------------------------------------------------------
// SUM2: the `b` source is a sum of two lazily normalised tensors (ConvSrc::ptr2: the decoder's residual sum, never materialised):
// its chunks load both tensors and both tables and add the two activated values on the way into LDS.
// RW: input rows per wave (block tile = 4 RW rows).  RW = 2 doubles the MFMAs behind every barrier pair and every B fragment read
// (two rows x four phases x NT tiles = 128 accumulator registers at NT = 4: two blocks per CU instead of three).
// LAZY = false (raw sources only, as conv_wreg_kernel): no table fetch and no norm / PReLU arithmetic while staging
template <int NT, bool SUM2 = false, int RW = 1, bool LAZY = true>
__global__ __launch_bounds__(256, RW == 2 ? 2 : (SUM2 ? (NT == 4 ? 2 : 4) : (NT == 4 ? 3 : 2))) void convt_mfma_kernel(ConvParams p) {
    static_assert(LAZY || !SUM2, "a sum source is two lazily normalised tensors");
    constexpr int TH = 4 * RW, TW = 16, IH = TH + 1, IW = TW + 1, CS = 24;
    constexpr int IN_ITERS = (IH * IW + 63) / 64;
    constexpr int W_ITERS = (9 * NT * 64 + 255) / 256;
    constexpr int IN_FLOATS = (IH * IW * CS + 3) / 4 * 4;
    constexpr int EPS = 16 * NT + 4;
    constexpr int EP_FLOATS = 4 * 32 * EPS + 4 * 16 * NT * 2 * 2;   // per wave: ONE output row of 32 pixels per pass (two passes keep
                                                                    // the slabs below the staging footprint: 3 blocks per CU); + statistics
    constexpr int SM_FLOATS = (IN_FLOATS + 9 * NT * 256) > EP_FLOATS ? (IN_FLOATS + 9 * NT * 256) : EP_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];
    float* const in_s = smem;
    float* const w_s = smem + IN_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, g = lane >> 4;
    const WorkId wk = decode_work(p.tiles_x * p.tiles_y * p.n, p.ntile_total / NT);
    if (!wk.valid) return;
    int bid = wk.spatial;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int iy0 = ty * TH, ix0 = tx * TW;
    const int nt0 = wk.ct * NT;
    const int q = tid & 3;

    f32x4 acc[RW][4][NT];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[r][ph][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int goff[IN_ITERS];
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (tid >> 2) + 64 * it;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        goff[it] = (pix < IH * IW && gy < p.hin && gx < p.win) ? gy * p.win + gx : -1;
    }

    const int chunks_a = p.a.c >> 4, chunks = (p.a.c + p.b.c) >> 4;
    f32x4 rin[IN_ITERS], rw[W_ITERS];
    f32x4 rin2[SUM2 ? IN_ITERS : 1];
    Lazy4 lz, lz2;
    bool two = false;                                          // the chunk in flight comes from the SUM source
    auto load_chunk = [&](int ch) {
        const bool from_a = ch < chunks_a;
        const SrcDev& src = from_a ? p.a : p.b;
        const int c0 = (from_a ? ch : ch - chunks_a) << 4;
        if constexpr (LAZY) lz.load(src, img, c0 + 4 * q, p.wpack);
        const float* base = src.ptr + (size_t)img * p.hin * p.win * src.pitch + c0 + 4 * q;
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int o = goff[it] >= 0 ? goff[it] : 0;
            rin[it] = *reinterpret_cast<const f32x4*>(base + (size_t)o * src.pitch);
        }
        if constexpr (SUM2) {
            two = !from_a;
            if (two) {
                const SrcDev s2 = second_of(p.b);
                lz2.load(s2, img, c0 + 4 * q, p.wpack);
                const float* base2 = s2.ptr + (size_t)img * p.hin * p.win * s2.pitch + c0 + 4 * q;
#pragma unroll
                for (int it = 0; it < IN_ITERS; ++it) {
                    const int o = goff[it] >= 0 ? goff[it] : 0;
                    rin2[it] = *reinterpret_cast<const f32x4*>(base2 + (size_t)o * s2.pitch);
                }
            }
        }
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.wpack) + ((size_t)ch * 9 * p.ntile_total + nt0) * 64;
#pragma unroll
        for (int it = 0; it < W_ITERS; ++it) {
            const int i = tid + 256 * it;
            if ((9 * NT * 64) % 256 == 0 || i < 9 * NT * 64) {
                const int tap = i / (NT * 64), r = i - tap * (NT * 64);
                rw[it] = wsrc[(size_t)tap * p.ntile_total * 64 + r];
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int pix = (tid >> 2) + 64 * it;
            if (pix < IH * IW) {
                f32x4 v = rin[it];
                if constexpr (LAZY) v = lz.apply(v);
                if constexpr (SUM2)
                    if (two) v += lz2.apply(rin2[it]);        // (a raw second tensor - no table - is added as it is)
                if (goff[it] < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(&in_s[pix * CS + 4 * q]) = v;
            }
        }
#pragma unroll
        for (int it = 0; it < W_ITERS; ++it) {
            const int i = tid + 256 * it;
            if ((9 * NT * 64) % 256 == 0 || i < 9 * NT * 64) reinterpret_cast<f32x4*>(w_s)[i] = rw[it];
        }
    };

    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int ch = 0; ch < chunks; ++ch) {
        const bool more = ch + 1 < chunks;
        if (more) load_chunk(ch + 1);
        f32x4 a[RW + 1][2];
#pragma unroll
        for (int di = 0; di < RW + 1; ++di)
#pragma unroll
            for (int dj = 0; dj < 2; ++dj)
                a[di][dj] = *reinterpret_cast<const f32x4*>(&in_s[((wave * RW + di) * IW + pl + dj) * CS + 4 * g]);
        // (reading every B fragment one step ahead of its 4 MFMAs - as the stride-2 kernel does with its A fragments - was measured:
        // 10.4 vs 9.2 ms per slide.  The 64-channel variant sits at its register cap (166 of 168 for three blocks per CU): the second
        // fragment buffer spilled the weight prefetch to scratch and made the allocator rotate accumulators through s_nop hazards.)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int pa = (ky == 1) ? 0 : 1, di = (ky == 0) ? 1 : 0;
            const int pb = (kx == 1) ? 0 : 1, dj = (kx == 0) ? 1 : 0;
            const int ph = pa * 2 + pb;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 b = reinterpret_cast<const f32x4*>(w_s)[(tap * NT + nt) * 64 + lane];
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    acc[r][ph][nt] = mfma4(a[r + di][dj].x, b.x, acc[r][ph][nt]);
                    acc[r][ph][nt] = mfma4(a[r + di][dj].y, b.y, acc[r][ph][nt]);
                    acc[r][ph][nt] = mfma4(a[r + di][dj].z, b.z, acc[r][ph][nt]);
                    acc[r][ph][nt] = mfma4(a[r + di][dj].w, b.w, acc[r][ph][nt]);
                }
            }
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    float* ep = smem + wave * 32 * EPS;
    double* st_s = reinterpret_cast<double*>(smem + 4 * 32 * EPS);     // [4 waves][16*NT][2]: block-reduced statistics
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (nt0 + nt) * 16 + pl;
        const float bias = p.bias ? p.bias[co] : 0.f;
        double s = 0.0, qq = 0.0;
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int iy = iy0 + wave * RW + rr;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ix = ix0 + g * 4 + r;
                    const float v = acc[rr][ph][nt][r] + bias;
                    acc[rr][ph][nt][r] = v;
                    if (iy < p.hin && ix < p.win) {
                        s += (double)v;
                        qq += (double)v * (double)v;
                    }
                }
            }
        }
        if (p.stats) {
            s += __shfl_xor(s, 16);
            qq += __shfl_xor(qq, 16);
            s += __shfl_xor(s, 32);
            qq += __shfl_xor(qq, 32);
            if (lane < 16) {
                st_s[(wave * 16 * NT + nt * 16 + pl) * 2] = s;
                st_s[(wave * 16 * NT + nt * 16 + pl) * 2 + 1] = qq;
            }
        }
    }
    constexpr int F4_PER_PIX = NT * 4;
    const OutSel o = select_out(p, nt0);
    const int cols_valid = 2 * (p.win - ix0);
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int iy = iy0 + wave * RW + rr;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {                       // output row 2*iy + pa through the wave-private slab
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ep[(2 * (g * 4 + r) + pb) * EPS + nt * 16 + pl] = acc[rr][pa * 2 + pb][nt][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if (iy < p.hin) {
                const size_t pix0 = ((size_t)img * p.hout + 2 * iy + pa) * p.wout + 2 * ix0;
                float* out0 = o.out + pix0 * o.pitch + o.ch0;
                const float* add0 = o.add ? o.add + pix0 * o.pitch_add + o.ch0 : nullptr;
#pragma unroll
                for (int i = lane; i < 32 * F4_PER_PIX; i += 64) {
                    const int x = i / F4_PER_PIX, c4 = i - x * F4_PER_PIX;
                    if (x < cols_valid) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(&ep[x * EPS + c4 * 4]);
                        if (add0) v += *reinterpret_cast<const f32x4*>(add0 + (int64_t)x * o.pitch_add + c4 * 4);
                        *reinterpret_cast<f32x4*>(out0 + (int64_t)x * o.pitch + c4 * 4) = v;
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                // slab reads done before the next pass overwrites it
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (p.stats) {
        __syncthreads();
        if (tid < 32 * NT) {
            const int c = tid >> 1, kk = tid & 1;
            const double v = st_s[(0 * 16 * NT + c) * 2 + kk] + st_s[(1 * 16 * NT + c) * 2 + kk] + st_s[(2 * 16 * NT + c) * 2 + kk] +
                             st_s[(3 * 16 * NT + c) * 2 + kk];
            atomicAdd(p.stats + ((size_t)img * p.cout + nt0 * 16 + c) * 2 + kk, v);
        }
    }
}

int launch_convt_mfma(hipStream_t s, const ConvSrc& a, const ConvSrc& b, const float* wpack, const float* bias,
                      float* out, int pitch_out, int cout, double* stats, int n, int hin, int win, const ConvOutExtra* ex) {
    CS_REQUIRE(a.c % 16 == 0 && b.c % 16 == 0 && a.c > 0, "convT: input channels (%d,%d) must be multiples of 16", a.c, b.c);
    CS_REQUIRE(cout % 16 == 0, "convT: cout %d must be a multiple of 16", cout);
    CS_REQUIRE(!a.ptr2, "convT: only the second operand may be a sum source");
    ConvParams p{};
    p.a = to_dev(a);
    p.b = to_dev(b);
    p.wpack = wpack; p.bias = bias; p.out = out; p.stats = stats;
    p.pitch_out = pitch_out; p.cout = cout; p.ntile_total = cout / 16;
    p.n = n; p.hin = hin; p.win = win; p.hout = 2 * hin; p.wout = 2 * win;
    p.tiles_x = ceil_div(win, 16);
    p.tiles_y = ceil_div(hin, 4);
    int nt = (cout % 64 == 0) ? 4 : (cout % 32 == 0) ? 2 : 1;
    if (int rc = apply_extra(p, ex, cout, &nt)) return rc;
    // small launches (the deep levels of a batch-8 training step: 32 blocks of 64 channels with K = 384) take narrower channel blocks
    // until the grid covers the chip once: the same kernel, the same accumulation order, 2-4 x the blocks
    while (nt > 1 && xcd_grid(p.tiles_x * p.tiles_y * n, p.ntile_total / nt) < SMALL_GRID_BLOCKS) nt /= 2;
    // two input rows per wave for the 64-channel blocks when the grid still fills the chip at two blocks per CU
    if (nt == 4 && !b.ptr2 && hin >= 8 && xcd_grid(p.tiles_x * ceil_div(hin, 8) * n, p.ntile_total / nt) >= 1024) {
        p.tiles_y = ceil_div(hin, 8);
        dim3 grid2(xcd_grid(p.tiles_x * p.tiles_y * n, p.ntile_total / nt));
        const double flops2 = 2.0 * n * hin * win * (double)cout * (a.c + b.c) * 9;
        ProfScope prof2(K_CONVT, s, flops2, 4.0 * n * ((double)hin * win * (a.c + b.c) + 4.0 * hin * win * cout));
        hipLaunchKernelGGL((convt_mfma_kernel<4, false, 2>), grid2, dim3(256), 0, s, p);
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
    dim3 grid(xcd_grid(p.tiles_x * p.tiles_y * n, p.ntile_total / nt));
    const double flops = 2.0 * n * hin * win * (double)cout * (a.c + b.c) * 9;
    const double bytes = 4.0 * n * ((double)hin * win * (a.c + (b.ptr2 ? 2 : 1) * b.c) + 4.0 * hin * win * cout);
    ProfScope prof(K_CONVT, s, flops, bytes);
    if (b.ptr2) {
        CS_REQUIRE(b.c > 0, "convT: empty sum source");
        switch (nt) {
            case 4: hipLaunchKernelGGL((convt_mfma_kernel<4, true>), grid, dim3(256), 0, s, p); break;
            case 2: hipLaunchKernelGGL((convt_mfma_kernel<2, true>), grid, dim3(256), 0, s, p); break;
            default: hipLaunchKernelGGL((convt_mfma_kernel<1, true>), grid, dim3(256), 0, s, p); break;
        }
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
    const bool raw = !a.tab && (b.c == 0 || !b.tab);             // training: skip tensors, materialised sums, data gradients
    switch (nt) {
        case 4: if (raw) hipLaunchKernelGGL((convt_mfma_kernel<4, false, 1, false>), grid, dim3(256), 0, s, p);
                else hipLaunchKernelGGL((convt_mfma_kernel<4>), grid, dim3(256), 0, s, p);
                break;
        case 2: if (raw) hipLaunchKernelGGL((convt_mfma_kernel<2, false, 1, false>), grid, dim3(256), 0, s, p);
                else hipLaunchKernelGGL((convt_mfma_kernel<2>), grid, dim3(256), 0, s, p);
                break;
        default: if (raw) hipLaunchKernelGGL((convt_mfma_kernel<1, false, 1, false>), grid, dim3(256), 0, s, p);
                 else hipLaunchKernelGGL((convt_mfma_kernel<1>), grid, dim3(256), 0, s, p);
                 break;
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
// A training step re-packs every weight tensor after the optimiser update (63 tensors, forward + data-gradient layouts).
// Between pack_batch_begin() and pack_batch_flush() the pack_* / copy entry points only RECORD a job; the flush uploads the
// job table and runs two kernels: the tile kernel for the MFMA weight layouts (one block per 16 x 16 x taps weight tile: the
// tile is read once with coalesced row segments, staged in LDS and written out as whole 1 KB fragments, in the forward AND
// the data-gradient order) and an element-wise kernel for everything small (first layer, biases, slopes, 16-bit planes).
// Round 1 ran all of it element-wise with a per-element job search and strided gathers: 0.6-0.85 ms per training step for
// ~80 MB of traffic; the tile kernel moves the same bytes in tens of microseconds.
static inline int64_t pack_job_elems(const PackJob& jb) {
    switch (jb.kind) {
        case PK_CONV: return (int64_t)(jb.cin / 16) * (jb.cout / 16);       // BLOCKS: one per (chunk, channel tile)
        case PK_FIRST: return (int64_t)jb.ksteps * (jb.cout / 16) * 64;
        case PK_FIRST16: return (int64_t)(jb.cout / 16) * 512;
        case PK_BF16: return (int64_t)jb.chunks_total * jb.taps * (jb.cout / 16) * 64;     // OCTETS: 8 halfs of 3 planes per thread
        case PK_WINO: return (int64_t)(jb.cin / 16) * 16 * (jb.cout / 16) * 256;
        case PK_WINO4: return (int64_t)(jb.cin / 16) * 36 * (jb.cout / 16) * 256;
        case PK_CONVTW: return (int64_t)(jb.cin / 16) * 25 * (jb.cout / 16) * 256;
        default: return jb.cout;                              // PK_COPY: cout = element count
    }
}

template <typename T>
__device__ __forceinline__ int find_job(const PackJob* jobs, int n_jobs, T e) {
    int lo = 0, hi = n_jobs - 1;                              // last job with begin <= e
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].begin <= (long long)e) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void pack_jobs_kernel(const PackJob* jobs, int n_jobs, long long total) {
    // a block's 256 work items almost always belong to one job: one thread searches the table and the descriptor is read from LDS (every
    // thread searching the global table and copying its 120-byte descriptor was most of this kernel's time)
    __shared__ PackJob jb_s;
    __shared__ int single_s;
    const long long e0 = (long long)blockIdx.x * 256;
    if (threadIdx.x == 0) {
        const long long e1 = e0 + 255 < total ? e0 + 255 : total - 1;
        const int j0 = find_job(jobs, n_jobs, e0);
        single_s = j0 == find_job(jobs, n_jobs, e1) ? 1 : 0;
        jb_s = jobs[j0];
    }
    __syncthreads();
    const long long e = e0 + threadIdx.x;
    if (e >= total) return;
    PackJob own;
    if (!single_s) own = jobs[find_job(jobs, n_jobs, e)];
    const PackJob& jb = single_s ? jb_s : own;
    const int64_t le = e - jb.begin;
    switch (jb.kind) {
        case PK_FIRST: pack_first_elem(jb, le); break;
        case PK_FIRST16: pack_first16_elem(jb, le); break;
        case PK_BF16: pack_bf16_elem(jb, le); break;
        case PK_WINO: pack_wino_elem(jb, le); break;
        case PK_WINO4: pack_wino4_elem(jb, le); break;
        case PK_CONVTW: pack_convt_wino_elem(jb, le); break;
        default: reinterpret_cast<float*>(jb.dst)[le] = jb.src[le]; break;
    }
}

// one block = one 16 x 16 x TAPS tile of a conv weight.  T[r][c * taps + tap]: r = the tensor's slow channel axis (cout of a
// conv, cin of a transposed conv), c = the fast one.
__global__ __launch_bounds__(256) void pack_tiles_kernel(const PackJob* jobs, int n_jobs) {
    __shared__ float tile[16 * 145];
    const int tid = threadIdx.x;
    const PackJob jb = jobs[find_job(jobs, n_jobs, blockIdx.x)];
    const int b = (int)(blockIdx.x - jb.begin);
    const int ntiles = jb.cout >> 4, taps = jb.taps;
    const int chunk = b / ntiles, ntl = b - chunk * ntiles;
    const int row_len = 16 * taps, pitch = row_len + 1;
    const int row_base = jb.transposed ? chunk * 16 : ntl * 16, col_base = jb.transposed ? ntl * 16 : chunk * 16;
    const int slow_len = jb.transposed ? jb.cout : jb.cin;      // elements of the fast channel axis per slow-axis row
    for (int i = tid; i < 16 * row_len; i += 256) {
        const int r = i / row_len, e = i - r * row_len;
        tile[r * pitch + e] = jb.src[((size_t)(row_base + r) * slow_len + col_base) * taps + e];
    }
    __syncthreads();
    const int lane = tid >> 2, j = tid & 3;
    const int co_l = lane & 15, ci_l = 4 * (lane >> 4) + j;
    {   // forward layout: fragment element (lane, j) = W[cout = ntl*16 + co_l][cin = chunk*16 + ci_l]
        float* dst = reinterpret_cast<float*>(jb.dst);
        const int r = jb.transposed ? ci_l : co_l, c = jb.transposed ? co_l : ci_l;
        const int nt = (jb.cout_off >> 4) + ntl;
        for (int tap = 0; tap < taps; ++tap) {
            const int st = jb.flip ? taps - 1 - tap : tap;
            dst[(((size_t)(chunk + jb.chunk_off) * taps + tap) * jb.ntile_total + nt) * 256 + tid] = tile[r * pitch + c * taps + st];
        }
    }
    if (jb.dst2) {   // data-gradient layout: the channel axes trade places: "cout" = cin tile `chunk`, "cin" chunk = cout tile `ntl`
        float* dst = reinterpret_cast<float*>(jb.dst2);
        const int r = jb.transposed ? co_l : ci_l, c = jb.transposed ? ci_l : co_l;
        for (int tap = 0; tap < taps; ++tap) {
            const int st = jb.flip2 ? taps - 1 - tap : tap;
            dst[(((size_t)(ntl + jb.chunk_off2) * taps + tap) * jb.ntile_total2 + chunk) * 256 + tid] = tile[r * pitch + c * taps + st];
        }
    }    // 3x3 stride-1 layers (conv, not transposed): the same tile as Winograd weights U = G g G^T, 16 fragments instead of 9,
    // forward order and / or data-gradient order (channel roles swapped, taps reversed) - kernels_conv_wino.hip
    if (jb.dstw) {
        float g[9], u[16];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) g[tap] = tile[co_l * pitch + ci_l * 9 + tap];
        wino_weight_transform(g, u);
        float* dst = reinterpret_cast<float*>(jb.dstw);
        const int nt = (jb.cout_off >> 4) + ntl;
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) dst[(((size_t)chunk * 16 + pos) * jb.ntile_total + nt) * 256 + tid] = u[pos];
    }
    if (jb.dstw2) {
        float g[9], u[16];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) g[tap] = tile[ci_l * pitch + co_l * 9 + (8 - tap)];
        wino_weight_transform(g, u);
        float* dst = reinterpret_cast<float*>(jb.dstw2);
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) dst[(((size_t)ntl * 16 + pos) * jb.ntile_total2 + chunk) * 256 + tid] = u[pos];
    }
}

PackCtx* pack_ctx_create(int max_jobs) {
    PackCtx* pc = new PackCtx();
    pc->cap = (size_t)(max_jobs < 64 ? 64 : max_jobs);
    if (hipGetDevice(&pc->device) != hipSuccess || hipHostMalloc((void**)&pc->host, pc->cap * sizeof(PackJob), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&pc->dev, pc->cap * sizeof(PackJob)) != hipSuccess ||
        hipEventCreateWithFlags(&pc->copied, hipEventDisableTiming) != hipSuccess) {
        pack_ctx_destroy(pc);
        return nullptr;
    }
    pc->jobs.reserve(pc->cap);
    return pc;
}

void pack_ctx_destroy(PackCtx* pc) {
    if (!pc) return;
    if (pc->copied) {
        if (pc->copied_pending) (void)hipEventSynchronize(pc->copied);
        (void)hipEventDestroy(pc->copied);
    }
    if (pc->host) (void)hipHostFree(pc->host);
    if (pc->dev) (void)hipFree(pc->dev);
    delete pc;
}

void pack_batch_begin(PackCtx* pc) {
    pc->jobs.clear();
    pc->recording = true;
}

static int pack_submit(PackCtx* pc, hipStream_t s, const PackJob& jb) {
    if (pc->recording) {
        if (pc->jobs.size() >= pc->cap) {
            set_error("pack: more than %zu jobs in one batch", pc->cap);
            return CS_ERR_INVALID;
        }
        pc->jobs.push_back(jb);
        return CS_OK;
    }
    pc->jobs.assign(1, jb);                                   // immediate mode: a batch of one
    return pack_batch_flush(pc, s);
}

static int pack_launch(PackCtx* pc, hipStream_t s) {
    ProfScope prof(K_PACK, s, 0, pc->bytes);
    if (pc->total_elems > 0)
        hipLaunchKernelGGL(pack_jobs_kernel, dim3((unsigned)((pc->total_elems + 255) / 256)), dim3(256), 0, s, pc->dev, (int)pc->n_elem,
                           pc->total_elems);
    if (pc->total_blocks > 0)
        hipLaunchKernelGGL(pack_tiles_kernel, dim3((unsigned)pc->total_blocks), dim3(256), 0, s, pc->dev + pc->n_elem, (int)(pc->n_jobs - pc->n_elem));
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int pack_batch_flush(PackCtx* pc, hipStream_t s) {
    pc->recording = false;
    if (pc->jobs.empty()) return CS_OK;
    CS_REQUIRE(!pc->frozen, "pack: this job table is frozen (pack_replay re-runs it)");
    // element-wise jobs first, tile jobs behind them; `begin` = first element / first block inside its own group
    std::vector<PackJob>& jobs = pc->jobs;
    std::stable_partition(jobs.begin(), jobs.end(), [](const PackJob& j) { return j.kind != PK_CONV; });
    long long total_elems = 0, total_blocks = 0;
    size_t n_elem = 0;
    for (auto& jb : jobs) {
        if (jb.kind != PK_CONV) {
            jb.begin = total_elems;
            total_elems += pack_job_elems(jb);
            ++n_elem;
        } else {
            jb.begin = total_blocks;
            total_blocks += pack_job_elems(jb);
        }
    }
    const size_t n = jobs.size();
    // The pinned staging table is rewritten here and its device twin is overwritten by the copy below: the previous flush - its
    // copy out of the pinned table AND its pack kernels, which read the device table - must be complete.  `copied` is recorded
    // behind those kernels; the host waits for it (pinned table) and `s` is ordered behind it (device table: consecutive flushes
    // may arrive on different streams).
    if (pc->copied_pending) {
        CS_HIP(hipEventSynchronize(pc->copied));
        CS_HIP(hipStreamWaitEvent(s, pc->copied, 0));
    }
    memcpy(pc->host, jobs.data(), n * sizeof(PackJob));
    CS_HIP(hipMemcpyAsync(pc->dev, pc->host, n * sizeof(PackJob), hipMemcpyHostToDevice, s));
    double bytes = 8.0 * (double)total_elems;
    for (size_t i = n_elem; i < n; ++i)
        bytes += 4.0 * 256 * pack_job_elems(jobs[i]) * (jobs[i].taps * (jobs[i].dst2 ? 3 : 2) + 16 * ((jobs[i].dstw ? 1 : 0) + (jobs[i].dstw2 ? 1 : 0)));
    pc->n_elem = n_elem; pc->n_jobs = n; pc->total_elems = total_elems; pc->total_blocks = total_blocks; pc->bytes = bytes;
    if (int rc = pack_launch(pc, s)) return rc;
    CS_HIP(hipEventRecord(pc->copied, s));
    pc->copied_pending = true;
    jobs.clear();
    return CS_OK;
}

// Flush once and keep the device job table for good: pack_replay() then re-runs exactly these jobs (same sources, same
// destinations) with two kernel launches and nothing else - no job recording, no table upload, nothing a stream capture cannot
// hold.  Used for the per-step re-pack of a training loop whose parameter tensors stay where they are.
int pack_batch_freeze(PackCtx* pc, hipStream_t s) {
    if (int rc = pack_batch_flush(pc, s)) return rc;
    CS_HIP(hipStreamSynchronize(s));                          // the pinned table has been read: nothing refers to it any more
    pc->frozen = true;
    return CS_OK;
}

int pack_replay(PackCtx* pc, hipStream_t s) {
    CS_REQUIRE(pc && pc->frozen, "pack: no frozen job table to replay");
    if (pc->n_jobs == 0) return CS_OK;
    return pack_launch(pc, s);
}

int pack_conv_weights(PackCtx* pc, hipStream_t s, const float* w, bool transposed, int cout, int cin, int ksize, float* packed,
                      int cout_off, int cout_total, float* dgrad, bool dgrad_flip, int dgrad_chunk_off, float* wino, float* wino_dgrad) {
    CS_REQUIRE(cout % 16 == 0 && cin % 16 == 0 && cout_off % 16 == 0 && cout_total % 16 == 0, "pack: bad channel counts");
    CS_REQUIRE(ksize == 3 || ksize == 1, "pack: ksize %d", ksize);
    PackJob jb{};
    jb.src = w; jb.dst = packed; jb.kind = PK_CONV; jb.transposed = transposed ? 1 : 0; jb.cout = cout; jb.cin = cin;
    jb.taps = ksize * ksize; jb.cout_off = cout_off; jb.ntile_total = cout_total / 16;
    jb.dst2 = dgrad; jb.flip2 = dgrad_flip ? 1 : 0; jb.chunk_off2 = dgrad_chunk_off; jb.ntile_total2 = cin / 16;
    CS_REQUIRE(!(wino || wino_dgrad) || (!transposed && ksize == 3 && cout_off == 0 && cout_total == cout), "pack: Winograd weights are for plain 3x3 convs");
    jb.dstw = wino; jb.dstw2 = wino_dgrad;
    return pack_submit(pc, s, jb);
}

int pack_first_weights(PackCtx* pc, hipStream_t s, const float* w, int cout, int cin, float* packed, int cout_off, int cout_total) {
    PackJob jb{};
    jb.src = w; jb.dst = packed; jb.kind = PK_FIRST; jb.cout = cout; jb.cin = cin; jb.cout_off = cout_off; jb.ntile_total = cout_total / 16;
    jb.ksteps = (9 * cin + 3) / 4;
    return pack_submit(pc, s, jb);
}

int pack_first_weights16(PackCtx* pc, hipStream_t s, const float* w, int cout, int cin, unsigned short* packed16, int cout_off) {
    CS_REQUIRE(9 * cin <= 32, "pack(first16): %d input channels do not fit one K-step", cin);
    PackJob jb{};
    jb.src = w; jb.dst = packed16; jb.kind = PK_FIRST16; jb.cout = cout; jb.cin = cin; jb.cout_off = cout_off;
    return pack_submit(pc, s, jb);
}

int pack_copy(PackCtx* pc, hipStream_t s, const float* src, float* dst, size_t n) {
    PackJob jb{};
    jb.src = src; jb.dst = dst; jb.kind = PK_COPY; jb.cout = (int)n;
    return pack_submit(pc, s, jb);
}

int pack_bf16_submit(PackCtx* pc, hipStream_t s, const float* w, bool transposed, int cout, int cin, int taps, unsigned short* packed,
                     int cout_off, int ntile_total, int cin_split, int chunks_total, bool f16, bool flip, int chunk_off) {
    PackJob jb{};
    jb.flip = flip ? 1 : 0;
    jb.chunk_off = chunk_off;
    jb.src = w; jb.dst = packed; jb.kind = PK_BF16; jb.transposed = transposed ? 1 : 0; jb.cout = cout; jb.cin = cin; jb.taps = taps;
    jb.cout_off = cout_off; jb.ntile_total = ntile_total; jb.cin_split = cin_split; jb.chunks_total = chunks_total; jb.f16 = f16 ? 1 : 0;
    return pack_submit(pc, s, jb);
}

int pack_wino_weights(PackCtx* pc, hipStream_t s, const float* w, bool dgrad, int cout, int cin, float* packed) {
    CS_REQUIRE(cout % 16 == 0 && cin % 16 == 0, "pack(winograd): bad channel counts");
    PackJob jb{};
    jb.src = w; jb.dst = packed; jb.kind = PK_WINO; jb.transposed = dgrad ? 1 : 0; jb.flip = dgrad ? 1 : 0; jb.cout = cout; jb.cin = cin;
    jb.ntile_total = cout / 16;
    return pack_submit(pc, s, jb);
}

int pack_wino4_weights(PackCtx* pc, hipStream_t s, const float* w, int cout, int cin, float* packed, bool dgrad) {
    CS_REQUIRE(cout % 16 == 0 && cin % 16 == 0, "pack(winograd 4x4): bad channel counts");
    PackJob jb{};
    jb.transposed = dgrad ? 1 : 0; jb.flip = dgrad ? 1 : 0;
    jb.src = w; jb.dst = packed; jb.kind = PK_WINO4; jb.cout = cout; jb.cin = cin; jb.ntile_total = cout / 16;
    jb.f16 = wino4_layout32(cin, cout) ? 1 : 0;      // (field reused) 1: the 32x32x2 fragment order of kernels_conv_wino4x.hip
    return pack_submit(pc, s, jb);
}

int pack_convt_wino_weights(PackCtx* pc, hipStream_t s, const float* w, int cout, int cin, float* packed) {
    CS_REQUIRE(cout % 16 == 0 && cin % 16 == 0, "pack(transposed conv, winograd): bad channel counts");
    PackJob jb{};
    jb.src = w; jb.dst = packed; jb.kind = PK_CONVTW; jb.cout = cout; jb.cin = cin; jb.ntile_total = cout / 16; jb.transposed = 1;
    return pack_submit(pc, s, jb);
}

size_t packed_conv_floats(int cout_total, int cin_total, int ksize) {
    return (size_t)(cin_total / 16) * ksize * ksize * (cout_total / 16) * 256;
}
size_t packed_first_floats(int cout_total, int cin) { return (size_t)((9 * cin + 3) / 4) * (cout_total / 16) * 64; }

}  // namespace cs
