// gfx950 (MI355X): the 3x3 stride-1 convolutions of the U-Net in Winograd form F(2x2, 3x3) on the fp32 matrix pipe.
//
// Replaces the same torch conv2d dispatches as kernels_conv.hip (reference call sites baseline/predict.py:109 -> :45-52:
// the stride-1 subunits of every monai ResidualUnit); all arithmetic is fp32 (v_mfma_f32_16x16x4_f32 + fp32 VALU
// transforms), the logits stay inside the 1e-4 budget (tests/diagnostics/winograd_precision.py: 6.7e-6 vs fp64, the
// direct fp32 form 7.0e-6).
//
// Why: the fp32 matrix pipe runs at the fp32 VECTOR rate (157 TFLOP/s), so the direct kernels are bound by the number of
// multiplies.  Y = A^T [ (G g G^T) . (B^T d B) ] A needs 16 multiplies per 2x2 output tile and (cin, cout) pair instead of
// 36: 2.25x fewer matrix-pipe cycles for the layers that hold 65 % of the network's FLOPs.
//
// Shape of the kernel (one block = 256 threads = 4 waves, 8 x 16 output pixels = 4 x 8 Winograd tiles, 16*NTW channels):
//   * K loop over 16-channel chunks.  The halo'd 10 x 18 input patch of the chunk goes global -> registers one chunk
//     ahead (lazy InstanceNorm + PReLU of the producer applied on the way, zero padding after the activation) -> LDS
//     `raw` [180 px][20]; then every thread transforms one (tile, 2 channels) 4x4 patch (B^T d B: 64 adds) and writes
//     the 16 transform-domain values to `V` [16 positions][4 k-groups][32 tiles][4], the A-fragment order of the MFMA
//     (ds_read_b128, bank-conflict free without padding).
//   * wave xi owns the 4 positions (xi, nu = 0..3) for all 32 tiles (2 M-tiles) and all NTW channel tiles: per chunk and nu
//     2 A fragments from LDS, NTW B fragments L2 -> registers one nu ahead (the transformed weights never touch LDS),
//     2 * NTW * 4 MFMAs; 32 * NTW accumulator registers.
//   * epilogue: the nu half of A^T M A is taken in registers (the wave holds a whole row of positions), the xi half after
//     an exchange through LDS; bias, fp64 InstanceNorm statistics, whole 64 * NTW byte pixel records to HBM.
#include "conv_common.h"
#include "pack_jobs.h"

#include <stdlib.h>

namespace cs {

namespace {
constexpr int WIH = 10, WIW = 18;                  // input patch of a block
constexpr int WPS = 20;                            // floats per staged raw pixel (16 + 4: conflict-free transform reads)
constexpr int RAW_FLOATS = WIH * WIW * WPS;        // 3600
constexpr int V_FLOATS = 16 * 4 * 32 * 4;          // 8192
}  // namespace

template <int NTW, int OCC, bool LAZY>
__global__ __launch_bounds__(256, OCC) void conv_wino_kernel(ConvParams p) {
    constexpr int IN_ITERS = (WIH * WIW + 63) / 64;            // 3
    constexpr int RS = 16 * NTW + 4;                           // floats per (xi, j, tile) record of the epilogue exchange
    constexpr int R_FLOATS = 4 * 2 * 32 * RS;
    constexpr int MAIN_FLOATS = RAW_FLOATS + V_FLOATS;
    constexpr int BODY_FLOATS = MAIN_FLOATS > R_FLOATS ? MAIN_FLOATS : R_FLOATS;
    constexpr int ST_FLOATS = 4 * 16 * NTW * 2 * 2;            // fp64 statistics of the 4 waves
    __shared__ __attribute__((aligned(16))) float smem[BODY_FLOATS + ST_FLOATS];
    float* const raw = smem;
    float* const vs = smem + RAW_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: everything indexed by the wave id stays on the SALU
    const int pl = lane & 15, g = lane >> 4;
    const WorkId wk = decode_work(p.tiles_x * p.tiles_y * p.n, p.ntile_total / NTW);
    if (!wk.valid) return;
    int bid = wk.spatial;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int oy0 = ty * 8, ox0 = tx * 16;
    const int nt0 = wk.ct * NTW;
    const int q = tid & 3;

    f32x4 acc[4][2][NTW];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[nu][m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-thread staging geometry, fixed for the whole K loop: BYTE offset of the pixel's channel quad inside the image's tensor
    // (32 bits: the launcher checks the tensor size) and a padding bit; per chunk only the scalar base moves, so the loop carries
    // no 64-bit vector address arithmetic (on the fp32 path every VALU instruction is taken from the MFMA stream's cycles)
    unsigned boff[IN_ITERS], inb = 0;
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (tid >> 2) + 64 * it;
        const int iy = pix / WIW, ix = pix - iy * WIW;
        const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
        const bool ok = pix < WIH * WIW && gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win;
        inb |= ok ? (1u << it) : 0u;
        boff[it] = ok ? ((unsigned)(gy * p.win + gx) * (unsigned)p.a.pitch + 4u * q) * 4u : 16u * q;
    }
    const char* const img_base = reinterpret_cast<const char*>(p.a.ptr + (size_t)img * p.hin * p.win * p.a.pitch);

    const int chunks = p.a.c >> 4;
    f32x4 rin[IN_ITERS];
    Lazy4 lz;
    auto load_in = [&](int ch) {
        const int c0 = ch << 4;
        if (LAZY) lz.load(p.a, img, c0 + 4 * q, p.wpack);
        const char* base = img_base + (size_t)c0 * 4;
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it)      // unconditional (clamped) loads; padding is zeroed at store time
            rin[it] = *reinterpret_cast<const f32x4*>(base + boff[it]);
    };
    auto store_raw = [&]() {
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int pix = (tid >> 2) + 64 * it;
            if (pix < WIH * WIW) {
                const f32x4 a = LAZY ? lz.apply_on(rin[it]) : rin[it];
                const f32x4 v = (inb >> it) & 1u ? a : f32x4{0.f, 0.f, 0.f, 0.f};   // padding stays 0 after the activation
                *reinterpret_cast<f32x4*>(&raw[pix * WPS + 4 * q]) = v;
            }
        }
    };
    // input transform V = B^T d B of one (tile, channel pair): wave = tile row, lane = (k-group, channel pair, tile column)
    const int t_tx = lane & 7, t_h = (lane >> 3) & 1, t_g = lane >> 4;
    const float* const t_src = raw + ((2 * wave) * WIW + 2 * t_tx) * WPS + 4 * t_g + 2 * t_h;
    float* const t_dst = vs + (t_g * 32 + wave * 8 + t_tx) * 4 + 2 * t_h;
    auto transform = [&]() {
        float2 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const float2*>(t_src + (i * WIW + j) * WPS);
        float2 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = make_float2(d[0][j].x - d[2][j].x, d[0][j].y - d[2][j].y);
            t[1][j] = make_float2(d[1][j].x + d[2][j].x, d[1][j].y + d[2][j].y);
            t[2][j] = make_float2(d[2][j].x - d[1][j].x, d[2][j].y - d[1][j].y);
            t[3][j] = make_float2(d[1][j].x - d[3][j].x, d[1][j].y - d[3][j].y);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 v0 = make_float2(t[i][0].x - t[i][2].x, t[i][0].y - t[i][2].y);
            const float2 v1 = make_float2(t[i][1].x + t[i][2].x, t[i][1].y + t[i][2].y);
            const float2 v2 = make_float2(t[i][2].x - t[i][1].x, t[i][2].y - t[i][1].y);
            const float2 v3 = make_float2(t[i][1].x - t[i][3].x, t[i][1].y - t[i][3].y);
            *reinterpret_cast<float2*>(t_dst + (i * 4 + 0) * 512) = v0;
            *reinterpret_cast<float2*>(t_dst + (i * 4 + 1) * 512) = v1;
            *reinterpret_cast<float2*>(t_dst + (i * 4 + 2) * 512) = v2;
            *reinterpret_cast<float2*>(t_dst + (i * 4 + 3) * 512) = v3;
        }
    };

    // B fragments: packed[chunk][position 16][ntile_total][64 lanes][4]; this wave reads positions 4*wave + nu
    // (scalar base + 32-bit lane offset: no vector address arithmetic in the loop)
    const char* const wbase = reinterpret_cast<const char*>(p.wpack) + ((size_t)nt0 * 64 + (size_t)wave * 4 * p.ntile_total * 64) * 16;
    const size_t wpos = (size_t)p.ntile_total * 64 * 16;              // bytes between positions
    const unsigned wlane = (unsigned)lane * 16u;
    f32x4 wc[2][NTW];                          // two slots, refilled ONE nu step (8 * NTW MFMAs) ahead
    const int last_step = chunks * 4 - 1;
    auto wload = [&](int step, f32x4 (&dst)[NTW]) {
        const char* w = wbase + (size_t)((step >> 2) * 16 + (step & 3)) * wpos;
#pragma unroll
        for (int t = 0; t < NTW; ++t) dst[t] = *reinterpret_cast<const f32x4*>(w + t * 1024 + wlane);
    };

    load_in(0);
    wload(0, wc[0]);
    store_raw();
    __syncthreads();
    transform();
    __syncthreads();
    const float* const a_src = vs + ((wave * 4 * 4 + g) * 32 + pl) * 4;   // + nu * 512 + m * 64
    for (int ch = 0; ch < chunks; ++ch) {
        const bool more = ch + 1 < chunks;
        load_in(more ? ch + 1 : ch);           // unconditional: see conv_wreg_kernel
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            {
                const int step = ch * 4 + nu + 1;
                wload(step < last_step ? step : last_step, wc[(nu + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 a[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const f32x4*>(a_src + nu * 512 + m * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int t = 0; t < NTW; ++t) acc[nu][m][t] = mfma4(a[m][j], wc[nu & 1][t][j], acc[nu][m][t]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) asm volatile("" ::"v"(rin[it]));   // keeps the loads above the MFMA section
        if (more) store_raw();
        __syncthreads();
        if (more) {
            transform();
            __syncthreads();
        }
    }

    // ---- epilogue ----
    // nu half of the output transform in registers: R[xi][j] = sum_nu M[xi][nu] A[nu][j]
    float* const rs = smem;                                      // [xi 4][j 2][tile 32][RS]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const f32x4 r0 = (acc[0][m][t] + acc[1][m][t]) + acc[2][m][t];
            const f32x4 r1 = (acc[1][m][t] - acc[2][m][t]) - acc[3][m][t];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tile = m * 16 + g * 4 + r;
                rs[((wave * 2 + 0) * 32 + tile) * RS + t * 16 + pl] = r0[r];
                rs[((wave * 2 + 1) * 32 + tile) * RS + t * 16 + pl] = r1[r];
            }
        }
    __syncthreads();
    // xi half + bias + statistics + stores: thread = (pixel, channel quad)
    constexpr int F4 = NTW * 4, PPP = 256 / F4, XS = 2 * 32 * RS;
    const int c4 = tid % F4;
    const int co = nt0 * 16 + c4 * 4;
    // (no (out, out2) split in the Winograd kernels: none of their callers needs one)
    const OutSel o = OutSel{p.out, p.addend, p.pitch_out, p.pitch_add, (nt0 + c4 / 4) * 16};
    const int cw = (c4 & 3) * 4;
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + co);
    // Output addressing without 64-bit vector arithmetic: the image's base is a scalar, a thread's first pixel one 32-bit element
    // offset (the launcher keeps an image's tensor below 4 GB), and every pass of the loop moves a whole number of rows down, i.e.
    // adds a loop-invariant stride.  (Written as ((size_t)img * hout + oy) * wout + ox the loop carried 25 quarter-rate v_mul_lo_u32
    // and 12 v_mad_u64_u32 per wave - on the fp32 path every VALU cycle is taken from the MFMA stream's.)
    constexpr int ROWS_PER_IT = PPP / 16;
    const int pix0 = tid / F4, oyl0 = pix0 >> 4, oxl = pix0 & 15;
    const int ox = ox0 + oxl;
    float* const out_img = o.out + (size_t)img * p.hout * p.wout * o.pitch;
    const float* const add_img = o.add ? o.add + (size_t)img * p.hout * p.wout * o.pitch_add : nullptr;
    unsigned off_out = (unsigned)((oy0 + oyl0) * p.wout + ox) * (unsigned)o.pitch + (unsigned)(o.ch0 + cw);
    unsigned off_add = (unsigned)((oy0 + oyl0) * p.wout + ox) * (unsigned)o.pitch_add + (unsigned)(o.ch0 + cw);
    const unsigned step_out = (unsigned)(ROWS_PER_IT * p.wout) * (unsigned)o.pitch, step_add = (unsigned)(ROWS_PER_IT * p.wout) * (unsigned)o.pitch_add;
    const int jj = oxl & 1;
    const float* const rb0 = rs + (jj * 32 + (oxl >> 1)) * RS + c4 * 4;      // + (oyl >> 1) * 8 * RS per row pair
    float sf[4] = {0.f, 0.f, 0.f, 0.f}, qf[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 128 / PPP; ++it) {
        const int oyl = oyl0 + ROWS_PER_IT * it;
        const int ii = oyl & 1;
        const float* rb = rb0 + (oyl >> 1) * 8 * RS;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(rb + (ii ? 3 * XS : 0));
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(rb + XS);
        const f32x4 x2 = *reinterpret_cast<const f32x4*>(rb + 2 * XS);
        f32x4 y = ii ? (x1 - x2) - x0 : (x0 + x1) + x2;
        y += bias4;
        if (oy0 + oyl < p.hout && ox < p.wout) {
            if (add_img) y += *reinterpret_cast<const f32x4*>(add_img + off_add);
            *reinterpret_cast<f32x4*>(out_img + off_out) = y;
            if (p.stats) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    sf[k] += y[k];
                    qf[k] = __builtin_fmaf(y[k], y[k], qf[k]);
                }
            }
        }
        off_out += step_out;
        off_add += step_add;
    }
    if (p.stats) {
        // A thread's (at most 8) values per channel and then the 64 / F4 lanes that hold the same channels are summed in fp32 - at most
        // 32 values per partial sum, a relative error of a few 2^-24, below the rounding the values themselves carry - and everything
        // from there on (the block's 4 waves through LDS, the per-(image, channel) atomics, mean and variance) is fp64.  (Shuffling
        // the sums as doubles cost 48 ds_bpermute + 24 v_add_f64 per wave in this epilogue; a butterfly that halves the values a lane
        // carries at every step was measured slower still.)
        double* st_s = reinterpret_cast<double*>(smem + BODY_FLOATS);      // [4 waves][16*NTW][2]
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int sh = F4; sh < 64; sh <<= 1) {
                sf[k] += __shfl_xor(sf[k], sh);
                qf[k] += __shfl_xor(qf[k], sh);
            }
        }
        if (lane < F4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                st_s[(wave * 16 * NTW + c4 * 4 + k) * 2] = (double)sf[k];
                st_s[(wave * 16 * NTW + c4 * 4 + k) * 2 + 1] = (double)qf[k];
            }
        }
        __syncthreads();
        if (tid < 32 * NTW) {
            const int c = tid >> 1, k = tid & 1;
            const double v = st_s[(0 * 16 * NTW + c) * 2 + k] + st_s[(1 * 16 * NTW + c) * 2 + k] + st_s[(2 * 16 * NTW + c) * 2 + k] +
                             st_s[(3 * 16 * NTW + c) * 2 + k];
            atomicAdd(p.stats + ((size_t)img * p.cout + nt0 * 16 + c) * 2 + k, v);
        }
    }
}


// A persistent, role-split form of this kernel (one 512-thread block per CU: four "matrix" waves issuing only MFMAs, four
// "staging" waves doing loads / norm / transform / output transform one chunk ahead through a double-buffered V, one barrier per
// chunk) was built and measured in round 2 (git history: "persistent role-split kernel"): its matrix waves alone ran the K loop at
// 93 % of the MFMA rate, but with the staging waves active the layer times were the SUM of the two parts, not the maximum (512->512
// at 256 windows: 1104 us matrix-only + ~600 us staging = 1700 us, against 1433 us for this kernel).  v_mfma_f32_16x16x4_f32 runs at
// the fp32 VECTOR rate and, unlike the 16-bit MFMAs, does not execute beside another wave's VALU instructions on the same SIMD -
// so on the fp32 path the only lever is the number of VALU instructions per MFMA and not stalling, and specialising waves buys
// nothing.  Also measured without effect (+-1.5 %): a channel-tile-major block order (the blocks resident on an XCD stream ONE weight
// slice) and a weight ring that runs a whole chunk ahead - the weight stream is not what this kernel waits for; a 128-channel form
// with 512 threads (both channel halves contracting the same V, two waves per SIMD from the same block, one barrier per chunk):
// 512->512 in 1469 us against 1438 us (barrier skew); 32-channel blocks at three per CU for every layer: 10 % slower; several
// tiles per block for the short-K layers (the next tile's first patch requested before the epilogue of the current one): no
// change for 32->32 (683 us) - the wait for the first patch is already covered by the other blocks of the CU.

// ------------------------------------------------------------------------------------------------
// 128-channel form for the layers with cout % 128 == 0 (60 % of the Winograd FLOPs of the 32-base net): ONE wave per SIMD.
// Because an fp32 MFMA and another wave's VALU instruction do not execute together, a second wave on the SIMD buys overlap
// only for WAITS - and those can be scheduled away by hand when a wave has the SIMD to itself.  So: 256 threads, every wave
// owns its 4 positions for all 32 tiles and all 8 channel tiles (256 accumulator registers in the AGPR half of the file; every
// staged value feeds twice the MFMAs of the 64-channel kernel), raw patch and V are double-buffered around ONE barrier per chunk,
// and the staging of the next chunks is cut into pieces that sit BETWEEN the four nu steps of the running chunk:
//     [LDS reads of the transform of chunk c+1] MFMA nu=0 [transform arithmetic + V stores] MFMA nu=1
//     [norm + park of chunk c+2, global loads of chunk c+3] MFMA nu=2 MFMA nu=3 [barrier]
// every LDS round trip and the global latency run under 64 MFMAs; what is left in the MFMA stream is the VALU work itself.
// LDS: 2 x (14 KB + 32 KB) in the loop; the epilogue exchange R [4][2][32][132] (132 KB) + statistics reuse it: one block per CU.
// ------------------------------------------------------------------------------------------------
template <bool LAZY>
__global__ __launch_bounds__(256, 1) void conv_wino1_kernel(ConvParams p) {
    constexpr int NTW = 8;
    constexpr int IN_ITERS = (WIH * WIW + 63) / 64;            // 3
    constexpr int RS = 16 * NTW + 4;
    constexpr int R_FLOATS = 4 * 2 * 32 * RS;                  // 33792
    constexpr int MAIN_FLOATS = 2 * (RAW_FLOATS + V_FLOATS);   // 23584
    constexpr int BODY_FLOATS = MAIN_FLOATS > R_FLOATS ? MAIN_FLOATS : R_FLOATS;
    constexpr int ST_FLOATS = 4 * 16 * NTW * 2 * 2;
    __shared__ __attribute__((aligned(16))) float smem[BODY_FLOATS + ST_FLOATS];
    float* const raw0 = smem;                                  // raw[b] = raw0 + b * RAW_FLOATS
    float* const vs0 = smem + 2 * RAW_FLOATS;                  // V[b]   = vs0 + b * V_FLOATS

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const WorkId wk = decode_work(p.tiles_x * p.tiles_y * p.n, p.ntile_total / NTW);
    if (!wk.valid) return;
    int bid = wk.spatial;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int oy0 = ty * 8, ox0 = tx * 16;
    const int nt0 = wk.ct * NTW;
    const int q = tid & 3;

    f32x4 acc[4][2][NTW];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[nu][m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned boff[IN_ITERS], inb = 0;
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (tid >> 2) + 64 * it;
        const int iy = pix / WIW, ix = pix - iy * WIW;
        const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
        const bool ok = pix < WIH * WIW && gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win;
        inb |= ok ? (1u << it) : 0u;
        boff[it] = ok ? ((unsigned)(gy * p.win + gx) * (unsigned)p.a.pitch + 4u * q) * 4u : 16u * q;
    }
    const char* const img_base = reinterpret_cast<const char*>(p.a.ptr + (size_t)img * p.hin * p.win * p.a.pitch);
    const int chunks = p.a.c >> 4;
    f32x4 rin[IN_ITERS];
    Lazy4 lz;
    auto load_in = [&](int ch) {                               // unconditional (clamped past the end: re-reads the last chunk)
        const int c0 = (ch < chunks ? ch : chunks - 1) << 4;
        if (LAZY) lz.load(p.a, img, c0 + 4 * q, p.wpack);
        const char* base = img_base + (size_t)c0 * 4;
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) rin[it] = *reinterpret_cast<const f32x4*>(base + boff[it]);
    };
    auto park = [&](float* raw) {                              // registers -> raw patch (lazy norm + PReLU, zero padding after it)
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int pix = (tid >> 2) + 64 * it;
            if (pix < WIH * WIW) {
                const f32x4 a = LAZY ? lz.apply_on(rin[it]) : rin[it];
                const f32x4 v = (inb >> it) & 1u ? a : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(&raw[pix * WPS + 4 * q]) = v;
            }
        }
    };
    // input transform of one (tile, channel pair), in two pieces: the LDS reads, then arithmetic + V stores
    const int t_tx = lane & 7, t_h = (lane >> 3) & 1, t_g = lane >> 4;
    const int t_src_off = ((2 * wave) * WIW + 2 * t_tx) * WPS + 4 * t_g + 2 * t_h;
    const int t_dst_off = (t_g * 32 + wave * 8 + t_tx) * 4 + 2 * t_h;
    float2 d[4][4];
    auto transform_read = [&](const float* raw) {
        const float* src = raw + t_src_off;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const float2*>(src + (i * WIW + j) * WPS);
    };
    auto transform_write = [&](float* vs) {
        float2 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = make_float2(d[0][j].x - d[2][j].x, d[0][j].y - d[2][j].y);
            t[1][j] = make_float2(d[1][j].x + d[2][j].x, d[1][j].y + d[2][j].y);
            t[2][j] = make_float2(d[2][j].x - d[1][j].x, d[2][j].y - d[1][j].y);
            t[3][j] = make_float2(d[1][j].x - d[3][j].x, d[1][j].y - d[3][j].y);
        }
        float* dst = vs + t_dst_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float2*>(dst + (i * 4 + 0) * 512) = make_float2(t[i][0].x - t[i][2].x, t[i][0].y - t[i][2].y);
            *reinterpret_cast<float2*>(dst + (i * 4 + 1) * 512) = make_float2(t[i][1].x + t[i][2].x, t[i][1].y + t[i][2].y);
            *reinterpret_cast<float2*>(dst + (i * 4 + 2) * 512) = make_float2(t[i][2].x - t[i][1].x, t[i][2].y - t[i][1].y);
            *reinterpret_cast<float2*>(dst + (i * 4 + 3) * 512) = make_float2(t[i][1].x - t[i][3].x, t[i][1].y - t[i][3].y);
        }
    };

    // B fragments of this wave: positions 4 wave + nu, channel tiles nt0 + t (scalar base + 32-bit lane offset), one nu step ahead
    const char* const wbase = reinterpret_cast<const char*>(p.wpack) + ((size_t)nt0 * 64 + (size_t)wave * 4 * p.ntile_total * 64) * 16;
    const size_t wpos = (size_t)p.ntile_total * 64 * 16;
    const unsigned wlane = (unsigned)lane * 16u;
    // four slots: the fragments of a whole chunk, refilled one CHUNK ahead (a lone wave has nobody to cover a late fragment: one nu
    // step = 2048 cycles of lead was not always enough for the L2 under the weight streams of 256 CUs)
    f32x4 wc[4][NTW];
    auto wload = [&](int step, f32x4 (&dst)[NTW]) {
        const char* w = wbase + (size_t)((step >> 2) * 16 + (step & 3)) * wpos;
#pragma unroll
        for (int t = 0; t < NTW; ++t) dst[t] = *reinterpret_cast<const f32x4*>(w + t * 1024 + wlane);
    };

    // prologue: chunk 0 -> raw[0] -> V[0]; chunk 1 -> raw[1]; chunk 2 in registers
    load_in(0);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) wload(nu, wc[nu]);
    park(raw0);
    load_in(1);
    __syncthreads();
    transform_read(raw0);
    transform_write(vs0);
    park(raw0 + RAW_FLOATS);
    load_in(2);
    __syncthreads();
    const int a_off = ((wave * 16 + g) * 32 + pl) * 4;            // + nu * 512 + m * 64
    for (int ch = 0; ch < chunks; ++ch) {
        const int b = ch & 1;
        const bool more1 = ch + 1 < chunks, more2 = ch + 2 < chunks;
        const float* const a_src = vs0 + b * V_FLOATS + a_off;
        f32x4 a[2][2];
        // piece 1: the transform's LDS reads of chunk c + 1 and the first A fragments, all in flight under the first MFMAs
        if (more1) transform_read(raw0 + (b ^ 1) * RAW_FLOATS);
#pragma unroll
        for (int m = 0; m < 2; ++m) a[0][m] = *reinterpret_cast<const f32x4*>(a_src + m * 64);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            if (nu < 3) {
#pragma unroll
                for (int m = 0; m < 2; ++m) a[(nu + 1) & 1][m] = *reinterpret_cast<const f32x4*>(a_src + (nu + 1) * 512 + m * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int t = 0; t < NTW; ++t) acc[nu][m][t] = mfma4(a[nu & 1][m][j], wc[nu][t][j], acc[nu][m][t]);
            __builtin_amdgcn_sched_barrier(0);
            // last k-step of the fragments: channel tile by channel tile, each slot refilled (same nu of the NEXT chunk; the last chunk
            // re-reads its own) right behind its last two MFMAs - eight loads spread under sixteen MFMAs instead of one burst
            {
                const char* wn = wbase + (size_t)((more1 ? ch + 1 : ch) * 16 + nu) * wpos;
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    acc[nu][0][t] = mfma4(a[nu & 1][0][3], wc[nu][t][3], acc[nu][0][t]);
                    acc[nu][1][t] = mfma4(a[nu & 1][1][3], wc[nu][t][3], acc[nu][1][t]);
                    wc[nu][t] = *reinterpret_cast<const f32x4*>(wn + t * 1024 + wlane);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (nu == 0 && more1) {            // piece 2: transform arithmetic + V stores of chunk c + 1
                transform_write(vs0 + (b ^ 1) * V_FLOATS);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (nu == 1) {                     // piece 3: park chunk c + 2 (its loads were issued an iteration ago), request chunk c + 3
                if (more2) park(raw0 + b * RAW_FLOATS);
                load_in(ch + 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: nu half of A^T M A in registers, xi half after the exchange through LDS ----
    float* const rs = smem;                                      // [xi 4][j 2][tile 32][RS]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const f32x4 r0 = (acc[0][m][t] + acc[1][m][t]) + acc[2][m][t];
            const f32x4 r1 = (acc[1][m][t] - acc[2][m][t]) - acc[3][m][t];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tile = m * 16 + g * 4 + r;
                rs[((wave * 2 + 0) * 32 + tile) * RS + t * 16 + pl] = r0[r];
                rs[((wave * 2 + 1) * 32 + tile) * RS + t * 16 + pl] = r1[r];
            }
        }
    __syncthreads();
    constexpr int F4 = NTW * 4, PPP = 256 / F4, XS = 2 * 32 * RS;
    const int c4 = tid % F4;
    const int co = nt0 * 16 + c4 * 4;
    const OutSel o = OutSel{p.out, p.addend, p.pitch_out, p.pitch_add, (nt0 + c4 / 4) * 16};
    const int cw = (c4 & 3) * 4;
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + co);
    double sq8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};     // (sum, sum of squares) of this thread's 4 channels
    float sf[4] = {0.f, 0.f, 0.f, 0.f}, qf[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 128 / PPP; ++it) {
        const int pix = tid / F4 + PPP * it;
        const int oyl = pix >> 4, oxl = pix & 15;
        const int tile = (oyl >> 1) * 8 + (oxl >> 1), ii = oyl & 1, jj = oxl & 1;
        const float* rb = rs + (jj * 32 + tile) * RS + c4 * 4;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(rb + (ii ? 3 * XS : 0));
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(rb + XS);
        const f32x4 x2 = *reinterpret_cast<const f32x4*>(rb + 2 * XS);
        f32x4 y = ii ? (x1 - x2) - x0 : (x0 + x1) + x2;
        y += bias4;
        const int oy = oy0 + oyl, ox = ox0 + oxl;
        if (oy < p.hout && ox < p.wout) {
            const size_t gp = ((size_t)img * p.hout + oy) * p.wout + ox;
            if (o.add) y += *reinterpret_cast<const f32x4*>(o.add + gp * o.pitch_add + o.ch0 + cw);
            *reinterpret_cast<f32x4*>(o.out + gp * o.pitch + o.ch0 + cw) = y;
            if (p.stats) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    sf[k] += y[k];
                    qf[k] = __builtin_fmaf(y[k], y[k], qf[k]);
                }
            }
        }
        // at most 4 values per channel are summed in fp32 (relative error <= 2^-22 of a partial sum, below the rounding the
        // values themselves carry); everything from there on is fp64.  The fp64 conversions and FMAs of every single value were
        // 10 % of the 32-channel layers' time (the VALU work of a wave is not hidden behind another wave's fp32 MFMAs).
        if (p.stats && ((it & 3) == 3 || it == 128 / PPP - 1)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sq8[2 * k] += (double)sf[k];
                sq8[2 * k + 1] += (double)qf[k];
                sf[k] = 0.f;
                qf[k] = 0.f;
            }
        }
    }
    if (p.stats) {
        double* st_s = reinterpret_cast<double*>(smem + BODY_FLOATS);      // [4 waves][16*NTW][2]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int sh = F4; sh < 64; sh <<= 1) sq8[e] += __shfl_xor(sq8[e], sh);
        }
        if (lane < F4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) st_s[(wave * 16 * NTW + c4 * 4) * 2 + e] = sq8[e];
        }
        __syncthreads();
        if (tid < 32 * NTW) {
            const int c = tid >> 1, k = tid & 1;
            const double v = st_s[(0 * 16 * NTW + c) * 2 + k] + st_s[(1 * 16 * NTW + c) * 2 + k] + st_s[(2 * 16 * NTW + c) * 2 + k] +
                             st_s[(3 * 16 * NTW + c) * 2 + k];
            atomicAdd(p.stats + ((size_t)img * p.cout + nt0 * 16 + c) * 2 + k, v);
        }
    }
}

int g_wino = 2;   // the one A/B switch of the fp32 3x3 stride-1 layers (CS_WINO): 0 = direct kernels, 1 = Winograd F(2x2,3x3) everywhere (what a
                  // training forward runs), 2 (default) = F(4x4,3x3) where the layer has a kernel for it (inference), F(2x2) elsewhere
static const int g_wino_init = [] {
    if (const char* e = getenv("CS_WINO")) g_wino = atoi(e);
    return 0;
}();
bool wino_enabled() { return g_wino != 0; }
int wino_level() { return g_wino; }
// which F(4x4) kernel a layer takes: 0 = none (was the round-3 kernel: 16x16x4 MFMAs, a wave owns positions: the general form, channel counts that are
// not multiples of 32), 1 / 2 = the 32x32x2 forms of kernels_conv_wino4x.hip (64- / 32-channel blocks), 3 = its 16x16x4 form with the output
// transform in registers for the short K loops (cin <= 128: measured per layer, DESIGN 3.1a''; beyond, its half-size blocks are 4-6 % faster per
// launch too, but the slide loses what the window pass's two streams overlap: 35.3 vs 34.4 ms)
int wino4_form(int cin, int cout) {
    if (cin % 32 != 0) return 0;
    if (cout % 64 == 0) return cin <= 128 ? 3 : 1;
    return cout % 32 == 0 ? 2 : 0;
}
bool wino4_layout32(int cin, int cout) { const int f = wino4_form(cin, cout); return f == 1 || f == 2; }

size_t packed_wino_floats(int cout_total, int cin_total) { return (size_t)(cin_total / 16) * 16 * (cout_total / 16) * 256; }

int launch_conv_wino(hipStream_t s, const ConvSrc& a, const float* wpack_wino, const float* bias, float* out, int pitch_out, int cout,
                     double* stats, int n, int h, int w, const ConvOutExtra* ex) {
    CS_REQUIRE(a.c % 16 == 0 && a.c > 0 && cout % 32 == 0, "conv(winograd): channels (%d -> %d) unsupported", a.c, cout);
    CS_REQUIRE((a.pitch % 4) == 0 && (pitch_out % 4) == 0, "conv(winograd): pixel pitch must be a multiple of 4 floats");
    CS_REQUIRE(!a.h16 && !a.q8 && !a.ptr2, "conv(winograd): plain fp32 activations only");
    ConvParams p{};
    p.a = SrcDev{a.ptr, a.tab, a.slope, a.c, a.pitch, a.tab_c ? a.tab_c : a.c, a.tab_off};
    p.wpack = wpack_wino;
    p.bias = bias;
    p.out = out;
    p.stats = stats;
    p.pitch_out = pitch_out;
    p.cout = cout;
    p.ntile_total = cout / 16;
    p.n = n;
    p.hin = p.hout = h;
    p.win = p.wout = w;
    p.tiles_x = ceil_div(w, 16);
    p.tiles_y = ceil_div(h, 8);
    if (ex) {                                  // data-gradient launches: out = conv + addend (no (out, out2) split in this kernel)
        CS_REQUIRE(!ex->out2, "conv(winograd): the output split is not supported");
        p.addend = ex->addend;
        p.pitch_add = ex->pitch_add;
    }
    // the multiplies this algorithm performs on the matrix pipe: 16 per 2x2 output tile and (cin, cout) pair (a direct 3x3 needs 36)
    const double flops = 2.0 * n * (double)ceil_div(h, 2) * ceil_div(w, 2) * 16.0 * (double)cout * a.c;
    const double bytes = 4.0 * n * (double)h * w * (a.c + cout);
    ProfScope prof(K_CONV3_WINO, s, flops, bytes);
    const int nsp = p.tiles_x * p.tiles_y * n;
    CS_REQUIRE((size_t)h * w * a.pitch * 4 < ((size_t)1 << 32), "conv(winograd): one image's activation tensor must stay below 4 GB");
    const bool lazy = a.tab != nullptr;
#define CS_WINO_LAUNCH(NTW_, OCC_, LAZY_)                                                                                       \
    hipLaunchKernelGGL((conv_wino_kernel<NTW_, OCC_, LAZY_>), dim3(xcd_grid(nsp, p.ntile_total / NTW_)), dim3(256), 0, s, p)
    // 128-channel blocks at one wave per SIMD (conv_wino1_kernel) where the grid still fills the chip twice over: measured 2-8 %
    // faster than the 64-channel kernel on the 128...512-channel layers of the slide; small (training-sized) launches keep the
    // 64-channel kernel and its twice as many blocks.
    if (cout % 128 == 0 && (long long)xcd_grid(nsp, p.ntile_total / 8) >= 512) {
        const dim3 grid1(xcd_grid(nsp, p.ntile_total / 8));
        if (lazy) hipLaunchKernelGGL((conv_wino1_kernel<true>), grid1, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_wino1_kernel<false>), grid1, dim3(256), 0, s, p);
    } else if (cout % 64 == 0 && (long long)xcd_grid(nsp, p.ntile_total / 4) >= SMALL_GRID_BLOCKS) {     // (else: 32-channel blocks, twice the grid)
        if (lazy) CS_WINO_LAUNCH(4, 2, true);
        else CS_WINO_LAUNCH(4, 2, false);
    } else {
        if (lazy) CS_WINO_LAUNCH(2, 3, true);
        else CS_WINO_LAUNCH(2, 3, false);
    }
#undef CS_WINO_LAUNCH
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
