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

template <int NTW, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_wino_kernel(ConvParams p, long long* stamps) {
    constexpr int IN_ITERS = (WIH * WIW + 63) / 64;            // 3
    constexpr int RS = 16 * NTW + 4;                           // floats per (xi, j, tile) record of the epilogue exchange
    constexpr int R_FLOATS = 4 * 2 * 32 * RS;
    constexpr int MAIN_FLOATS = RAW_FLOATS + V_FLOATS;
    constexpr int BODY_FLOATS = MAIN_FLOATS > R_FLOATS ? MAIN_FLOATS : R_FLOATS;
    constexpr int ST_FLOATS = 4 * 16 * NTW * 2 * 2;            // fp64 statistics of the 4 waves
    __shared__ __attribute__((aligned(16))) float smem[BODY_FLOATS + ST_FLOATS];
    float* const raw = smem;
    float* const vs = smem + RAW_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, g = lane >> 4;
    // in-kernel stamps (diagnostics; null in production): cycles per phase of block 0, wave 0
    long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
    auto stamp = [&](int k) {
        if (stamps) {
            __builtin_amdgcn_sched_barrier(0);
            const long long t = (long long)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            tsum[k] += t - tlast;
            tlast = t;
        }
    };
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

    int goff[IN_ITERS];
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (tid >> 2) + 64 * it;
        const int iy = pix / WIW, ix = pix - iy * WIW;
        const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
        goff[it] = (pix < WIH * WIW && gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win) ? gy * p.win + gx : -1;
    }

    const int chunks = p.a.c >> 4;
    f32x4 rin[IN_ITERS];
    Lazy4 lz;
    auto load_in = [&](int ch) {
        const int c0 = ch << 4;
        lz.load(p.a, img, c0 + 4 * q, p.wpack);
        const float* base = p.a.ptr + (size_t)img * p.hin * p.win * p.a.pitch + c0 + 4 * q;
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {   // unconditional (clamped) loads; padding is zeroed at store time
            const int o = goff[it] >= 0 ? goff[it] : 0;
            rin[it] = *reinterpret_cast<const f32x4*>(base + (size_t)o * p.a.pitch);
        }
    };
    auto store_raw = [&]() {
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int pix = (tid >> 2) + 64 * it;
            if (pix < WIH * WIW) {
                const f32x4 v = goff[it] >= 0 ? lz.apply(rin[it]) : f32x4{0.f, 0.f, 0.f, 0.f};   // padding stays 0 after the activation
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
    const f32x4* const wbase = reinterpret_cast<const f32x4*>(p.wpack) + (size_t)nt0 * 64 + lane;
    const size_t wpos = (size_t)p.ntile_total * 64;
    f32x4 wc[2][NTW];
    const int last_step = chunks * 4 - 1;
    auto wload = [&](int step, f32x4 (&dst)[NTW]) {
        const f32x4* w = wbase + (size_t)((step >> 2) * 16 + wave * 4 + (step & 3)) * wpos;
#pragma unroll
        for (int t = 0; t < NTW; ++t) dst[t] = w[t * 64];
    };

    load_in(0);
    wload(0, wc[0]);
    store_raw();
    __syncthreads();
    transform();
    __syncthreads();
    stamp(0);                                  // [0] prologue
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
        stamp(1);                              // [1] loads issued + MFMA phase
        if (more) store_raw();
        stamp(2);                              // [2] norm + patch stores
        __syncthreads();
        stamp(3);                              // [3] barrier
        if (more) {
            transform();
            stamp(4);                          // [4] transform
            __syncthreads();
            stamp(5);                          // [5] barrier
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
    double ssum[4] = {0.0, 0.0, 0.0, 0.0}, qsum[4] = {0.0, 0.0, 0.0, 0.0};
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
                    ssum[k] += (double)y[k];
                    qsum[k] += (double)y[k] * (double)y[k];
                }
            }
        }
    }
    if (p.stats) {
        // lanes with equal c4 (lane % F4) -> the waves of the block (LDS) -> ONE fp64 atomic pair per block and channel
        double* st_s = reinterpret_cast<double*>(smem + BODY_FLOATS);      // [4 waves][16*NTW][2]
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int sh = F4; sh < 64; sh <<= 1) {
                ssum[k] += __shfl_xor(ssum[k], sh);
                qsum[k] += __shfl_xor(qsum[k], sh);
            }
        }
        if (lane < F4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                st_s[(wave * 16 * NTW + c4 * 4 + k) * 2] = ssum[k];
                st_s[(wave * 16 * NTW + c4 * 4 + k) * 2 + 1] = qsum[k];
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
    stamp(6);                                  // [6] epilogue
    if (stamps && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 8; ++k) stamps[wave * 8 + k] = tsum[k];
}


// ------------------------------------------------------------------------------------------------
// Persistent, role-split form of the same algorithm (the one that runs; the kernel above is kept for A/B, CS_WINO=1).
// ONE 512-thread block per CU walks its list of tiles; waves 0-3 ("matrix waves") issue nothing but MFMAs, their A-fragment
// reads and their B-fragment loads; waves 4-7 ("staging waves") do everything else, one 16-channel chunk ahead:
//   * staging wave w owns tile row w of the block tile: it loads its 4 x 18 input pixels of the next chunk (16-byte coalesced
//     loads issued one interval before they are needed), applies the producer's lazy norm + PReLU, parks them in a
//     wave-private LDS patch (no cross-wave synchronisation), transforms (tile, channel pair) patches and writes V into the
//     OTHER half of a double-buffered V - so there is ONE workgroup barrier per chunk and the matrix waves never wait for a
//     transform; the walk continues across tile boundaries, so a tile's prologue overlaps the previous tile's last chunk;
//   * at the end of a tile the matrix waves take the nu half of A^T M A in registers and drop R[xi][j] into LDS; the staging
//     waves finish the xi half, bias, fp64 statistics and the global stores during the next tile's first chunk.
// LDS: V 2 x 32 KB, patches 4 x 5.6 KB, R 68 KB (NTW = 4), statistics 4 KB = 158.5 KB of the CU's 160 KB.
// ------------------------------------------------------------------------------------------------
struct WinoTile {
    int img, oy0, ox0, nt0;
};
template <int Q>
struct IntC {
    static constexpr int value = Q;
};

template <int NTW>
__global__ __launch_bounds__(512, 2) void conv_wino_ps_kernel(ConvParams p, int vtotal, int dbg, long long* stamps) {
    // dbg (CS_WINO_DBG, diagnostics only; 0 in production): 1 = no staging, 2 = no combine, 4 = no MFMAs, 8 = no R hand-over,
    // 16 = matrix waves at raised priority, 32 = non-temporal weight loads
    constexpr int RS = 16 * NTW + 4;
    constexpr int R_FLOATS = 4 * 2 * 32 * RS;
    constexpr int RAWW = 72 * WPS;                              // floats of one staging wave's 4 x 18 pixel patch
    constexpr int ST_FLOATS = 4 * 16 * NTW * 2 * 2;
    __shared__ __attribute__((aligned(16))) float smem[2 * V_FLOATS + 4 * RAWW + R_FLOATS + ST_FLOATS];
    float* const vs = smem;
    float* const rs = smem + 2 * V_FLOATS + 4 * RAWW;             // [xi 4][j 2][tile 32][RS]
    double* const st_s = reinterpret_cast<double*>(rs + R_FLOATS);   // [4 staging waves][16*NTW][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_sp = p.tiles_x * p.tiles_y * p.n, n_ct = p.ntile_total / NTW;
    const int n_sg = (n_sp + 7) >> 3;
    const int chunks = p.a.c >> 4;                              // even (the launcher checks): the V / register-set parity is static
    const int gstep = gridDim.x;
    // in-kernel stamps (diagnostics: tests/diagnostics/wino_layer_bench; `stamps` is null in production): cycles per phase summed
    // over the tile walk of block 0, one row of 8 counters per wave
    long long tsum[6] = {0, 0, 0, 0, 0, 0};
    long long tlast = stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
    auto stamp = [&](int k) {
        if (stamps) {
            __builtin_amdgcn_sched_barrier(0);
            const long long t = (long long)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            tsum[k] += t - tlast;
            tlast = t;
        }
    };
    auto stamps_out = [&]() {
        if (stamps && blockIdx.x == 0 && lane == 0)
            for (int k = 0; k < 6; ++k) stamps[wave * 8 + k] = tsum[k];
    };

    // The tile walk, identical in both roles: virtual ids blockIdx.x + k * gridDim.x.  The grid is a multiple of 8, so a block
    // keeps (v & 7) - its XCD under round-robin dispatch - for its whole walk.  Within an XCD the walk is CHANNEL-TILE MAJOR:
    // all of the XCD's blocks work on the same output-channel tile at the same time and stream the same transformed weights
    // (one L2-resident slice per XCD instead of n_ct slices competing for the 4 MB).
    auto next_tile = [&](int& v, WinoTile& t) -> bool {
        for (; v < vtotal; v += gstep) {
            const int lane8 = v & 7, r = v >> 3;
            const int ct = r / n_sg, sg = r - ct * n_sg;
            int sp = sg * 8 + lane8;
            if (sp >= n_sp) continue;
            const int tx = sp % p.tiles_x;
            sp /= p.tiles_x;
            const int ty = sp % p.tiles_y;
            t.img = sp / p.tiles_y;
            t.oy0 = ty * 8;
            t.ox0 = tx * 16;
            t.nt0 = ct * NTW;
            return true;
        }
        return false;
    };

    if (wave < 4) {
        // ============================== matrix waves ==============================
        if (dbg & 16) __builtin_amdgcn_s_setprio(3);
        const bool nt_loads = (dbg & 32) != 0;
        const int pl = lane & 15, g = lane >> 4;
        f32x4 acc[4][2][NTW];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[nu][m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4* const wbase = reinterpret_cast<const f32x4*>(p.wpack) + lane;
        const size_t wpos = (size_t)p.ntile_total * 64;
        auto wptr = [&](int nt0, int ch) { return wbase + (size_t)(ch * 16 + wave * 4) * wpos + (size_t)nt0 * 64; };
        const float* const a_base = vs + ((wave * 16 + g) * 32 + pl) * 4;      // + q * V_FLOATS + nu * 512 + m * 64

        int v = blockIdx.x;
        WinoTile cur, nxt;
        bool have = next_tile(v, cur);
        f32x4 wc[4][NTW];                                         // B fragments of one chunk, refilled one CHUNK ahead
        {
            const f32x4* w0 = wptr(have ? cur.nt0 : 0, 0);
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                for (int t = 0; t < NTW; ++t) wc[nu][t] = w0[(size_t)nu * wpos + t * 64];
        }
        __syncthreads();                                          // V(first chunk) is ready
        // one interval = one 16-channel chunk out of V[Q]; wnext = the fragments of the NEXT interval
        auto interval = [&](auto Q, const f32x4* wnext, bool last) {
            constexpr int q = decltype(Q)::value;
            const float* const a_src = a_base + q * V_FLOATS;
            stamp(0);                                             // [0] barrier wait + loop overhead
            f32x4 a[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[0][m] = *reinterpret_cast<const f32x4*>(a_src + m * 64);
            if (!(dbg & 4))
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                if (nu < 3) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) a[(nu + 1) & 1][m] = *reinterpret_cast<const f32x4*>(a_src + (nu + 1) * 512 + m * 64);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int t = 0; t < NTW; ++t) acc[nu][m][t] = mfma4(a[nu & 1][m][j], wc[nu][t][j], acc[nu][m][t]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NTW; ++t)
                    wc[nu][t] = nt_loads ? __builtin_nontemporal_load(wnext + (size_t)nu * wpos + t * 64) : wnext[(size_t)nu * wpos + t * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp(1);                                             // [1] MFMA phase
            if (last && !(dbg & 8)) {
                // nu half of the output transform in registers, R[xi][j] to LDS, accumulators cleared for the next tile
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
#pragma unroll
                        for (int nu = 0; nu < 4; ++nu) acc[nu][m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
            }
            stamp(2);                                             // [2] R hand-over
            __syncthreads();
        };
        while (have) {
            int vn = v + gstep;
            const bool have_n = next_tile(vn, nxt);
            for (int ch = 0; ch < chunks; ch += 2) {
                interval(IntC<0>(), wptr(cur.nt0, ch + 1), false);
                // the very last interval of the walk re-reads its own fragments (unconditional loads)
                const bool last = ch + 2 == chunks;
                interval(IntC<1>(), last ? (have_n ? wptr(nxt.nt0, 0) : wptr(cur.nt0, ch + 1)) : wptr(cur.nt0, ch + 2), last);
            }
            cur = nxt;
            v = vn;
            have = have_n;
        }
        __syncthreads();                                          // staging waves: last combine
        __syncthreads();                                          // staging waves: last statistics flush
        stamps_out();
        return;
    }

    // ============================== staging waves ==============================
    const int pw = wave - 4, ptid = tid - 256;
    float* const raw_w = smem + 2 * V_FLOATS + pw * RAWW;
    // per-lane geometry of the wave's 72-pixel x 4-quad patch: item = lane + 64 * it -> (pixel, quad)
    constexpr int L_ITERS = 5;
    int l_iy[L_ITERS], l_ix[L_ITERS];
    const int lq = lane & 3;
#pragma unroll
    for (int it = 0; it < L_ITERS; ++it) {
        const int pix = (lane >> 2) + 16 * it;
        l_iy[it] = pix / WIW;
        l_ix[it] = pix - l_iy[it] * WIW;
    }
    const int t_tx = lane & 7, t_h = (lane >> 3) & 1, t_g = lane >> 4;
    const float* const t_src = raw_w + (2 * t_tx) * WPS + 4 * t_g + 2 * t_h;
    float* const t_dst0 = vs + (t_g * 32 + pw * 8 + t_tx) * 4 + 2 * t_h;

    // loader: chunk i of the walk lands in register set i & 1, TWO intervals before it is staged (the weight streams of 256 CUs
    // keep the memory system busy enough that a load issued one interval ahead arrived thousands of cycles late).  Loads are
    // UNCONDITIONAL - past the end of the walk the loader re-reads its last chunk - so that hipcc can count the younger loads
    // and wait with vmcnt(8) instead of vmcnt(0).
    int vl = blockIdx.x;
    WinoTile tl;
    bool have_l = next_tile(vl, tl);
    const bool any_tile = have_l;
    if (!any_tile) tl = WinoTile{0, 0, 0, 0};
    int ch_l = 0;
    f32x4 rin[2][L_ITERS];
    Lazy4 lz[2];
    unsigned vmask[2] = {0, 0};
    auto issue_load = [&](auto Q) {
        constexpr int q = decltype(Q)::value;
        const int c0 = ch_l << 4;
        lz[q].load(p.a, tl.img, c0 + 4 * lq, p.wpack);
        const float* base = p.a.ptr + (size_t)tl.img * p.hin * p.win * p.a.pitch + c0 + 4 * lq;
        unsigned vm = 0;
#pragma unroll
        for (int it = 0; it < L_ITERS; ++it) {
            const int gy = tl.oy0 - 1 + 2 * pw + l_iy[it], gx = tl.ox0 - 1 + l_ix[it];
            const bool ok = gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win && (it < L_ITERS - 1 || lane < 32);
            vm |= ok ? (1u << it) : 0u;
            const int o = ok ? gy * p.win + gx : 0;
            rin[q][it] = *reinterpret_cast<const f32x4*>(base + (size_t)o * p.a.pitch);
        }
        vmask[q] = vm;
        if (have_l) {                                             // advance; at the end of the walk stay on the last chunk
            if (ch_l + 1 < chunks) {
                ++ch_l;
            } else {
                int v2 = vl + gstep;
                WinoTile t2;
                if (next_tile(v2, t2)) {
                    vl = v2;
                    tl = t2;
                    ch_l = 0;
                } else {
                    have_l = false;
                }
            }
        }
    };
    // register set Q -> wave-private patch (the set is then refilled with chunk i + 2, unconditionally) -> V[Q]
    auto park = [&](auto Q) {
        constexpr int q = decltype(Q)::value;
#pragma unroll
        for (int it = 0; it < L_ITERS; ++it) {
            if (it < L_ITERS - 1 || lane < 32) {
                const f32x4 vv = (vmask[q] >> it) & 1u ? lz[q].apply(rin[q][it]) : f32x4{0.f, 0.f, 0.f, 0.f};   // padding stays 0 after the activation
                *reinterpret_cast<f32x4*>(&raw_w[((lane >> 2) + 16 * it) * WPS + 4 * lq]) = vv;
            }
        }
        stamp(1);                             // [1] wait for the chunk's loads + norm + patch stores
        issue_load(Q);
        stamp(2);                             // [2] loads of chunk i + 2 issued
    };
    auto stage = [&](auto Q) {
        constexpr int q = decltype(Q)::value;
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the patch is wave-private, no barrier needed
        __builtin_amdgcn_wave_barrier();
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
        float* const t_dst = t_dst0 + q * V_FLOATS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float2*>(t_dst + (i * 4 + 0) * 512) = make_float2(t[i][0].x - t[i][2].x, t[i][0].y - t[i][2].y);
            *reinterpret_cast<float2*>(t_dst + (i * 4 + 1) * 512) = make_float2(t[i][1].x + t[i][2].x, t[i][1].y + t[i][2].y);
            *reinterpret_cast<float2*>(t_dst + (i * 4 + 2) * 512) = make_float2(t[i][2].x - t[i][1].x, t[i][2].y - t[i][1].y);
            *reinterpret_cast<float2*>(t_dst + (i * 4 + 3) * 512) = make_float2(t[i][1].x - t[i][3].x, t[i][1].y - t[i][3].y);
        }
        __builtin_amdgcn_wave_barrier();      // the wave's next patch stores stay behind these reads (same wave: LDS ops are in order)
        stamp(3);                             // [3] transform
    };
    // xi half of the output transform + bias + statistics + stores of a finished tile: thread = (pixel, channel quad)
    constexpr int F4 = NTW * 4, PPP = 256 / F4, XS = 2 * 32 * RS;
    const int c4 = ptid % F4;
    // the bias of a tile is fetched when the tile STARTS: by the time the combine needs it the load is older than the two chunk
    // loads in flight (a load issued inside the combine would be the youngest, and waiting for it would drain the prefetches)
    auto load_bias = [&](const WinoTile& tt) { return *reinterpret_cast<const f32x4*>((p.bias ? p.bias : p.wpack) + tt.nt0 * 16 + c4 * 4); };
    auto combine = [&](const WinoTile& tt, f32x4 bias4) {
        const OutSel o = OutSel{p.out, p.addend, p.pitch_out, p.pitch_add, (tt.nt0 + c4 / 4) * 16};
        const int cw = (c4 & 3) * 4;
        if (!p.bias) bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
        double ssum[4] = {0.0, 0.0, 0.0, 0.0}, qsum[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int it = 0; it < 128 / PPP; ++it) {
            const int pix = ptid / F4 + PPP * it;
            const int oyl = pix >> 4, oxl = pix & 15;
            const int tile = (oyl >> 1) * 8 + (oxl >> 1), ii = oyl & 1, jj = oxl & 1;
            const float* rb = rs + (jj * 32 + tile) * RS + c4 * 4;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(rb + (ii ? 3 * XS : 0));
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(rb + XS);
            const f32x4 x2 = *reinterpret_cast<const f32x4*>(rb + 2 * XS);
            f32x4 y = ii ? (x1 - x2) - x0 : (x0 + x1) + x2;
            y += bias4;
            const int oy = tt.oy0 + oyl, ox = tt.ox0 + oxl;
            if (oy < p.hout && ox < p.wout) {
                const size_t gp = ((size_t)tt.img * p.hout + oy) * p.wout + ox;
                if (o.add) y += *reinterpret_cast<const f32x4*>(o.add + gp * o.pitch_add + o.ch0 + cw);
                *reinterpret_cast<f32x4*>(o.out + gp * o.pitch + o.ch0 + cw) = y;
                if (p.stats) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        ssum[k] += (double)y[k];
                        qsum[k] += (double)y[k] * (double)y[k];
                    }
                }
            }
        }
        if (p.stats) {      // lanes with equal c4 -> this wave's partial in LDS; the block's sum is flushed after the next barrier
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int sh = F4; sh < 64; sh <<= 1) {
                    ssum[k] += __shfl_xor(ssum[k], sh);
                    qsum[k] += __shfl_xor(qsum[k], sh);
                }
            }
            if (lane < F4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    st_s[(pw * 16 * NTW + c4 * 4 + k) * 2] = ssum[k];
                    st_s[(pw * 16 * NTW + c4 * 4 + k) * 2 + 1] = qsum[k];
                }
            }
        }
    };
    auto flush = [&](const WinoTile& tt) {   // ONE fp64 atomic pair per tile and channel
        if (p.stats && ptid < 32 * NTW) {
            const int c = ptid >> 1, k = ptid & 1;
            const double vsum = st_s[(0 * 16 * NTW + c) * 2 + k] + st_s[(1 * 16 * NTW + c) * 2 + k] + st_s[(2 * 16 * NTW + c) * 2 + k] +
                                st_s[(3 * 16 * NTW + c) * 2 + k];
            atomicAdd(p.stats + ((size_t)tt.img * p.cout + tt.nt0 * 16 + c) * 2 + k, vsum);
        }
    };

    issue_load(IntC<0>());                    // chunk 0 -> set 0
    issue_load(IntC<1>());                    // chunk 1 -> set 1
    park(IntC<0>());                          // chunk 0 -> patch; set 0 <- chunk 2
    if (any_tile && !(dbg & 1)) stage(IntC<0>());   // patch -> V[0]
    __syncthreads();
    {
        int v = blockIdx.x;
        WinoTile cur, nxt, pend_t, fl_t;
        bool have = next_tile(v, cur);
        bool pend = false, fl = false;
        f32x4 bias_cur = f32x4{0.f, 0.f, 0.f, 0.f}, bias_pend = bias_cur;
        // one interval: the matrix waves run chunk i out of V[i & 1]; chunk i + 1 is staged into V[Q] (Q = (i + 1) & 1)
        auto interval = [&](auto Q, bool more) {
            stamp(0);                                             // [0] barrier wait
            if (fl) {
                if (!(dbg & 2)) flush(fl_t);
                fl = false;
            }
            park(Q);
            if (more && !(dbg & 1)) stage(Q);
            if (pend) {
                if (!(dbg & 2)) combine(pend_t, bias_pend);
                pend = false;
                fl = true;
                fl_t = pend_t;
                stamp(4);                                         // [4] combine
            }
            __syncthreads();
        };
        while (have) {
            int vn = v + gstep;
            const bool have_n = next_tile(vn, nxt);
            bias_cur = load_bias(cur);
            for (int ch = 0; ch < chunks; ch += 2) {
                interval(IntC<1>(), true);
                interval(IntC<0>(), ch + 2 < chunks || have_n);
            }
            pend = true;                                          // R(cur) is complete behind the barrier that closed its last chunk
            pend_t = cur;
            bias_pend = bias_cur;
            cur = nxt;
            v = vn;
            have = have_n;
        }
        if (fl && !(dbg & 2)) flush(fl_t);
        if (pend && !(dbg & 2)) combine(pend_t, bias_pend);
        __syncthreads();
        if (pend && !(dbg & 2)) flush(pend_t);
        __syncthreads();
        stamps_out();
    }
}

long long* g_wino_stamps = nullptr;    // diagnostics: device buffer of 8 x 8 counters (tests/diagnostics/wino_layer_bench.hip)
void wino_set_stamp_buffer(long long* dev) { g_wino_stamps = dev; }

int g_wino = 2;   // A/B switch (CS_WINO): 0 = direct kernels, 1 = one-tile-per-block Winograd kernel, 2 = persistent role-split kernel
static const int g_wino_init = [] {
    if (const char* e = getenv("CS_WINO")) g_wino = atoi(e);
    return 0;
}();
bool wino_enabled() { return g_wino != 0; }

size_t packed_wino_floats(int cout_total, int cin_total) { return (size_t)(cin_total / 16) * 16 * (cout_total / 16) * 256; }

int launch_conv_wino(hipStream_t s, const ConvSrc& a, const float* wpack_wino, const float* bias, float* out, int pitch_out, int cout,
                     double* stats, int n, int h, int w) {
    CS_REQUIRE(a.c % 16 == 0 && a.c > 0 && cout % 32 == 0, "conv(winograd): channels (%d -> %d) unsupported", a.c, cout);
    CS_REQUIRE((a.pitch % 4) == 0 && (pitch_out % 4) == 0, "conv(winograd): pixel pitch must be a multiple of 4 floats");
    CS_REQUIRE(!a.h16 && !a.q8, "conv(winograd): fp32 activations only");
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
    // the multiplies this algorithm performs on the matrix pipe: 16 per 2x2 output tile and (cin, cout) pair (a direct 3x3 needs 36)
    const double flops = 2.0 * n * (double)ceil_div(h, 2) * ceil_div(w, 2) * 16.0 * (double)cout * a.c;
    const double bytes = 4.0 * n * (double)h * w * (a.c + cout);
    ProfScope prof(K_CONV3_WINO, s, flops, bytes);
    const int nsp = p.tiles_x * p.tiles_y * n;
    if (g_wino >= 2 && a.c % 32 == 0) {
        // persistent form: one 512-thread block per CU (the kernel needs a whole CU's LDS); the grid is a multiple of 8 so that a
        // block keeps its XCD over its whole tile walk
        static int n_cu = 0;
        if (n_cu == 0) {
            int dev = 0, cus = 0;
            CS_HIP(hipGetDevice(&dev));
            CS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            n_cu = cus > 8 ? cus / 8 * 8 : 8;
        }
        const int nt = cout % 64 == 0 ? 4 : 2;
        const int vtotal = (int)xcd_grid(nsp, p.ntile_total / nt);
        const int grid = vtotal < n_cu ? vtotal : n_cu;
        static const int dbg = getenv("CS_WINO_DBG") ? atoi(getenv("CS_WINO_DBG")) : 0;
        if (nt == 4) hipLaunchKernelGGL((conv_wino_ps_kernel<4>), dim3(grid), dim3(512), 0, s, p, vtotal, dbg, g_wino_stamps);
        else hipLaunchKernelGGL((conv_wino_ps_kernel<2>), dim3(grid), dim3(512), 0, s, p, vtotal, dbg, g_wino_stamps);
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
    if (cout % 64 == 0) {
        hipLaunchKernelGGL((conv_wino_kernel<4, 2>), dim3(xcd_grid(nsp, p.ntile_total / 4)), dim3(256), 0, s, p, g_wino_stamps);
    } else {
        hipLaunchKernelGGL((conv_wino_kernel<2, 3>), dim3(xcd_grid(nsp, p.ntile_total / 2)), dim3(256), 0, s, p, g_wino_stamps);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
