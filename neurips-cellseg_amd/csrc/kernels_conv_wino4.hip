// gfx950 (MI355X): the WIDE 3x3 stride-1 convolutions (>= 128 input channels) in Winograd form F(4x4, 3x3) on the fp32 matrix pipe.
//
// Replaces the same torch conv2d dispatches as kernels_conv_wino.hip (reference call sites baseline/predict.py:109 -> :45-52: the
// stride-1 subunits of the deep ResidualUnits, 60 % of the Winograd FLOPs of the 32-base net).  Y = A^T [ (G g G^T) . (B^T d B) ] A
// with a 6x6 input patch per 4x4 output tile: 36 multiplies per 16 outputs = 2.25 per output pixel and (cin, cout) pair, where
// F(2x2, 3x3) needs 4 and a direct convolution 9.  All arithmetic is fp32 (v_mfma_f32_16x16x4_f32 + fp32 VALU transforms, weights
// transformed in fp64 and rounded once at pack time).  The larger transform constants cost precision: measured with
// tests/diagnostics/winograd_f4_precision.py (F(4x4) on every layer with >= 128 input channels, the rest F(2x2)) the logits land
// 2.1e-5 ... 3.8e-5 from an fp64 evaluation (direct fp32: 5e-6 ... 8e-6) - inside the 1e-4 budget of the parity tests, which is why
// only the wide layers, where the matrix pipe is the bound, take this form.
//
// Shape (one block = 256 threads = 4 waves, 16 x 16 output pixels = 16 tiles = ONE MFMA M tile, 64 output channels):
//   * K loop over 16-channel chunks: the halo'd 18 x 18 patch goes global -> registers one chunk ahead (lazy InstanceNorm + PReLU on
//     the way, zero padding after the activation) -> LDS `raw` [324 px][20]; thread (tile, channel) transforms its 6 x 6 patch
//     (B^T d B: 12 six-point transforms of 14 operations) into V [36 positions][4 k-groups][16 tiles][4], the A-fragment order.
//   * wave w owns positions 9 w ... 9 w + 8 for all 16 tiles and the 4 channel tiles: per chunk 9 A fragments (ds_read_b128),
//     36 B fragments L2 -> registers one position ahead, 144 MFMAs, 144 accumulator registers.
//   * epilogue: the 36 x 16 x 64 products cross the waves through LDS in two 32-channel halves [36][16 tiles][32]; thread (tile,
//     channel quad) applies A^T M A (10 six-point transforms of 10 operations on float4), bias, fp64 statistics, 16-byte stores.
#include <type_traits>

#include "conv_common.h"
#include "pack_jobs.h"

namespace cs {

namespace {
constexpr int W4_IH = 18, W4_IW = 18;              // input patch of a block
constexpr int W4_PS = 20;                          // floats per staged raw pixel (16 + 4: the transform's reads of two tiles of a row spread over the banks)
constexpr int RAW4_FLOATS = W4_IH * W4_IW * W4_PS; // 6480
constexpr int V4_FLOATS = 36 * 4 * 16 * 4;         // 9216
constexpr int M4_FLOATS = 36 * 16 * 32;            // 18432: one 32-channel half of the products

// y = B^T x for one 6-vector (F(4x4,3x3) input transform, Lavin & Gray)
__device__ __forceinline__ void bt6(const float (&x)[6], float (&y)[6]) {
    const float a = __builtin_fmaf(-4.f, x[2], x[4]), b = __builtin_fmaf(-4.f, x[1], x[3]);
    const float c = x[4] - x[2], d = 2.f * (x[3] - x[1]);
    y[0] = __builtin_fmaf(4.f, x[0], __builtin_fmaf(-5.f, x[2], x[4]));
    y[1] = a + b;
    y[2] = a - b;
    y[3] = c + d;
    y[4] = c - d;
    y[5] = __builtin_fmaf(4.f, x[1], __builtin_fmaf(-5.f, x[3], x[5]));
}
// o = A^T m for one 6-vector of float4 (output transform)
__device__ __forceinline__ void at6(const f32x4 (&m)[6], f32x4 (&o)[4]) {
    const f32x4 s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
    o[0] = (m[0] + s1) + s2;
    o[1] = d1 + 2.f * d2;
    o[2] = s1 + 4.f * s2;
    o[3] = (d1 + 8.f * d2) + m[5];
}
}  // namespace

// NTW4: channel tiles per block - 4 (64 output channels) or 2 (the 32-channel layers)
template <bool LAZY, int NTW4>
__global__ __launch_bounds__(256, 2) void conv_wino4_kernel(ConvParams p) {
    constexpr int IN_ITERS = (W4_IH * W4_IW * 4 + 255) / 256;                 // 6: (pixel, channel quad) items per thread
    constexpr int MAIN_FLOATS = RAW4_FLOATS + V4_FLOATS;
    constexpr int BODY_FLOATS = MAIN_FLOATS > M4_FLOATS ? MAIN_FLOATS : M4_FLOATS;
    constexpr int ST_FLOATS = 4 * 32 * 2 * 2;                                 // fp64 statistics of the four waves, 32 channels of a half
    __shared__ __attribute__((aligned(16))) float smem[BODY_FLOATS + ST_FLOATS];
    float* const raw = smem;
    float* const vs = smem + RAW4_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const WorkId wk = decode_work(p.tiles_x * p.tiles_y * p.n, p.ntile_total / NTW4);
    if (!wk.valid) return;
    int bid = wk.spatial;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int oy0 = ty * 16, ox0 = tx * 16;
    const int nt0 = wk.ct * NTW4;
    const int q = tid & 3;

    f32x4 acc[9][NTW4];
#pragma unroll
    for (int ps = 0; ps < 9; ++ps)
#pragma unroll
        for (int t = 0; t < NTW4; ++t) acc[ps][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging geometry, fixed for the K loop: byte offset of the pixel's channel quad inside the image's tensor + a padding bit
    unsigned boff[IN_ITERS], inb = 0;
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (tid >> 2) + 64 * it;
        const int iy = pix / W4_IW, ix = pix - iy * W4_IW;
        const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
        const bool ok = pix < W4_IH * W4_IW && gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win;
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
        for (int it = 0; it < IN_ITERS; ++it) rin[it] = *reinterpret_cast<const f32x4*>(base + boff[it]);   // unconditional (clamped)
    };
    auto store_raw = [&]() {
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int pix = (tid >> 2) + 64 * it;
            if (pix < W4_IH * W4_IW) {
                const f32x4 a = LAZY ? lz.apply_on(rin[it]) : rin[it];
                const f32x4 v = (inb >> it) & 1u ? a : f32x4{0.f, 0.f, 0.f, 0.f};   // padding stays 0 after the activation
                *reinterpret_cast<f32x4*>(&raw[pix * W4_PS + 4 * q]) = v;
            }
        }
    };
    // input transform of one (tile, channel): thread = (tile = tid >> 4, channel = tid & 15)
    const int t_tile = tid >> 4, t_c = tid & 15;
    const float* const t_src = raw + (((t_tile >> 2) * 4) * W4_IW + (t_tile & 3) * 4) * W4_PS + t_c;
    // V[position][k-group g][tile ^ 2 g][4]: the XOR spreads the four k-groups a 32-lane store group holds over all 32 banks (without it
    // the 36 stores of a thread were 4-way conflicted); the A-fragment read permutes its tiles the same way and stays conflict-free
    float* const t_dst = vs + (((t_c >> 2) * 16 + (t_tile ^ (2 * (t_c >> 2)))) * 4) + (t_c & 3);          // + position * 256
    auto transform = [&]() {
        float t[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {                       // columns: t[:, j] = B^T d[:, j] (a column is read when it is needed: the
            float x[6], y[6];                               // 144 accumulator registers are live across this phase)
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] = t_src[(i * W4_IW + j) * W4_PS];
            bt6(x, y);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][j] = y[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {                       // rows: V[i, :] = B^T t[i, :]
            float y[6];
            bt6(t[i], y);
#pragma unroll
            for (int j = 0; j < 6; ++j) t_dst[(i * 6 + j) * 256] = y[j];
        }
    };

    // B fragments: packed[chunk][position 36][ntile_total][64 lanes][4]; this wave reads positions 9 * wave + ps
    const char* const wbase = reinterpret_cast<const char*>(p.wpack) + ((size_t)nt0 * 64 + (size_t)wave * 9 * p.ntile_total * 64) * 16;
    const size_t wpos = (size_t)p.ntile_total * 64 * 16;              // bytes between positions
    const unsigned wlane = (unsigned)lane * 16u;
    f32x4 wc[2][NTW4];
    const int last_step = chunks * 9 - 1;
    auto wload = [&](int step, f32x4 (&dst)[NTW4]) {
        const int ch = step / 9, ps = step - ch * 9;
        const char* w = wbase + (size_t)(ch * 36 + ps) * wpos;
#pragma unroll
        for (int t = 0; t < NTW4; ++t) dst[t] = *reinterpret_cast<const f32x4*>(w + t * 1024 + wlane);
    };

    load_in(0);
    wload(0, wc[0]);
    store_raw();
    __syncthreads();
    transform();
    __syncthreads();
    const float* const a_src = vs + ((wave * 9 * 4 + g) * 16 + (pl ^ (2 * g))) * 4;       // + ps * 256
    // one chunk: 9 positions x 16 MFMAs.  A chunk has an ODD number of weight-ring steps, so the ring slot of a position alternates
    // from chunk to chunk: the K loop is unrolled by two (PAR = parity of the chunk) to keep the slot a compile-time constant
    auto do_chunk = [&](int ch, auto par) {
        constexpr int PAR = decltype(par)::value;
        const bool more = ch + 1 < chunks;
        load_in(more ? ch + 1 : ch);           // unconditional: a skipped load group would mis-size the weight ring's waits
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ps = 0; ps < 9; ++ps) {
            {
                const int step = ch * 9 + ps + 1;
                wload(step < last_step ? step : last_step, wc[(ps + PAR + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 a = *reinterpret_cast<const f32x4*>(a_src + ps * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NTW4; ++t) acc[ps][t] = mfma4(a[j], wc[(ps + PAR) & 1][t][j], acc[ps][t]);
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
    };
    for (int ch = 0; ch < chunks; ch += 2) {   // (the launcher requires an even number of chunks: cin % 32 == 0)
        do_chunk(ch, std::integral_constant<int, 0>());
        do_chunk(ch + 1, std::integral_constant<int, 1>());
    }

    // ---- epilogue: two 32-channel halves through LDS ----
    float* const ms = smem;                                     // [36 positions][16 tiles][32 channels], channel index XOR 16 for tiles 4..7, 12..15
    double* const st_s = reinterpret_cast<double*>(smem + BODY_FLOATS);      // [4 waves][32 channels][2]
    const int e_grp = tid >> 7;                                 // output rows 2 e_grp, 2 e_grp + 1 of the tile
    const int e_tile = (tid & 127) >> 3, e_q = tid & 7;         // output thread: tile, channel quad of the half
    const int e_ty = e_tile >> 2, e_tx = e_tile & 3;
    const int e_sw = ((e_tile >> 2) & 1) << 4;
#pragma unroll
    for (int hf = 0; hf < NTW4 / 2; ++hf) {
        // C layout of an accumulator tile: row (tile) = 4 g + r, column (channel) = pl
#pragma unroll
        for (int ps = 0; ps < 9; ++ps)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tile = 4 * g + r;                  // (tile >> 2) & 1 == g & 1
                    ms[((wave * 9 + ps) * 16 + tile) * 32 + ((tt * 16 + pl) ^ ((g & 1) << 4))] = acc[ps][2 * hf + tt][r];
                }
        __syncthreads();
        {
            const int co = nt0 * 16 + hf * 32 + 4 * e_q;
            f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + co);
            const float* src = ms + e_tile * 32 + ((4 * e_q) ^ e_sw);
            float sf[4] = {0.f, 0.f, 0.f, 0.f}, qf[4] = {0.f, 0.f, 0.f, 0.f};
            const int ox = ox0 + 4 * e_tx;
            // a thread takes two of the tile's four output rows (both halves of the block read all 36 products: holding all four rows
            // of A^T M next to the second half's 72 accumulators spilled 48 registers, and 128 idle threads doubled the epilogue)
            f32x4 tm[2][6];                                     // rows 2 e_grp, 2 e_grp + 1 of A^T M: 6 position columns each
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                f32x4 m[6];
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) m[xi] = *reinterpret_cast<const f32x4*>(src + (xi * 6 + nu) * 512);
                f32x4 o[4];
                at6(m, o);
                tm[0][nu] = e_grp ? o[2] : o[0];
                tm[1][nu] = e_grp ? o[3] : o[1];
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                f32x4 y[4];
                at6(tm[ii], y);
                const int oy = oy0 + 4 * e_ty + 2 * e_grp + ii;
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    f32x4 v = y[jx] + bias4;
                    if (oy < p.hout && ox + jx < p.wout) {
                        const size_t gp = ((size_t)img * p.hout + oy) * p.wout + ox + jx;
                        if (p.addend) v += *reinterpret_cast<const f32x4*>(p.addend + gp * p.pitch_add + co);
                        *reinterpret_cast<f32x4*>(p.out + gp * p.pitch_out + co) = v;
                        if (p.stats) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                sf[k] += v[k];
                                qf[k] = __builtin_fmaf(v[k], v[k], qf[k]);
                            }
                        }
                    }
                }
            }
            if (p.stats) {
                // a thread's 8 values per channel and the 8 lanes (tiles) of the wave that hold the same channel quad are summed in
                // fp32 (<= 64 values per partial sum), everything from there on is fp64 (as in kernels_conv_wino.hip)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int sh = 8; sh < 64; sh <<= 1) {
                        sf[k] += __shfl_xor(sf[k], sh);
                        qf[k] += __shfl_xor(qf[k], sh);
                    }
                }
                if (lane < 8) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        st_s[(wave * 32 + 4 * e_q + k) * 2] = (double)sf[k];
                        st_s[(wave * 32 + 4 * e_q + k) * 2 + 1] = (double)qf[k];
                    }
                }
            }
        }
        __syncthreads();
        if (p.stats && tid < 64) {
            const int c = tid >> 1, k = tid & 1;
            atomicAdd(p.stats + ((size_t)img * p.cout + nt0 * 16 + hf * 32 + c) * 2 + k,
                      (st_s[(0 * 32 + c) * 2 + k] + st_s[(1 * 32 + c) * 2 + k]) + (st_s[(2 * 32 + c) * 2 + k] + st_s[(3 * 32 + c) * 2 + k]));
        }
    }
}

size_t packed_wino4_floats(int cout_total, int cin_total) { return (size_t)(cin_total / 16) * 36 * (cout_total / 16) * 256; }

bool wino4_fits(int cin, int cout, int h, int w) { return cin >= WINO4_MIN_CIN && cin % 32 == 0 && cout % 32 == 0 && h >= 8 && w >= 8; }

int launch_conv_wino4(hipStream_t s, const ConvSrc& a, const float* wpack_wino4, const float* bias, float* out, int pitch_out, int cout,
                      double* stats, int n, int h, int w, const ConvOutExtra* ex) {
    if (wino4_form(a.c, cout) != 0) return launch_conv_wino4x(s, a, wpack_wino4, bias, out, pitch_out, cout, stats, n, h, w, ex);   // (weights packed to match)
    CS_REQUIRE(a.c % 32 == 0 && a.c > 0 && cout % 32 == 0, "conv(winograd 4x4): channels (%d -> %d) unsupported", a.c, cout);
    CS_REQUIRE((a.pitch % 4) == 0 && (pitch_out % 4) == 0, "conv(winograd 4x4): pixel pitch must be a multiple of 4 floats");
    CS_REQUIRE(!a.h16 && !a.q8 && !a.ptr2, "conv(winograd 4x4): plain fp32 activations only");
    CS_REQUIRE((size_t)h * w * a.pitch * 4 < ((size_t)1 << 32), "conv(winograd 4x4): one image's activation tensor must stay below 4 GB");
    ConvParams p{};
    p.a = SrcDev{a.ptr, a.tab, a.slope, a.c, a.pitch, a.tab_c ? a.tab_c : a.c, a.tab_off};
    p.wpack = wpack_wino4;
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
    p.tiles_y = ceil_div(h, 16);
    if (ex) {
        CS_REQUIRE(!ex->out2, "conv(winograd 4x4): the output split is not supported");
        p.addend = ex->addend;
        p.pitch_add = ex->pitch_add;
    }
    // the multiplies this algorithm performs on the matrix pipe: 36 per 4x4 output tile and (cin, cout) pair (a direct 3x3 needs 144)
    const double flops = 2.0 * n * (double)ceil_div(h, 4) * ceil_div(w, 4) * 36.0 * (double)cout * a.c;
    const double bytes = 4.0 * n * (double)h * w * (a.c + cout);
    ProfScope prof(K_CONV3_WINO4, s, flops, bytes);
    const int ntw = cout % 64 == 0 ? 4 : 2;
    const dim3 grid(xcd_grid(p.tiles_x * p.tiles_y * n, p.ntile_total / ntw));
    if (ntw == 4) {
        if (a.tab) hipLaunchKernelGGL((conv_wino4_kernel<true, 4>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_wino4_kernel<false, 4>), grid, dim3(256), 0, s, p);
    } else {
        if (a.tab) hipLaunchKernelGGL((conv_wino4_kernel<true, 2>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_wino4_kernel<false, 2>), grid, dim3(256), 0, s, p);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
