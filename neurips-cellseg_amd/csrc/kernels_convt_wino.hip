// gfx950 (MI355X): the transposed convolutions of the decoder (k3 s2 p1 op1) in Winograd form F(2,2) on the fp32 matrix pipe.
//
// Replaces the torch conv_transpose2d dispatches of MONAI's up layers (reference call sites baseline/predict.py:109 -> :45-52,
// `UNet._get_up_layer`: Convolution(..., is_transposed=True) at every decoder level but the 3-channel top one), inference only.
//
// A transposed 3x3 stride-2 convolution is four stride-1 correlations, one per output parity phase:
//     out[2i    ] = w[1] x[i]                        (one tap along that axis)
//     out[2i + 1] = w[2] x[i] + w[0] x[i + 1]        (two taps;  x beyond the edge is 0)
// along rows and columns alike (kernels_conv.hip: convt_mfma_kernel's pa / di table).  The direct form spends 1 + 2 + 2 + 4 = 9
// multiplies per input pixel and (cin, cout) pair.  The two-tap axes take the minimal-filtering form F(2,2): two outputs from three
// inputs with THREE multiplies,
//     m1 = (d0 - d1) g0,  m2 = d1 (g0 + g1),  m3 = (d2 - d1) g1;   o0 = m1 + m2,  o1 = m2 + m3          (g0 = w[2], g1 = w[0])
// so a 2 x 2 tile of input pixels (a 4 x 4 block of outputs) costs (2 + 3)^2 = 25 multiplies instead of 36: 1.44 x fewer matrix-pipe
// cycles, with transform matrices of 0 / +-1 only (the weights (g0 + g1) are summed in fp64 and rounded once at pack time: the
// arithmetic error stays at the direct kernels' level).  Per axis the five "forms" are
//     a = 0: (d0 | w[1])   a = 1: (d1 | w[1])   a = 2: (d0 - d1 | w[2])   a = 3: (d1 | w[2] + w[0])   a = 4: (d2 - d1 | w[0])
// and the 25 products are (row form a) x (column form b); a = 1 and a = 3 share their transformed input, so only 4 x 4 = 16 distinct
// transformed planes exist: V[16][k-group][tile][4] in LDS, 16 A fragments per chunk feeding 25 x 4 MFMAs.
//
// Shape (one block = 64 NT threads, 4 x 4 tiles = 8 x 8 input pixels = 16 x 16 outputs = ONE MFMA M fragment, 16 NT output channels):
//   * wave w owns channel tile w for ALL 25 products (100 accumulator registers).  The C layout of v_mfma_f32_16x16x4_f32 then puts
//     all 25 products of (tile 4 g + r, channel pl) into ONE lane: the output transform (18 additions per tile and channel), bias and
//     statistics need no exchange through LDS and no barrier - a lane stores its 4 x 16 outputs as dwords (16 lanes = one 64-byte
//     channel group of a pixel record; the four waves of the block fill the neighbouring groups of the same records);
//   * K loop over 16-channel chunks of the skip tensor, then of the tensor from below (optionally a SUM source: two tensors with their
//     lazy InstanceNorm + PReLU): 9 x 9 patch global -> registers two chunks ahead -> LDS raw (double buffered) -> thread (tile,
//     channel) forms its 16 planes -> V (double buffered): ONE barrier per chunk;
//   * B fragments: packed[chunk][25][cout / 16][64 lanes][4] as buffer loads (descriptor + offset in SGPRs) two products ahead.
#include "conv_common.h"
#include "pack_jobs.h"

namespace cs {

namespace {
SrcDev to_dev(const ConvSrc& s) { return SrcDev{s.ptr, s.tab, s.slope, s.c, s.pitch, s.tab_c ? s.tab_c : s.c, s.tab_off, s.ptr2, s.tab2, s.slope2}; }
constexpr int TW_IH = 9, TW_IW = 9;                  // input patch of a block: 8 x 8 pixels + the row / column behind them
constexpr int TW_PS = 24;                            // floats per staged raw pixel (16 + 8: two tiles of a row, 2 pixels apart, on disjoint banks)
constexpr int TW_RAW = TW_IH * TW_IW * TW_PS;        // 1944
constexpr int TW_V = 16 * 4 * 16 * 4;                // 4096: V[16 planes][4 k-groups][16 tiles][4]
// plane of an axis form: a -> {d0, d1, d0 - d1, d1, d2 - d1}
__host__ __device__ constexpr int tw_plane(int a) { return a == 0 ? 0 : a == 1 ? 1 : a == 2 ? 2 : a == 3 ? 1 : 3; }
}  // namespace

template <int NT, bool SUM2>
__global__ __launch_bounds__(64 * NT, 2) void convt_wino_kernel(ConvParams p) {
    constexpr int NTHR = 64 * NT;
    constexpr int IN_ITERS = (TW_IH * TW_IW * 4 + NTHR - 1) / NTHR;         // 2 (256 threads) / 3 (128 threads)
    constexpr int T_ITERS = 256 / NTHR;                                      // (tile, channel) items per thread
    __shared__ __attribute__((aligned(16))) float smem[2 * TW_RAW + 2 * TW_V];
    float* const raw_s = smem;
    float* const v_s = smem + 2 * TW_RAW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const WorkId wk = decode_work(p.tiles_x * p.tiles_y * p.n, p.ntile_total / NT);
    if (!wk.valid) return;
    int bid = wk.spatial;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int iy0 = ty * 8, ix0 = tx * 8;
    const int nt = wk.ct * NT + wave;                           // this wave's channel tile
    const int q = tid & 3;

    f32x4 acc[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging geometry, fixed for the K loop: element offset of the pixel inside one image plane (-1: beyond the edge = 0)
    int goff[IN_ITERS];
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (tid >> 2) + (NTHR / 4) * it;
        const int iy = pix / TW_IW, ix = pix - iy * TW_IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        goff[it] = (pix < TW_IH * TW_IW && gy < p.hin && gx < p.win) ? gy * p.win + gx : -1;
    }
    const int chunks_a = p.a.c >> 4, chunks = (p.a.c + p.b.c) >> 4;
    f32x4 rin[IN_ITERS];
    f32x4 rin2[SUM2 ? IN_ITERS : 1];
    Lazy4 lz, lz2;
    bool two = false;
    auto load_chunk = [&](int ch) {
        const bool from_a = ch < chunks_a;
        const SrcDev& src = from_a ? p.a : p.b;
        const int c0 = (from_a ? ch : ch - chunks_a) << 4;
        if constexpr (SUM2) lz.load(src, img, c0 + 4 * q, p.wpack);          // (plain launches carry no lazy table: the launcher checks)
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
    };
    auto store_raw = [&](float* raw) {
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int pix = (tid >> 2) + (NTHR / 4) * it;
            if (pix < TW_IH * TW_IW) {
                f32x4 v = rin[it];
                if constexpr (SUM2) {
                    v = lz.apply(v);
                    if (two) v += lz2.apply(rin2[it]);          // (a raw second tensor - no table - is added as it is)
                }
                if (goff[it] < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(&raw[pix * TW_PS + 4 * q]) = v;
            }
        }
    };
    // the 16 planes of one (tile, channel): rows {d0, d1, d0 - d1, d2 - d1} x columns likewise; V[plane][c >> 2][tile ^ 2 (c >> 2)][c & 3]
    // (the XOR spreads the four k-groups a 32-lane store group holds over all banks; the A-fragment read permutes its tiles the same way)
    auto transform = [&](const float* raw, float* v) {
#pragma unroll
        for (int k = 0; k < T_ITERS; ++k) {
            const int item = tid + NTHR * k;
            const int tile = item >> 4, c = item & 15;
            const float* src = raw + (((tile >> 2) * 2) * TW_IW + (tile & 3) * 2) * TW_PS + c;
            float d[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) d[i][j] = src[(i * TW_IW + j) * TW_PS];
            float r[4][3];                                      // row forms
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                r[0][j] = d[0][j];
                r[1][j] = d[1][j];
                r[2][j] = d[0][j] - d[1][j];
                r[3][j] = d[2][j] - d[1][j];
            }
            float* dst = v + (((c >> 2) * 16 + (tile ^ (2 * (c >> 2)))) * 4) + (c & 3);       // + plane * 256
#pragma unroll
            for (int ra = 0; ra < 4; ++ra) {
                dst[(ra * 4 + 0) * 256] = r[ra][0];
                dst[(ra * 4 + 1) * 256] = r[ra][1];
                dst[(ra * 4 + 2) * 256] = r[ra][0] - r[ra][1];
                dst[(ra * 4 + 3) * 256] = r[ra][2] - r[ra][1];
            }
        }
    };

    // B fragments of this wave's channel tile: packed[chunk][product a * 5 + b][ntile_total][64 lanes][4]
    const unsigned wpos = (unsigned)p.ntile_total * 64u * 16u;       // bytes between products
    const unsigned wblock = (unsigned)nt * 64u * 16u;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpack), 0, (int)((size_t)chunks * 25 * wpos), 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    auto wload = [&](int ch, int prod) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, wblock + (unsigned)(ch * 25 + prod) * wpos, 0));
    };
    // visiting order of the 25 products: plane by plane (row plane major), so that an A fragment is read once
    // plane pr in {0, 1, 2, 3} holds row forms {0}, {1, 3}, {2}, {4}
    struct Ord { int prod[25]; int plane[25]; };
    constexpr Ord ord = [] {
        Ord o{};
        int n = 0;
        for (int pr = 0; pr < 4; ++pr)
            for (int pc = 0; pc < 4; ++pc)
                for (int a = 0; a < 5; ++a)
                    for (int b = 0; b < 5; ++b)
                        if (tw_plane(a) == pr && tw_plane(b) == pc) {
                            o.prod[n] = a * 5 + b;
                            o.plane[n] = pr * 4 + pc;
                            ++n;
                        }
        return o;
    }();

    float* const raw0 = raw_s;
    float* const raw1 = raw_s + TW_RAW;
    // prologue: chunk 0 staged and transformed, chunk 1 staged, chunk 2 in flight
    load_chunk(0);
    store_raw(raw0);
    if (chunks > 1) load_chunk(1);
    __syncthreads();
    transform(raw0, v_s);
    if (chunks > 1) store_raw(raw1);
    if (chunks > 2) load_chunk(2);
    __syncthreads();

    const float* const a_base = v_s + (g * 16 + (pl ^ (2 * g))) * 4;       // + plane * 256 (+ TW_V for the odd chunks)
#pragma unroll 1
    for (int ch = 0; ch < chunks; ++ch) {
        const int par = ch & 1;
        if (ch + 1 < chunks) transform(par ? raw0 : raw1, v_s + (par ? 0 : TW_V));        // chunk ch + 1: raw[(ch + 1) & 1] -> V[(ch + 1) & 1]
        const float* a_src = a_base + par * TW_V;
        f32x4 wc[3];
        wc[0] = wload(ch, ord.prod[0]);
        wc[1] = wload(ch, ord.prod[1]);
        f32x4 av = *reinterpret_cast<const f32x4*>(a_src + ord.plane[0] * 256);
#pragma unroll
        for (int k = 0; k < 25; ++k) {
            if (k + 2 < 25) wc[(k + 2) % 3] = wload(ch, ord.prod[k + 2]);
            f32x4 an = av;
            if (k + 1 < 25 && ord.plane[k + 1] != ord.plane[k]) an = *reinterpret_cast<const f32x4*>(a_src + ord.plane[k + 1] * 256);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 b = wc[k % 3];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ord.prod[k]] = mfma4(av[j], b[j], acc[ord.prod[k]]);
            __builtin_amdgcn_sched_barrier(0);
            av = an;
        }
        if (ch + 2 < chunks) store_raw(par ? raw1 : raw0);                  // chunk ch + 2 -> raw[ch & 1] (its previous content was transformed a barrier ago)
        if (ch + 3 < chunks) load_chunk(ch + 3);
        __syncthreads();
    }

    // ---- epilogue, all in registers: lane (pl, g) holds every product of tiles 4 g ... 4 g + 3 for output channel 16 nt + pl ----
    const int co = nt * 16 + pl;
    const float bias = p.bias ? p.bias[co] : 0.f;
    float sf = 0.f, qf = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tile = 4 * g + r;
        const int iy = iy0 + 2 * (tile >> 2), ix = ix0 + 2 * (tile & 3);          // first input pixel of the tile
        float z[5][4];                                                          // column forms folded: z[a][ox]
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            z[a][0] = acc[a * 5 + 0][r];
            z[a][1] = acc[a * 5 + 2][r] + acc[a * 5 + 3][r];
            z[a][2] = acc[a * 5 + 1][r];
            z[a][3] = acc[a * 5 + 3][r] + acc[a * 5 + 4][r];
        }
#pragma unroll
        for (int oy = 0; oy < 4; ++oy) {
            const bool row_ok = iy + (oy >> 1) < p.hin;
            const size_t rowbase = ((size_t)img * p.hout + 2 * iy + oy) * p.wout + 2 * ix;
#pragma unroll
            for (int ox = 0; ox < 4; ++ox) {
                float v = oy == 0 ? z[0][ox] : oy == 1 ? z[2][ox] + z[3][ox] : oy == 2 ? z[1][ox] : z[3][ox] + z[4][ox];
                v += bias;
                if (row_ok && ix + (ox >> 1) < p.win) {
                    p.out[(rowbase + ox) * p.pitch_out + co] = v;
                    sf += v;
                    qf = __builtin_fmaf(v, v, qf);
                }
            }
        }
    }
    if (p.stats) {
        // a lane's <= 64 values of one channel are summed in fp32, the four lane groups and everything from there on in fp64
        double s = (double)sf, qq = (double)qf;
        s += __shfl_xor(s, 16);
        qq += __shfl_xor(qq, 16);
        s += __shfl_xor(s, 32);
        qq += __shfl_xor(qq, 32);
        if (lane < 16) {
            atomicAdd(p.stats + ((size_t)img * p.cout + co) * 2, s);
            atomicAdd(p.stats + ((size_t)img * p.cout + co) * 2 + 1, qq);
        }
    }
}

size_t packed_convt_wino_floats(int cout_total, int cin_total) { return (size_t)(cin_total / 16) * 25 * (cout_total / 16) * 256; }

static int g_convt_wino = [] { const char* e = getenv("CS_CONVT_WINO"); return e ? atoi(e) : 1; }();
bool convt_wino_enabled() { return g_convt_wino != 0; }
bool convt_wino_fits(int cin_a, int cin_b, int cout, int h, int w) {
    // (32-channel blocks measured no faster than the direct kernel for 64 + 64 -> 32 at 64 x 64 over 256 windows: two waves with 16-channel
    // chunks 847 vs 736 us; four waves as 2 channel tiles x 2 tile groups with 8-channel chunks - three blocks per CU, one barrier per
    // 50 MFMAs - 705 vs 727 us, and 862 vs 750 us when the tensor from below is a SUM source, as it is in the slide: the input
    // transform is shared by half as many output channels, 48 % / 40 % matrix-pipe busy against 61 % for the 64-channel blocks)
    return g_convt_wino != 0 && cin_a % 16 == 0 && cin_b % 16 == 0 && cin_a > 0 && cout % 64 == 0 && h >= 4 && w >= 4;
}

int launch_convt_wino(hipStream_t s, const ConvSrc& a, const ConvSrc& b, const float* wpack_tw, const float* bias, float* out, int pitch_out,
                      int cout, double* stats, int n, int hin, int win) {
    CS_REQUIRE(a.c % 16 == 0 && b.c % 16 == 0 && a.c > 0 && cout % 32 == 0, "convT(winograd): channels (%d + %d -> %d) unsupported", a.c, b.c, cout);
    CS_REQUIRE((a.pitch % 4) == 0 && (b.c == 0 || (b.pitch % 4) == 0), "convT(winograd): pixel pitch must be a multiple of 4 floats");
    CS_REQUIRE(!a.h16 && !a.q8 && !b.h16 && !b.q8 && !a.ptr2, "convT(winograd): plain fp32 activations only (the second operand may be a sum source)");
    CS_REQUIRE((size_t)(a.c + b.c) * 25 * cout * 4 < ((size_t)1 << 31), "convT(winograd): packed weights must stay below 2 GB");
    CS_REQUIRE(b.ptr2 || (!a.tab && !b.tab), "convT(winograd): lazy norm tables only with a sum source (the plain kernel reads raw tensors)");
    CS_REQUIRE(!a.tab, "convT(winograd): the skip tensor is read raw");
    ConvParams p{};
    p.a = to_dev(a);
    p.b = to_dev(b);
    p.wpack = wpack_tw; p.bias = bias; p.out = out; p.stats = stats;
    p.pitch_out = pitch_out; p.cout = cout; p.ntile_total = cout / 16;
    p.n = n; p.hin = hin; p.win = win; p.hout = 2 * hin; p.wout = 2 * win;
    p.tiles_x = ceil_div(win, 8);
    p.tiles_y = ceil_div(hin, 8);
    const int nt = cout % 64 == 0 ? 4 : 2;
    const dim3 grid(xcd_grid(p.tiles_x * p.tiles_y * n, p.ntile_total / nt));
    // the multiplies this algorithm performs on the matrix pipe: 25 per 2 x 2 input pixels and (cin, cout) pair (the direct form: 36)
    const double flops = 2.0 * n * (double)ceil_div(hin, 2) * ceil_div(win, 2) * 25.0 * (double)cout * (a.c + b.c);
    const double bytes = 4.0 * n * ((double)hin * win * (a.c + (b.ptr2 ? 2 : 1) * b.c) + 4.0 * hin * win * cout);
    ProfScope prof(K_CONVT, s, flops, bytes);
    if (b.ptr2) {
        if (nt == 4) hipLaunchKernelGGL((convt_wino_kernel<4, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((convt_wino_kernel<2, true>), grid, dim3(128), 0, s, p);
    } else {
        if (nt == 4) hipLaunchKernelGGL((convt_wino_kernel<4, false>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((convt_wino_kernel<2, false>), grid, dim3(128), 0, s, p);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
