// gfx950 (MI355X): Winograd F(4x4, 3x3) for the 3x3 stride-1 layers with >= 64 output channels, on the 32x32 fp32 matrix instruction.
//
// Winograd arithmetic (Y = A^T [ (G g G^T) . (B^T d B) ] A, 6 x 6 patch per 4 x 4 output tile, weights transformed
// in fp64 and rounded once, everything else fp32) and the same reference call sites (the stride-1 conv2d dispatches of
// baseline/predict.py:109 -> :45-52).  What changes against the removed round-3 kernel is the shape of the 36 point-wise GEMMs: that one contracted 16 tiles
// x 16 channels per v_mfma_f32_16x16x4_f32, so every MFMA consumes a fresh 256-byte weight fragment - a fifth of that kernel's time
// was the fragment stream through the vector L1 (DESIGN 3.1a').  Here a block owns TWO 16 x 16-pixel sub-tiles (32 Winograd tiles = the
// M of v_mfma_f32_32x32x2_f32) and 64 output channels (two N tiles of 32): a 256-byte fragment feeds twice the multiply-adds, the
// fragment stream and the A-fragment LDS reads per matrix-pipe cycle halve, and so does the MFMA instruction count.
//
// Shape (one block = 512 threads = 8 waves, two waves per SIMD):
//   * threads 0..255 stage and transform sub-tile 0, threads 256..511 sub-tile 1 - the two sub-tiles are consecutive items of the
//     16 x 16-pixel tile list, so they may lie in different images (the 16 x 16 bottom levels: one tile per window); each half has its
//     own image base, lazy-norm table and statistics slot;
//   * K loop over 16-channel chunks: halo'd 18 x 18 patch global -> registers one chunk ahead (lazy InstanceNorm + PReLU on the way)
//     -> LDS raw[sub][324 px][20]; thread (tile, channel) transforms its 6 x 6 patch into
//     V[36 positions][h][sh][32 tiles ^ swizzle][4]: channel 2 (4 sh + e) + h is element e - lane (tile, h) of the 32x32x2 A operand
//     reads its eight k-steps of a position as two ds_read_b128;
//   * wave w: positions 9 (w >> 1) ... + 8, N tile w & 1: per chunk 18 A fragment reads, 18 B fragment loads (L2 -> registers one
//     position ahead), 72 MFMAs of 64 cycles, 144 accumulator registers;
//   * epilogue: one sub-tile at a time through LDS [36][16 tiles][64] (registers 0..7 / 8..15 of every 32x32 accumulator); thread (row
//     pair, tile, channel quad) applies A^T M A, bias, fp64 statistics of the sub-tile's image, 16-byte stores.
#include <atomic>
#include <type_traits>

#include "conv_common.h"
#include "pack_jobs.h"

namespace cs {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));

constexpr int X_IH = 18, X_IW = 18;
constexpr int X_PS = 20;                               // floats per staged raw pixel (16 + 4: bank spread of the 80-byte pixel stride)
constexpr int X_RAW1 = X_IH * X_IW * X_PS;             // 6480 floats per sub-tile
constexpr int X_RAW = 2 * X_RAW1;                      // 12960
constexpr int X_V = 36 * 2 * 2 * 32 * 4;               // 18432
constexpr int X_M = 36 * 16 * 64;                      // 36864: the products of one sub-tile
constexpr int X_MAIN = X_RAW + X_V;
constexpr int X_BODY = X_MAIN > X_M ? X_MAIN : X_M;
constexpr int X_ST = 8 * 64 * 2 * 2;                   // fp64 statistics of the eight waves, 64 channels (as floats)
constexpr size_t X_LDS_BYTES = (size_t)(X_BODY + X_ST) * 4;

__device__ __forceinline__ void xbt6(const float (&x)[6], float (&y)[6]) {
    const float a = __builtin_fmaf(-4.f, x[2], x[4]), b = __builtin_fmaf(-4.f, x[1], x[3]);
    const float c = x[4] - x[2], d = 2.f * (x[3] - x[1]);
    y[0] = __builtin_fmaf(4.f, x[0], __builtin_fmaf(-5.f, x[2], x[4]));
    y[1] = a + b;
    y[2] = a - b;
    y[3] = c + d;
    y[4] = c - d;
    y[5] = __builtin_fmaf(4.f, x[1], __builtin_fmaf(-5.f, x[3], x[5]));
}
__device__ __forceinline__ void xat6(const f32x4 (&m)[6], f32x4 (&o)[4]) {
    const f32x4 s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
    o[0] = (m[0] + s1) + s2;
    o[1] = d1 + 2.f * d2;
    o[2] = s1 + 4.f * s2;
    o[3] = (d1 + 8.f * d2) + m[5];
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// rows 2 g, 2 g + 1 of A^T m for a WAVE-UNIFORM g (the output threads of a wave all own the same row pair): the branch is a scalar one,
// so a thread computes the two rows it needs instead of all four followed by eight selects
__device__ __forceinline__ void xat6_rows(const f32x4 (&m)[6], int g, f32x4& oa, f32x4& ob) {
    const f32x4 s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
    if (g == 0) {
        oa = (m[0] + s1) + s2;
        ob = d1 + 2.f * d2;
    } else {
        oa = s1 + 4.f * s2;
        ob = (d1 + 8.f * d2) + m[5];
    }
}
}  // namespace

template <bool LAZY>
__global__ __launch_bounds__(512, 2) void conv_wino4x_kernel(ConvParams p) {
    constexpr int IN_ITERS = 6;                                  // (pixel, channel quad) items per thread; the sixth only for 16 threads of a half
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const vs = smem + X_RAW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = wave >> 2;                                   // staging / transform half of this thread (wave-uniform)
    const int htid = tid & 255;
    const int n_spatial = p.tiles_x * p.tiles_y * p.n;
    const WorkId wk = decode_work((n_spatial + 1) >> 1, p.ntile_total >> 2);
    if (!wk.valid) return;
    // the two sub-tiles: items 2 s, 2 s + 1 of the tile list (the second may not exist)
    const int sp0 = 2 * wk.spatial;
    const bool has1 = sp0 + 1 < n_spatial;
    const int per_img = p.tiles_x * p.tiles_y;
    auto tile_of = [&](int sp, int& img, int& oy0, int& ox0) {
        img = sp / per_img;
        const int r = sp - img * per_img;
        const int ty = r / p.tiles_x;
        oy0 = ty * 16;
        ox0 = (r - ty * p.tiles_x) * 16;
    };
    int img_a, oy_a, ox_a, img_b, oy_b, ox_b;
    tile_of(sp0, img_a, oy_a, ox_a);
    tile_of(has1 ? sp0 + 1 : sp0, img_b, oy_b, ox_b);
    const int my_img = sub ? img_b : img_a, my_oy = sub ? oy_b : oy_a, my_ox = sub ? ox_b : ox_a;
    const bool my_valid = sub ? has1 : true;
    const int nt0 = wk.ct * 4;                                   // first 16-channel tile of the block's 64 output channels
    const int q = tid & 3;

    f32x16 acc[9];
#pragma unroll
    for (int ps = 0; ps < 9; ++ps)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ps][r] = 0.f;

    // staging geometry of this thread's half, fixed for the K loop
    unsigned boff[IN_ITERS], inb = 0;
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (htid >> 2) + 64 * it;
        const int iy = pix / X_IW, ix = pix - iy * X_IW;
        const int gy = my_oy - 1 + iy, gx = my_ox - 1 + ix;
        const bool ok = my_valid && pix < X_IH * X_IW && gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win;
        inb |= ok ? (1u << it) : 0u;
        boff[it] = ok ? ((unsigned)(gy * p.win + gx) * (unsigned)p.a.pitch + 4u * q) * 4u : 16u * q;
    }
    const char* const img_base = reinterpret_cast<const char*>(p.a.ptr + (size_t)my_img * p.hin * p.win * p.a.pitch);
    const int chunks = p.a.c >> 4;
    const bool tail_wave = (wave & 3) == 0;                      // the waves whose first 16 lanes own pixels 320..323 of their half
    // the staged patch travels in two halves of three (pixel, quad) items: half 0 = items 0..2, half 1 = items 3..5 (the last one only
    // for 16 threads of each staging half); inside the K loop both halves pass through the same three registers
    f32x4 rin[3];
    Lazy4 lz;
    float* const my_raw = smem + sub * X_RAW1;
    auto load_half = [&](auto hc, int ch) {
        constexpr int H = decltype(hc)::value;
        const char* base = img_base + (size_t)(ch << 4) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) rin[k] = *reinterpret_cast<const f32x4*>(base + boff[3 * H + k]);   // unconditional (clamped)
    };
    auto store_one = [&](int it, const f32x4& r) {
        const int pix = (htid >> 2) + 64 * it;
        const f32x4 a = LAZY ? lz.apply_on(r) : r;
        const f32x4 v = (inb >> it) & 1u ? a : f32x4{0.f, 0.f, 0.f, 0.f};           // padding stays 0 after the activation
        *reinterpret_cast<f32x4*>(&my_raw[pix * X_PS + 4 * q]) = v;
    };
    auto store_half = [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        if (H == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) store_one(k, rin[k]);
        } else {
            store_one(3, rin[0]);
            store_one(4, rin[1]);
            if (tail_wave && (htid >> 2) + 64 * 5 < X_IH * X_IW) store_one(5, rin[2]);
        }
    };
    const std::integral_constant<int, 0> H0;
    const std::integral_constant<int, 1> H1;
    // input transform of one (tile, channel): tile = tid >> 4 (0..31: sub-tile = tile >> 4), channel = tid & 15
    const int t_tile = tid >> 4, t_lt = t_tile & 15, t_c = tid & 15;
    const float* const t_src = my_raw + (((t_lt >> 2) * 4) * X_IW + (t_lt & 3) * 4) * X_PS + t_c;
    // V[position][h = c & 1][sh = c >> 3][tile ^ 2 (2 h + sh)][e = (c >> 1) & 3]: the XOR spreads the four (h, sh) planes a 32-lane
    // store group holds over all banks; the A-fragment read applies the same permutation
    const int t_h = t_c & 1, t_sh = t_c >> 3, t_e = (t_c >> 1) & 3;
    float* const t_dst = vs + ((t_h * 2 + t_sh) * 32 + (t_tile ^ (2 * (2 * t_h + t_sh)))) * 4 + t_e;        // + position * 512
    auto transform = [&]() {
        float t[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float x[6], y[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] = t_src[(i * X_IW + j) * X_PS];
            xbt6(x, y);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][j] = y[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float y[6];
            xbt6(t[i], y);
#pragma unroll
            for (int j = 0; j < 6; ++j) t_dst[(i * 6 + j) * 512] = y[j];
        }
    };

    // B fragments: packed[chunk][position 36][cout / 32][u 2][64 lanes][4]: cout = 32 nt + (lane & 31), cin = 16 chunk + 2 (4 u + e) + (lane >> 5)
    const int pg = wave >> 1, ntw = wave & 1;
    const int nt32_total = p.ntile_total >> 1;
    // every fragment load is a buffer load: descriptor + the step's byte offset in SGPRs, the lane's 16 bytes in ONE loop-invariant VGPR -
    // no vector address arithmetic in the MFMA stream (a 64-bit per-lane address cost three VALU instructions per load, and on the
    // fp32 path every VALU instruction is taken from the matrix pipe)
    const unsigned wpos = (unsigned)nt32_total * 2u * 64u * 16u;      // bytes between positions
    const unsigned wblock = (unsigned)(wk.ct * 2 + ntw) * 2u * 64u * 16u + (unsigned)pg * 9u * wpos;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpack), 0, (int)((size_t)chunks * 36 * wpos), 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    // three-slot ring: the fragments of a position are requested TWO positions ahead (9 steps per chunk: slot = position % 3)
    f32x4 wc[3][2];
    const int last_step = chunks * 9 - 1;
    auto wload = [&](int step, f32x4 (&dst)[2]) {
        const int ch = step / 9, ps = step - ch * 9;
        const unsigned so = wblock + (unsigned)(ch * 36 + ps) * wpos;
        dst[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so, 0));
        dst[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + 1024u, so, 0));
    };

    if (LAZY) lz.load(p.a, my_img, 4 * q, p.wpack);
    load_half(H0, 0);
    wload(0, wc[0]);
    wload(last_step < 1 ? last_step : 1, wc[1]);
    store_half(H0);
    load_half(H1, 0);
    store_half(H1);
    __syncthreads();
    transform();
    __syncthreads();
    const int a_i = lane & 31, a_h = lane >> 5;
    // + position * 512; the second read (sh = 1) sits 128 floats further with its own tile permutation
    const float* const a_src0 = vs + (pg * 9) * 512 + ((a_h * 2 + 0) * 32 + (a_i ^ (2 * (2 * a_h + 0)))) * 4;
    const float* const a_src1 = vs + (pg * 9) * 512 + ((a_h * 2 + 1) * 32 + (a_i ^ (2 * (2 * a_h + 1)))) * 4;
#pragma unroll 1
    for (int ch = 0; ch < chunks; ++ch) {
        const bool more = ch + 1 < chunks;
        const int nch = more ? ch + 1 : ch;        // the loads of the last trip re-read its own chunk: a skipped load group would mis-size the ring's waits
        f32x4 av[2][2];                            // A fragments of the current and the next position (slot = position & 1)
        av[0][0] = *reinterpret_cast<const f32x4*>(a_src0);
        av[0][1] = *reinterpret_cast<const f32x4*>(a_src1);
        // the next chunk's patch is staged INSIDE the MFMA phase (raw is free while V is being contracted): half 0 requested at position
        // 0 and stored at position 3, half 1 requested at 3 and stored at 7 - three registers in flight instead of six, and no separate
        // norm + store phase in front of the barrier
#pragma unroll
        for (int ps = 0; ps < 9; ++ps) {
            {
                const int step = ch * 9 + ps + 2;
                wload(step < last_step ? step : last_step, wc[(ps + 2) % 3]);
            }
            if (ps == 0) {
                if (LAZY) lz.load(p.a, my_img, (nch << 4) + 4 * q, p.wpack);
                load_half(H0, nch);
            }
            if (ps == 3) {
                if (more) store_half(H0);
                load_half(H1, nch);
            }
            if (ps == 7 && more) store_half(H1);
            if (ps < 8) {
                av[(ps + 1) & 1][0] = *reinterpret_cast<const f32x4*>(a_src0 + (ps + 1) * 512);
                av[(ps + 1) & 1][1] = *reinterpret_cast<const f32x4*>(a_src1 + (ps + 1) * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[ps] = mfma32(av[ps & 1][0][e], wc[ps % 3][0][e], acc[ps]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[ps] = mfma32(av[ps & 1][1][e], wc[ps % 3][1][e], acc[ps]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (more) {
            transform();
            __syncthreads();
        }
    }

    // ---- epilogue: one SUB-TILE (16 tiles x 64 channels) at a time through LDS ----
    // C layout of a 32x32 accumulator: row (tile) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column (channel) = lane & 31: registers 0..7 hold
    // sub-tile 0, registers 8..15 sub-tile 1 - every wave hands over half of each accumulator per pass, so only 72 accumulator registers
    // are live beside the output transform of the first pass
    float* const ms = smem;                                     // [36 positions][16 tiles][64 channels]
    double* const st_s = reinterpret_cast<double*>(smem + X_BODY);           // [8 waves][64 channels][2]
    const int e_grp = tid >> 8;                                 // output rows 2 e_grp, 2 e_grp + 1 of the tile
    const int e_lt = (tid >> 4) & 15, e_q = tid & 15;           // output thread: tile of the sub-tile, channel quad of the 64
    const int e_ty = e_lt >> 2, e_tx = e_lt & 3;
    const int co = nt0 * 16 + 4 * e_q;
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + co);
    auto pass = [&](auto sbc) {
        constexpr int SB = decltype(sbc)::value;
        const int e_img = SB ? img_b : img_a, e_oy0 = SB ? oy_b : oy_a, e_ox0 = SB ? ox_b : ox_a;
        const bool e_valid = SB ? has1 : true;
#pragma unroll
        for (int ps = 0; ps < 9; ++ps)
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int tile = (r8 & 3) + 8 * (r8 >> 2) + 4 * a_h;
                ms[((pg * 9 + ps) * 16 + tile) * 64 + ntw * 32 + a_i] = acc[ps][8 * SB + r8];
            }
        __syncthreads();
        {
            const float* src = ms + e_lt * 64 + 4 * e_q;
            float sf[4] = {0.f, 0.f, 0.f, 0.f}, qf[4] = {0.f, 0.f, 0.f, 0.f};
            const int ox = e_ox0 + 4 * e_tx;
            f32x4 tm[2][6];                                     // rows 2 e_grp, 2 e_grp + 1 of A^T M: 6 position columns each
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                f32x4 m[6];
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) m[xi] = *reinterpret_cast<const f32x4*>(src + (xi * 6 + nu) * 1024);
                f32x4 o[4];
                xat6(m, o);
                tm[0][nu] = e_grp ? o[2] : o[0];
                tm[1][nu] = e_grp ? o[3] : o[1];
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                f32x4 y[4];
                xat6(tm[ii], y);
                const int oy = e_oy0 + 4 * e_ty + 2 * e_grp + ii;
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    f32x4 v = y[jx] + bias4;
                    if (e_valid && oy < p.hout && ox + jx < p.wout) {
                        const size_t gp = ((size_t)e_img * p.hout + oy) * p.wout + ox + jx;
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
                // a thread's 8 values per channel and the 4 lanes (tiles) of the wave that hold the same channel quad are summed in fp32
                // (<= 32 values per partial sum), everything from there on is fp64
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    sf[k] += __shfl_xor(sf[k], 16);
                    qf[k] += __shfl_xor(qf[k], 16);
                    sf[k] += __shfl_xor(sf[k], 32);
                    qf[k] += __shfl_xor(qf[k], 32);
                }
                if (lane < 16) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        st_s[(wave * 64 + 4 * e_q + k) * 2] = (double)sf[k];
                        st_s[(wave * 64 + 4 * e_q + k) * 2 + 1] = (double)qf[k];
                    }
                }
            }
        }
        __syncthreads();
        if (p.stats && tid < 128 && e_valid) {
            const int c = tid >> 1, k = tid & 1;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += st_s[(w * 64 + c) * 2 + k];
            atomicAdd(p.stats + ((size_t)e_img * p.cout + nt0 * 16 + c) * 2 + k, v);
        }
    };
    pass(std::integral_constant<int, 0>());
    pass(std::integral_constant<int, 1>());
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same block (512 threads, two sub-tiles, 64 output channels, the staging and transform above) with the point-wise GEMMs on
// v_mfma_f32_16x16x4_f32 and the ownership turned round: wave w = (channel tile w & 3) x (sub-tile w >> 2) x ALL 36 positions.  The C
// layout then keeps every product of an output pixel block in one lane, so the output transform, bias and statistics run in registers -
// no exchange through LDS, no epilogue barriers (the two-pass exchange of the kernel above is 20 k cycles of a block that is alone on its
// CU: 40 % of a 64 -> 64 block's life).  A weight fragment feeds 4 instead of 8 MFMAs' worth of work; the two waves of a channel tile
// read the same fragments (one fetch from L2, two from the vector L1).  Used where the K loop is short (cin <= 128); the deep layers keep
// the 32x32x2 form, whose halved fragment stream weighs more there.
// ---------------------------------------------------------------------------------------------------------------------------------
// NSUB = 2: the 512-thread block of the kernel above (two sub-tiles share the staging phases and barriers, ONE block per CU).
// NSUB = 1: a 256-thread block of ONE sub-tile with the same per-wave work (raw 26 KB + V 37 KB: TWO independent blocks per CU, whose
// prologues, barriers and epilogues fall into each other's matrix phases instead of onto the same cycles).
template <bool LAZY, int NSUB>
__global__ __launch_bounds__(256 * NSUB, 2) void conv_wino4y_kernel(ConvParams p) {
    constexpr int TPP = 16 * NSUB;                               // tiles per position plane of V
    constexpr int VPOS = TPP * 16;                               // floats per position
    constexpr int IN_ITERS = 6;                                  // (pixel, channel quad) items per thread; the sixth only for 16 threads of a half
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const vs = smem + NSUB * X_RAW1;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = wave >> 2;                                   // staging / transform half of this thread (wave-uniform)
    const int htid = tid & 255;
    const int n_spatial = p.tiles_x * p.tiles_y * p.n;
    const WorkId wk = decode_work(NSUB == 2 ? (n_spatial + 1) >> 1 : n_spatial, p.ntile_total >> 2);
    if (!wk.valid) return;
    // the two sub-tiles: items 2 s, 2 s + 1 of the tile list (the second may not exist)
    const int sp0 = NSUB * wk.spatial;
    const bool has1 = NSUB == 2 && sp0 + 1 < n_spatial;
    const int per_img = p.tiles_x * p.tiles_y;
    auto tile_of = [&](int sp, int& img, int& oy0, int& ox0) {
        img = sp / per_img;
        const int r = sp - img * per_img;
        const int ty = r / p.tiles_x;
        oy0 = ty * 16;
        ox0 = (r - ty * p.tiles_x) * 16;
    };
    int img_a, oy_a, ox_a, img_b, oy_b, ox_b;
    tile_of(sp0, img_a, oy_a, ox_a);
    tile_of(has1 ? sp0 + 1 : sp0, img_b, oy_b, ox_b);
    const int my_img = sub ? img_b : img_a, my_oy = sub ? oy_b : oy_a, my_ox = sub ? ox_b : ox_a;
    const bool my_valid = sub ? has1 : true;
    const int nt0 = wk.ct * 4;                                   // first 16-channel tile of the block's 64 output channels
    const int q = tid & 3;

    f32x4 acc[36];                                               // every position of (16 tiles, 16 channels): the wave's own output tile
#pragma unroll
    for (int ps = 0; ps < 36; ++ps) acc[ps] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging geometry of this thread's half, fixed for the K loop
    unsigned boff[IN_ITERS], inb = 0;
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (htid >> 2) + 64 * it;
        const int iy = pix / X_IW, ix = pix - iy * X_IW;
        const int gy = my_oy - 1 + iy, gx = my_ox - 1 + ix;
        const bool ok = my_valid && pix < X_IH * X_IW && gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win;
        inb |= ok ? (1u << it) : 0u;
        boff[it] = ok ? ((unsigned)(gy * p.win + gx) * (unsigned)p.a.pitch + 4u * q) * 4u : 16u * q;
    }
    const char* const img_base = reinterpret_cast<const char*>(p.a.ptr + (size_t)my_img * p.hin * p.win * p.a.pitch);
    const int chunks = p.a.c >> 4;
    const bool tail_wave = (wave & 3) == 0;                      // the waves whose first 16 lanes own pixels 320..323 of their half
    // the staged patch travels in two halves of three (pixel, quad) items: half 0 = items 0..2, half 1 = items 3..5 (the last one only
    // for 16 threads of each staging half); inside the K loop both halves pass through the same three registers
    f32x4 rin[3];
    Lazy4 lz;
    float* const my_raw = smem + sub * X_RAW1;
    auto load_half = [&](auto hc, int ch) {
        constexpr int H = decltype(hc)::value;
        const char* base = img_base + (size_t)(ch << 4) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) rin[k] = *reinterpret_cast<const f32x4*>(base + boff[3 * H + k]);   // unconditional (clamped)
    };
    auto store_one = [&](int it, const f32x4& r) {
        const int pix = (htid >> 2) + 64 * it;
        const f32x4 a = LAZY ? lz.apply_on(r) : r;
        const f32x4 v = (inb >> it) & 1u ? a : f32x4{0.f, 0.f, 0.f, 0.f};           // padding stays 0 after the activation
        *reinterpret_cast<f32x4*>(&my_raw[pix * X_PS + 4 * q]) = v;
    };
    auto store_half = [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        if (H == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) store_one(k, rin[k]);
        } else {
            store_one(3, rin[0]);
            store_one(4, rin[1]);
            if (tail_wave && (htid >> 2) + 64 * 5 < X_IH * X_IW) store_one(5, rin[2]);
        }
    };
    const std::integral_constant<int, 0> H0;
    const std::integral_constant<int, 1> H1;
    // input transform of one (tile, channel): tile = tid >> 4 (0..31: sub-tile = tile >> 4), channel = tid & 15
    const int t_tile = tid >> 4, t_lt = t_tile & 15, t_c = tid & 15;
    const float* const t_src = my_raw + (((t_lt >> 2) * 4) * X_IW + (t_lt & 3) * 4) * X_PS + t_c;
    // V[position][k-group c >> 2][tile ^ 2 (c >> 2)][c & 3] - the A-fragment order of v_mfma_f32_16x16x4_f32 for 32 tiles; the XOR spreads the
    // four k-groups a 32-lane store group holds over all banks, the A-fragment read applies the same permutation
    float* const t_dst = vs + ((t_c >> 2) * TPP + (t_tile ^ (2 * (t_c >> 2)))) * 4 + (t_c & 3);               // + position * VPOS
    auto transform = [&]() {
        float t[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float x[6], y[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] = t_src[(i * X_IW + j) * X_PS];
            xbt6(x, y);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][j] = y[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float y[6];
            xbt6(t[i], y);
#pragma unroll
            for (int j = 0; j < 6; ++j) t_dst[(i * 6 + j) * VPOS] = y[j];
        }
    };

    // wave w: channel tile nt0 + (w & 3) (16 channels) x tile half w >> 2 (= sub-tile) x ALL 36 positions: 144 accumulator registers whose C
    // layout puts every position of (tile 4 g + r, channel pl) into one lane - the output transform needs no exchange.
    // B fragments: packed[chunk][position 36][cout / 16][64 lanes][4] (the order of packed_wino4_floats), buffer loads as in the kernel above
    const int th = wave >> 2, ntw = wave & 3;
    const int pl = lane & 15, g = lane >> 4;
    const unsigned wpos = (unsigned)p.ntile_total * 64u * 16u;        // bytes between positions
    const unsigned wblock = (unsigned)(nt0 + ntw) * 64u * 16u;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpack), 0, (int)((size_t)chunks * 36 * wpos), 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    // positions are contracted in PAIRS (two independent accumulator chains: the 16x16x4 instruction has a 40-cycle dependent latency on a
    // 32-cycle issue); six-slot ring: the fragments of a pair are requested two pairs ahead (18 pairs per chunk: slot = pair % 3)
    f32x4 wc[3][2];
    const int last_pair = chunks * 18 - 1;
    auto wload = [&](int pair, f32x4 (&dst)[2]) {
        const unsigned so = wblock + (unsigned)(2 * pair) * wpos;        // (pair index runs over chunks: 36 positions per chunk, consecutive)
        dst[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so, 0));
        dst[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so + wpos, 0));
    };

    if (LAZY) lz.load(p.a, my_img, 4 * q, p.wpack);
    load_half(H0, 0);
    wload(0, wc[0]);
    wload(last_pair < 1 ? last_pair : 1, wc[1]);
    store_half(H0);
    load_half(H1, 0);
    store_half(H1);
    __syncthreads();
    transform();
    __syncthreads();
    const float* const a_src = vs + (g * TPP + th * 16 + (pl ^ (2 * g))) * 4;              // + position * VPOS
#pragma unroll 1
    for (int ch = 0; ch < chunks; ++ch) {
        const bool more = ch + 1 < chunks;
        const int nch = more ? ch + 1 : ch;        // the loads of the last trip re-read its own chunk: a skipped load group would mis-size the ring's waits
        f32x4 av[2][2];                            // A fragments of the current and the next pair of positions
        av[0][0] = *reinterpret_cast<const f32x4*>(a_src);
        av[0][1] = *reinterpret_cast<const f32x4*>(a_src + VPOS);
        // the next chunk's patch is staged INSIDE the MFMA phase (raw is free while V is being contracted): half 0 requested at pair 0 and
        // stored at pair 6, half 1 requested at 6 and stored at 14
#pragma unroll
        for (int pr = 0; pr < 18; ++pr) {
            {
                const int pair = ch * 18 + pr + 2;
                wload(pair < last_pair ? pair : last_pair, wc[(pr + 2) % 3]);
            }
            if (pr == 0) {
                if (LAZY) lz.load(p.a, my_img, (nch << 4) + 4 * q, p.wpack);
                load_half(H0, nch);
            }
            if (pr == 6) {
                if (more) store_half(H0);
                load_half(H1, nch);
            }
            if (pr == 14 && more) store_half(H1);
            if (pr < 17) {
                av[(pr + 1) & 1][0] = *reinterpret_cast<const f32x4*>(a_src + (2 * pr + 2) * VPOS);
                av[(pr + 1) & 1][1] = *reinterpret_cast<const f32x4*>(a_src + (2 * pr + 3) * VPOS);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * pr] = mfma4(av[pr & 1][0][j], wc[pr % 3][0][j], acc[2 * pr]);
                acc[2 * pr + 1] = mfma4(av[pr & 1][1][j], wc[pr % 3][1][j], acc[2 * pr + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (more) {
            transform();
            __syncthreads();
        }
    }

    // ---- epilogue, all in registers: lane (pl, g) holds the 36 products of tiles 4 g ... 4 g + 3 of sub-tile `th` for channel 16 (nt0 + ntw) + pl ----
    {
        const int e_img = th ? img_b : img_a, e_oy0 = th ? oy_b : oy_a, e_ox0 = th ? ox_b : ox_a;
        const bool e_valid = th ? has1 : true;
        const int co = (nt0 + ntw) * 16 + pl;
        const float bias = p.bias ? p.bias[co] : 0.f;
        // outputs leave as buffer stores: descriptor of the image's output plane + the pixel's byte offset (i, jx, r: wave-uniform) in SGPRs,
        // ONE per-lane offset (tile row g, channel) - no 64-bit vector address arithmetic per store
        const size_t img_bytes = (size_t)p.hout * p.wout * p.pitch_out * 4;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)e_img * p.hout * p.wout * p.pitch_out, 0, (int)img_bytes, 0x00020000);
        const unsigned ovoff = (((unsigned)(e_oy0 + 4 * g) * (unsigned)p.wout + (unsigned)e_ox0) * (unsigned)p.pitch_out + (unsigned)co) * 4u;
        const unsigned prow = (unsigned)p.wout * (unsigned)p.pitch_out * 4u, ppix = (unsigned)p.pitch_out * 4u;
        const bool inside = e_oy0 + 16 <= p.hout && e_ox0 + 16 <= p.wout;          // wave-uniform: the whole sub-tile lies inside the image
        float sf = 0.f, qf = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // tile 4 g + r of the sub-tile: tile row g, tile column r
            const int oyt = e_oy0 + 4 * g, oxt = e_ox0 + 4 * r;
            float t[4][6];                                       // A^T M: rows of the output tile x position columns
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                const float m0 = acc[0 * 6 + nu][r], m1 = acc[1 * 6 + nu][r], m2 = acc[2 * 6 + nu][r], m3 = acc[3 * 6 + nu][r],
                            m4 = acc[4 * 6 + nu][r], m5 = acc[5 * 6 + nu][r];
                const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                t[0][nu] = (m0 + s1) + s2;
                t[1][nu] = d1 + 2.f * d2;
                t[2][nu] = s1 + 4.f * s2;
                t[3][nu] = (d1 + 8.f * d2) + m5;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s1 = t[i][1] + t[i][2], d1 = t[i][1] - t[i][2], s2 = t[i][3] + t[i][4], d2 = t[i][3] - t[i][4];
                float y[4];
                y[0] = ((t[i][0] + s1) + s2) + bias;
                y[1] = (d1 + 2.f * d2) + bias;
                y[2] = (s1 + 4.f * s2) + bias;
                y[3] = ((d1 + 8.f * d2) + t[i][5]) + bias;
                if (p.addend) {                                  // (data-gradient use: the residual path's gradient joins here)
                    const size_t rowbase = ((size_t)e_img * p.hout + oyt + i) * p.wout + oxt;
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx)
                        if (e_valid && oyt + i < p.hout && oxt + jx < p.wout) y[jx] += p.addend[(rowbase + jx) * p.pitch_add + co];
                }
                if (e_valid && inside) {
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[jx]), orsrc, ovoff, (unsigned)i * prow + (unsigned)(4 * r + jx) * ppix, 0);
                        sf += y[jx];
                        qf = __builtin_fmaf(y[jx], y[jx], qf);
                    }
                } else if (e_valid && oyt + i < p.hout) {
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx)
                        if (oxt + jx < p.wout) {
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[jx]), orsrc, ovoff, (unsigned)i * prow + (unsigned)(4 * r + jx) * ppix, 0);
                            sf += y[jx];
                            qf = __builtin_fmaf(y[jx], y[jx], qf);
                        }
                }
            }
        }
        if (p.stats && e_valid) {
            // a lane's <= 64 values of one channel in fp32, the four lane groups and everything from there on in fp64; the two waves that hold the
            // same channel tile (the two sub-tiles) add to their own images' slots
            double s = (double)sf, qq = (double)qf;
            s += __shfl_xor(s, 16);
            qq += __shfl_xor(qq, 16);
            s += __shfl_xor(s, 32);
            qq += __shfl_xor(qq, 32);
            if (lane < 16) {
                atomicAdd(p.stats + ((size_t)e_img * p.cout + co) * 2, s);
                atomicAdd(p.stats + ((size_t)e_img * p.cout + co) * 2 + 1, qq);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The 32-output-channel form (the 32 -> 32 layers at 128 x 128: K = 2 sixteen-channel chunks).  The round-3 kernel's block for
// them (16 tiles x 32 channels) lives 11 us for 2 us of matrix work: first patch's round trip, two stagings, exchange, output
// transform, store tail.  Here a block is 256 threads = 4 waves, TWO sub-tiles (32 tiles = the M of the 32x32x2 instruction) x 32 output
// channels (ONE N tile): wave w owns positions 9 w ... 9 w + 8, 144 accumulator registers, half the MFMA instructions and weight
// fragments per tile, and a prologue / epilogue per 32 tiles instead of per 16.  The K loop walks EIGHT-channel chunks (four k-steps
// of the instruction: one ds_read_b128 of A and one buffer load of B per position), which keeps raw (31 KB) + V (37 KB) and the
// per-sub-tile exchange (74 KB) small enough for two blocks per CU - the second block's matrix phase covers this one's staging,
// barriers and epilogue.  Same arithmetic, weights in the 32x32x2 fragment order of the kernel above.
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int Z_PS = 12;                               // floats per staged raw pixel (8 + 4: keeps the 16-byte stores aligned)
constexpr int Z_RAW1 = X_IH * X_IW * Z_PS;             // 3888 floats per sub-tile
constexpr int Z_RAW = 2 * Z_RAW1;
constexpr int Z_V = 36 * 2 * 32 * 4;                   // 9216: V[36 positions][h][32 tiles][4 k-steps]
constexpr int Z_M = 36 * 16 * 32;                      // 18432: the products of one sub-tile
constexpr int Z_MAIN = Z_RAW + Z_V;
constexpr int Z_BODY = Z_MAIN > Z_M ? Z_MAIN : Z_M;
constexpr int Z_ST = 4 * 32 * 2 * 2;                   // fp64 statistics of the four waves (as floats)
constexpr size_t Z_LDS_BYTES = (size_t)(Z_BODY + Z_ST) * 4;
}  // namespace

template <bool LAZY>
__global__ __launch_bounds__(256, 2) void conv_wino4z_kernel(ConvParams p) {
    constexpr int IN_ITERS = 6;                                  // (pixel, channel quad) items per thread; the sixth only for 8 threads of a half
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const vs = smem + Z_RAW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = wave >> 1;                                   // staging half of this thread (wave-uniform): threads 0..127 / 128..255
    const int htid = tid & 127;
    const int n_spatial = p.tiles_x * p.tiles_y * p.n;
    const WorkId wk = decode_work((n_spatial + 1) >> 1, p.ntile_total >> 1);
    if (!wk.valid) return;
    const int sp0 = 2 * wk.spatial;
    const bool has1 = sp0 + 1 < n_spatial;
    const int per_img = p.tiles_x * p.tiles_y;
    auto tile_of = [&](int sp, int& img, int& oy0, int& ox0) {
        img = sp / per_img;
        const int r = sp - img * per_img;
        const int ty = r / p.tiles_x;
        oy0 = ty * 16;
        ox0 = (r - ty * p.tiles_x) * 16;
    };
    int img_a, oy_a, ox_a, img_b, oy_b, ox_b;
    tile_of(sp0, img_a, oy_a, ox_a);
    tile_of(has1 ? sp0 + 1 : sp0, img_b, oy_b, ox_b);
    const int my_img = sub ? img_b : img_a, my_oy = sub ? oy_b : oy_a, my_ox = sub ? ox_b : ox_a;
    const bool my_valid = sub ? has1 : true;
    const int nt0 = wk.ct * 2;                                   // first 16-channel tile of the block's 32 output channels
    const int q = tid & 1;

    f32x16 acc[9];
#pragma unroll
    for (int ps = 0; ps < 9; ++ps)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ps][r] = 0.f;

    unsigned boff[IN_ITERS], inb = 0;
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (htid >> 1) + 64 * it;
        const int iy = pix / X_IW, ix = pix - iy * X_IW;
        const int gy = my_oy - 1 + iy, gx = my_ox - 1 + ix;
        const bool ok = my_valid && pix < X_IH * X_IW && gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win;
        inb |= ok ? (1u << it) : 0u;
        boff[it] = ok ? ((unsigned)(gy * p.win + gx) * (unsigned)p.a.pitch + 4u * q) * 4u : 16u * q;
    }
    const char* const img_base = reinterpret_cast<const char*>(p.a.ptr + (size_t)my_img * p.hin * p.win * p.a.pitch);
    const int chunks = p.a.c >> 3;                               // 8-channel chunks
    const bool tail_wave = (wave & 1) == 0;                      // the waves whose first 8 lanes own pixels 320..323 of their half
    f32x4 rin[3];
    Lazy4 lz;
    float* const my_raw = smem + sub * Z_RAW1;
    auto load_half = [&](auto hc, int ch) {
        constexpr int H = decltype(hc)::value;
        const char* base = img_base + (size_t)(ch << 3) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) rin[k] = *reinterpret_cast<const f32x4*>(base + boff[3 * H + k]);   // unconditional (clamped)
    };
    auto store_one = [&](int it, const f32x4& r) {
        const int pix = (htid >> 1) + 64 * it;
        const f32x4 a = LAZY ? lz.apply_on(r) : r;
        const f32x4 v = (inb >> it) & 1u ? a : f32x4{0.f, 0.f, 0.f, 0.f};           // padding stays 0 after the activation
        *reinterpret_cast<f32x4*>(&my_raw[pix * Z_PS + 4 * q]) = v;
    };
    auto store_half = [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        if (H == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) store_one(k, rin[k]);
        } else {
            store_one(3, rin[0]);
            store_one(4, rin[1]);
            if (tail_wave && (htid >> 1) + 64 * 5 < X_IH * X_IW) store_one(5, rin[2]);
        }
    };
    const std::integral_constant<int, 0> H0;
    const std::integral_constant<int, 1> H1;
    // input transform of one (tile, channel): tile = tid >> 3 (0..31: sub-tile = tile >> 4), channel = tid & 7
    const int t_tile = tid >> 3, t_lt = t_tile & 15, t_c = tid & 7;
    const float* const t_src = smem + (t_tile >> 4) * Z_RAW1 + (((t_lt >> 2) * 4) * X_IW + (t_lt & 3) * 4) * Z_PS + t_c;
    // V[position][h = c & 1][tile][e = c >> 1]: channel 2 e + h of the chunk is k-step e of lane half h
    float* const t_dst = vs + ((t_c & 1) * 32 + t_tile) * 4 + (t_c >> 1);            // + position * 256
    auto transform = [&]() {
        float t[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float x[6], y[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] = t_src[(i * X_IW + j) * Z_PS];
            xbt6(x, y);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][j] = y[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float y[6];
            xbt6(t[i], y);
#pragma unroll
            for (int j = 0; j < 6; ++j) t_dst[(i * 6 + j) * 256] = y[j];
        }
    };

    // B fragments: packed[chunk16][position 36][cout / 32][u 2][64 lanes][4]; the 8-channel chunk c8 is (chunk16, u) = (c8 >> 1, c8 & 1)
    const int nt32_total = p.ntile_total >> 1;
    const unsigned wpos = (unsigned)nt32_total * 2u * 64u * 16u;      // bytes between positions
    const unsigned wblock = (unsigned)wk.ct * 2u * 64u * 16u + (unsigned)wave * 9u * wpos;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpack), 0, (int)((size_t)(p.a.c >> 4) * 36 * wpos), 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    f32x4 wc[3];                                                 // three-slot ring, fragments requested two positions ahead
    const int last_step = chunks * 9 - 1;
    auto wload = [&](int step, f32x4& dst) {
        const int ch = step / 9, ps = step - ch * 9;
        const unsigned so = wblock + (unsigned)((ch >> 1) * 36 + ps) * wpos + (unsigned)(ch & 1) * 1024u;
        dst = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so, 0));
    };

    if (LAZY) lz.load(p.a, my_img, 4 * q, p.wpack);
    load_half(H0, 0);
    wload(0, wc[0]);
    wload(last_step < 1 ? last_step : 1, wc[1]);
    store_half(H0);
    load_half(H1, 0);
    store_half(H1);
    __syncthreads();
    transform();
    __syncthreads();
    const int a_i = lane & 31, a_h = lane >> 5;
    const float* const a_src = vs + (wave * 9) * 256 + (a_h * 32 + a_i) * 4;           // + position * 256
#pragma unroll 1
    for (int ch = 0; ch < chunks; ++ch) {
        const bool more = ch + 1 < chunks;
        const int nch = more ? ch + 1 : ch;        // (the last trip re-reads its own chunk: a skipped load group would mis-size the ring's waits)
        f32x4 av[2];
        av[0] = *reinterpret_cast<const f32x4*>(a_src);
#pragma unroll
        for (int ps = 0; ps < 9; ++ps) {
            {
                const int step = ch * 9 + ps + 2;
                wload(step < last_step ? step : last_step, wc[(ps + 2) % 3]);
            }
            if (ps == 0) {
                if (LAZY) lz.load(p.a, my_img, (nch << 3) + 4 * q, p.wpack);
                load_half(H0, nch);
            }
            if (ps == 3) {
                if (more) store_half(H0);
                load_half(H1, nch);
            }
            if (ps == 7 && more) store_half(H1);
            if (ps < 8) av[(ps + 1) & 1] = *reinterpret_cast<const f32x4*>(a_src + (ps + 1) * 256);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[ps] = mfma32(av[ps & 1][e], wc[ps % 3][e], acc[ps]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (more) {
            transform();
            __syncthreads();
        }
    }

    // ---- epilogue: one sub-tile (16 tiles x 32 channels) at a time through LDS; registers 0..7 / 8..15 of every accumulator ----
    float* const ms = smem;                                     // [36 positions][16 tiles][32 channels]
    double* const st_s = reinterpret_cast<double*>(smem + Z_BODY);           // [4 waves][32 channels][2]
    const int e_grp = __builtin_amdgcn_readfirstlane(tid >> 7);  // output rows 2 e_grp, 2 e_grp + 1 of the tile (wave-uniform)
    const int e_lt = (tid >> 3) & 15, e_q = tid & 7;
    const int e_ty = e_lt >> 2, e_tx = e_lt & 3;
    const int co = nt0 * 16 + 4 * e_q;
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + co);
    auto pass = [&](auto sbc) {
        constexpr int SB = decltype(sbc)::value;
        const int e_img = SB ? img_b : img_a, e_oy0 = SB ? oy_b : oy_a, e_ox0 = SB ? ox_b : ox_a;
        const bool e_valid = SB ? has1 : true;
#pragma unroll
        for (int ps = 0; ps < 9; ++ps)
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int tile = (r8 & 3) + 8 * (r8 >> 2) + 4 * a_h;
                ms[((wave * 9 + ps) * 16 + tile) * 32 + a_i] = acc[ps][8 * SB + r8];
            }
        __syncthreads();
        {
            const float* src = ms + e_lt * 32 + 4 * e_q;
            float sf[4] = {0.f, 0.f, 0.f, 0.f}, qf[4] = {0.f, 0.f, 0.f, 0.f};
            const int ox = e_ox0 + 4 * e_tx;
            f32x4 tm[2][6];
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                f32x4 m[6];
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) m[xi] = *reinterpret_cast<const f32x4*>(src + (xi * 6 + nu) * 512);
                xat6_rows(m, e_grp, tm[0][nu], tm[1][nu]);
            }
            // outputs leave as buffer stores: descriptor of the sub-tile's image + the (row, pixel) offset in SGPRs, ONE per-lane offset
            // (a 64-bit address per store cost six integer instructions, and on this path every VALU instruction is taken from the matrix pipe)
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)e_img * p.hout * p.wout * p.pitch_out, 0,
                                                                                   (int)((size_t)p.hout * p.wout * p.pitch_out * 4), 0x00020000);
            const int oy_t = e_oy0 + 4 * e_ty + 2 * e_grp;
            const unsigned ovoff = ((unsigned)(oy_t * p.wout + ox) * (unsigned)p.pitch_out + (unsigned)co) * 4u;
            const unsigned prow = (unsigned)p.wout * (unsigned)p.pitch_out * 4u, ppix = (unsigned)p.pitch_out * 4u;
            const bool inside = e_valid && e_oy0 + 16 <= p.hout && e_ox0 + 16 <= p.wout;      // block-uniform: no per-store bounds checks
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                f32x4 y[4];
                xat6(tm[ii], y);
                const int oy = oy_t + ii;
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    f32x4 v = y[jx] + bias4;
                    if (inside || (e_valid && oy < p.hout && ox + jx < p.wout)) {
                        if (p.addend) {
                            const size_t gp = ((size_t)e_img * p.hout + oy) * p.wout + ox + jx;
                            v += *reinterpret_cast<const f32x4*>(p.addend + gp * p.pitch_add + co);
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4w, v), orsrc, ovoff, (unsigned)ii * prow + (unsigned)jx * ppix, 0);
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
                // a thread's 8 values per channel and the 8 lanes (tiles) of the wave that hold the same channel quad are summed in fp32
                // (<= 64 values per partial sum), everything from there on is fp64
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
        if (p.stats && tid < 64 && e_valid) {
            const int c = tid >> 1, k = tid & 1;
            const double v = (st_s[(0 * 32 + c) * 2 + k] + st_s[(1 * 32 + c) * 2 + k]) + (st_s[(2 * 32 + c) * 2 + k] + st_s[(3 * 32 + c) * 2 + k]);
            atomicAdd(p.stats + ((size_t)e_img * p.cout + nt0 * 16 + c) * 2 + k, v);
        }
    };
    pass(std::integral_constant<int, 0>());
    pass(std::integral_constant<int, 1>());
}

int launch_conv_wino4x(hipStream_t s, const ConvSrc& a, const float* wpack_wino4x, const float* bias, float* out, int pitch_out, int cout,
                       double* stats, int n, int h, int w, const ConvOutExtra* ex) {
    CS_REQUIRE(a.c % 32 == 0 && a.c > 0 && cout % 32 == 0, "conv(winograd 4x4, 32x32): channels (%d -> %d) unsupported", a.c, cout);
    CS_REQUIRE((a.pitch % 4) == 0 && (pitch_out % 4) == 0, "conv(winograd 4x4, 32x32): pixel pitch must be a multiple of 4 floats");
    CS_REQUIRE(!a.h16 && !a.q8 && !a.ptr2, "conv(winograd 4x4, 32x32): plain fp32 activations only");
    CS_REQUIRE((size_t)h * w * a.pitch * 4 < ((size_t)1 << 32), "conv(winograd 4x4, 32x32): one image's activation tensor must stay below 4 GB");
    CS_REQUIRE((size_t)h * w * pitch_out * 4 < ((size_t)1 << 32), "conv(winograd 4x4, 32x32): one image's output tensor must stay below 4 GB");
    ConvParams p{};
    p.a = SrcDev{a.ptr, a.tab, a.slope, a.c, a.pitch, a.tab_c ? a.tab_c : a.c, a.tab_off};
    p.wpack = wpack_wino4x;
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
        CS_REQUIRE(!ex->out2, "conv(winograd 4x4, 32x32): the output split is not supported");
        p.addend = ex->addend;
        p.pitch_add = ex->pitch_add;
    }
    const int n_spatial = p.tiles_x * p.tiles_y * n;
    const double flops = 2.0 * n * (double)ceil_div(h, 4) * ceil_div(w, 4) * 36.0 * (double)cout * a.c;
    const double bytes = 4.0 * n * (double)h * w * (a.c + cout);
    ProfScope prof(K_CONV3_WINO4, s, flops, bytes);
    // opt-in to > 64 KB of dynamic LDS: an attribute of the function ON THE CURRENT DEVICE, so it is remembered per (kernel, device)
    static std::atomic<unsigned long long> lds_set[6];        // (launches may come from several host threads: fetch_or, not a read-modify-write)
    int dev = 0;
    CS_HIP(hipGetDevice(&dev));
    CS_REQUIRE(dev >= 0 && dev < 64, "conv(winograd 4x4, 32x32): device index %d out of range", dev);
    const int form = wino4_form(a.c, cout);
    CS_REQUIRE(form != 0, "conv(winograd 4x4, 32x32): the layer %d -> %d has no F(4x4) form", a.c, cout);
    const bool wide = form != 2;               // 64-channel blocks of 512 threads; else the 32-channel form (256 threads, two blocks per CU)
    const size_t lds = wide ? X_LDS_BYTES : Z_LDS_BYTES;
    const dim3 grid(wide ? xcd_grid((n_spatial + 1) / 2, p.ntile_total / 4) : xcd_grid((n_spatial + 1) / 2, p.ntile_total / 2));
    auto ensure = [&](const void* kern, int which) -> int {
        if (!((lds_set[which].load(std::memory_order_relaxed) >> dev) & 1ull)) {
            CS_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            lds_set[which].fetch_or(1ull << dev, std::memory_order_relaxed);
        }
        return CS_OK;
    };
    if (form == 3) {                           // 16x16x4 MFMAs, a wave = channel tile x sub-tile x all positions, output transform in registers
        // 256-thread blocks of ONE sub-tile (the 512-thread block of two sub-tiles measured 346 vs 323 us on the 64 -> 64 layer: DESIGN 3.1a'')
        static std::atomic<unsigned long long> lds_set_y[2];
        const size_t lds_y = (size_t)(X_RAW1 + 36 * 16 * 16) * 4;
        const dim3 grid_y(xcd_grid(n_spatial, p.ntile_total / 4));
        const int which = a.tab ? 1 : 0;
        const void* kern = a.tab ? reinterpret_cast<const void*>(&conv_wino4y_kernel<true, 1>) : reinterpret_cast<const void*>(&conv_wino4y_kernel<false, 1>);
        if (!((lds_set_y[which].load(std::memory_order_relaxed) >> dev) & 1ull)) {
            CS_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_y));
            lds_set_y[which].fetch_or(1ull << dev, std::memory_order_relaxed);
        }
        if (a.tab) hipLaunchKernelGGL((conv_wino4y_kernel<true, 1>), grid_y, dim3(256), lds_y, s, p);
        else hipLaunchKernelGGL((conv_wino4y_kernel<false, 1>), grid_y, dim3(256), lds_y, s, p);
    } else if (wide) {
        if (a.tab) {
            if (int rc = ensure(reinterpret_cast<const void*>(&conv_wino4x_kernel<true>), 1)) return rc;
            hipLaunchKernelGGL(conv_wino4x_kernel<true>, grid, dim3(512), lds, s, p);
        } else {
            if (int rc = ensure(reinterpret_cast<const void*>(&conv_wino4x_kernel<false>), 0)) return rc;
            hipLaunchKernelGGL(conv_wino4x_kernel<false>, grid, dim3(512), lds, s, p);
        }
    } else {
        if (a.tab) {
            if (int rc = ensure(reinterpret_cast<const void*>(&conv_wino4z_kernel<true>), 3)) return rc;
            hipLaunchKernelGGL(conv_wino4z_kernel<true>, grid, dim3(256), lds, s, p);
        } else {
            if (int rc = ensure(reinterpret_cast<const void*>(&conv_wino4z_kernel<false>), 2)) return rc;
            hipLaunchKernelGGL(conv_wino4z_kernel<false>, grid, dim3(256), lds, s, p);
        }
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ---- what decides whether a layer takes the F(4x4) form (was kernels_conv_wino4.hip until round 6) ----
// An inference packing exists (engine.hip: ww4_fresh), >= 32 input channels, channel counts in multiples of 32, and a map of at least
// ONE full 16 x 16 sub-tile: on smaller maps most of a block's 16 Winograd tiles are padding, and behind an InstanceNorm over a handful
// of pixels the form's rounding is amplified (a 32 x 32 window on the 5-level net, 2 x 2 pixels at the bottom, came out 1.3e-4 from the
// oracle in the round-5 randomised sweep); such layers run on the F(2x2) kernel.  Precision of the form: logits land 2.1e-5 ... 3.8e-5
// from an fp64 evaluation on random data (direct fp32: 5e-6 ... 8e-6; tests/diagnostics/winograd_f4_precision.py), and
// tests/test_gpu_blank_windows.py holds the margins on blank / constant / sparse windows.
size_t packed_wino4_floats(int cout_total, int cin_total) { return (size_t)(cin_total / 16) * 36 * (cout_total / 16) * 256; }

bool wino4_fits(int cin, int cout, int h, int w) { return cin >= WINO4_MIN_CIN && cin % 32 == 0 && cout % 32 == 0 && h >= 16 && w >= 16; }

int launch_conv_wino4(hipStream_t s, const ConvSrc& a, const float* wpack_wino4, const float* bias, float* out, int pitch_out, int cout,
                      double* stats, int n, int h, int w, const ConvOutExtra* ex) {
    CS_REQUIRE(wino4_form(a.c, cout) != 0, "conv(winograd 4x4): channels (%d -> %d) have no F(4x4) kernel (wino4_fits)", a.c, cout);
    return launch_conv_wino4x(s, a, wpack_wino4, bias, out, pitch_out, cout, stats, n, h, w, ex);
}

}  // namespace cs
