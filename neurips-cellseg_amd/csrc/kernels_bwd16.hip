// gfx950 (MI355X): weight gradients of the 3x3 convolutions on the 16-bit matrix pipe (v_mfma_f32_16x16x32_f16 / _bf16).
//
// Replaces the same autograd work as wgrad_mfma_kernel (kernels_bwd.hip): `loss.backward()` through every conv2d /
// conv_transpose2d of the U-Net (reference: baseline/model_training_3class.py:260).  The fp32 kernel contracts 4 pixels per matrix
// instruction at the fp32 vector rate (157 TFLOP/s); these contract 32 pixels per instruction at the 16-bit rate (2.5 PFLOP/s):
//   MODE 0 "fp16x3"  both operands split into two fp16 planes h0 = fp16(s v), h1 = fp16(s v - h0) (22 mantissa bits) of a
//                    power-of-two pre-scaled value, three terms h0 h0 + h0 h1 + h1 h0 in ONE fp32 accumulator: fp32-grade
//                    (every fp16 x fp16 product is exact in fp32).  Activations use the forward's static scale 2^4; GRADIENT
//                    operands span many decades between layers, so their scale is dynamic: 2^(13 - floor(log2 max|dy|)) from a
//                    device scalar the producer of dy maintains (norm backward: fused block maximum; residual-branch gradients:
//                    launch_amax) - the largest element lands in [2^13, 2^14), elements 2^-14 of it and smaller lose relative,
//                    not absolute, precision.  The unpack pass divides the scales back out (launch_unpack_wgrad's `unscale`).
//   MODE 1 "bf16"    one bf16 plane, one term, no scaling (BASELINE configs[1]'s flavour: bf16 arithmetic, fp32 master weights).
//
// dw[tap][co][ci] = sum over pixels of A[pix][co] * B[pix * S + tap - 1][ci]:  M = 16 co, N = 16 ci, K = 32 PIXELS per instruction.
// Both operands need 8 consecutive pixels of one channel per lane, i.e. the transpose of the NHWC tensors:
//   * A (dy-like, raw): straight from global memory into registers - lane (channel m, pixel group g) loads its 8 pixels as 8 dword
//     loads whose 16 lanes per pixel cover one 64-byte line; converted to planes in registers; never touches LDS.
//   * B (x-like, lazy InstanceNorm + PReLU applied while staging): the halo'd patch goes once to LDS as [pixel][16 ch] fp16 planes
//     (8-byte stores, the natural NHWC order) and is read back TRANSPOSED with ds_read_b64_tr_b16: per 16-lane group a block of 4
//     pixel rows x 16 channels comes back channel-major, two reads per 8-pixel fragment; the 9 taps and stride 2 are address
//     offsets (every row of a block has its own address).  Pixel pitch 32 B (stride 1) / 48 B (stride 2): conflict-free.
// Block = 4 waves on an 8 x 16 pixel tile (4 K-steps of 2 rows): waves (wk, wm) split the K-steps and the co tiles - a B fragment
// then feeds 3 * MTW MFMAs, and 9 * MTW accumulator tiles stay in registers for the block's whole split-K slice; the wk halves are
// added through LDS once, at the end of the slice.  Same split-K partial layout as the fp32 kernel: [split_k][tap][co][ci].
#include "kernels.h"

namespace cs {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Wgrad16Params {
    const float* a;
    int a_pitch;
    int a16, b16;                 // the operand holds bf16 elements (an activation tensor of the bf16 tape; pitches count elements)
    const float* a_amax;          // device scalar max|a| -> dynamic scale, or null: static scale (F16_ACT_SCALE)
    const float* b_ptr;
    const float2* b_tab;
    const float* b_slope;
    int b_pitch, b_tab_c, b_tab_off;
    const float* b_amax;          // as a_amax
    const float* dummy;
    float* part;
    int co_total, ci_total, co_off, ci_off;
    int n, ho, wo, hi, wi;
    int tiles_x, tiles_y, split_k, total_tiles;
};

template <int MODE>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (MODE == 0) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// two fp32 values -> one register of two 16-bit values of plane `pl` (low half = first value); MODE 0 leaves the remainder in v
template <int MODE>
__device__ __forceinline__ unsigned pack2(float& v0, float& v1) {
    if constexpr (MODE == 0) {
        const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;      // saturating: the kernel sets MODE.FP16_OVFL
        v0 -= (float)h0;
        v1 -= (float)h1;
        return (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    } else {
        return f32_to_bf16_bits(v0) | (f32_to_bf16_bits(v1) << 16);
    }
}
}  // namespace

__device__ __forceinline__ float prelu_f(float v, float slope) { return v >= 0.f ? v : v * slope; }

// A16 / B16: that operand is an ACTIVATION tensor of the bf16 tape (bf16 elements in HBM; plain-bf16 mode only, and never both: the
// other operand is the fp32 gradient).  Compile-time, so that the load groups stay straight-line code.
// ALDS (plain-bf16 mode, round 5): the A tile (8 x 16 pixels x 16 MT channels) takes the B operand's route - coalesced 16-byte loads of four
// channels of a pixel, converted once, 8-byte stores into [pixel][16 ch] bf16 planes in LDS, fragments read back TRANSPOSED
// (ds_read_tr16_b64) - instead of eight dword loads + conversions per fragment and lane straight from global memory (32 load
// instructions per lane and tile against 36 MFMAs).  Same values in the same MFMA order: bitwise the register form (CS_WGRAD_ALDS=0).
template <int S, int MT, int MODE, bool A16 = false, bool B16 = false, bool ALDS = false>
__global__ __launch_bounds__(256, 2) void wgrad16_kernel(Wgrad16Params p) {
    static_assert(!ALDS || MODE == 1, "the LDS route of the A operand is built for the one-plane (bf16) mode");
    constexpr int WM = MT >= 2 ? 2 : 1, WK = 4 / WM, MTW = MT / WM, KSW = 4 / WK;   // waves along co / along K, co tiles and K-steps per wave
    constexpr int NP = MODE == 0 ? 2 : 1;
    constexpr int TH = 8, TW = 16;
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int PITCH = S == 1 ? 32 : 48;                   // bytes per staged pixel (16 halfs + padding for the stride-2 reads)
    constexpr int PLANE = IH * IW * PITCH;
    constexpr int B_ITERS = (IH * IW + 63) / 64;
    constexpr int RED_BYTES = (WK - 1) * WM * 9 * MTW * 1024;
    constexpr int A_PLANE = TH * TW * 32;                     // one 16-channel tile of the A operand: [128 pixels][16 bf16]
    constexpr int A_BYTES = ALDS ? MT * A_PLANE : 0;
    constexpr int A_ITERS = ALDS ? 2 * MT : 1;                // 16-byte (8-byte: A16) loads per thread for the A tile
    constexpr int SMEM = NP * PLANE + A_BYTES > RED_BYTES ? NP * PLANE + A_BYTES : RED_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* const a_smem = smem + NP * PLANE;

    if constexpr (MODE == 0) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);   // MODE.FP16_OVFL: saturate, never inf

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wk = wave / WM;
    const int m16 = lane & 15, gi = lane >> 4;
    const int ks_id = blockIdx.x;
    const int co0 = blockIdx.y * 16 * MT;
    const int cb0 = blockIdx.z * 16;
    const int q = tid & 3;
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const float sa = MODE == 0 ? f16_dyn_scale(p.a_amax) : 1.f, sb = MODE == 0 ? f16_dyn_scale(p.b_amax) : 1.f;

    f32x4 acc[9][MTW];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int t = 0; t < MTW; ++t) acc[t9][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool lazy = p.b_tab != nullptr;
    float araw[ALDS ? 1 : KSW][ALDS ? 1 : MTW][8];
    f32x4 ra4[(ALDS && !A16) ? A_ITERS : 1];               // ALDS: the thread's share of the A tile, (pixel, channel quad) items
    uint2 ra2[(ALDS && A16) ? A_ITERS : 1];
    unsigned ok_a = 0;                       // bit kq * 8 + j: pixel j of K-step kq is inside the image
    f32x4 rb[B_ITERS], t0, t1;
    float slope = 0.f;
    unsigned ok_b = 0;

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
        if constexpr (ALDS) {
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) {
                const int i = tid + 256 * it;
                const int px = i / (4 * MT), c4 = i - px * (4 * MT);
                const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
                const bool ok = oy < p.ho && ox < p.wo;
                const size_t off = ok ? (((size_t)img * p.ho + oy) * p.wo + ox) * p.a_pitch + co0 + 4 * c4 : 0;
                if constexpr (A16) ra2[it] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.a) + off);
                else ra4[it] = *reinterpret_cast<const f32x4*>(p.a + off);
                ok_a |= (ok ? 1u : 0u) << it;
            }
        } else {
#pragma unroll
            for (int kq = 0; kq < KSW; ++kq) {
                const int oy = oy0 + 2 * (wk * KSW + kq) + (gi >> 1);
                const int oxb = ox0 + 8 * (gi & 1);
    #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = oy < p.ho && oxb + j < p.wo;
                    const size_t off = ok ? (((size_t)img * p.ho + oy) * p.wo + oxb + j) * p.a_pitch + co0 + 16 * (wm * MTW) + m16 : (size_t)m16;
    #pragma unroll
                    for (int t = 0; t < MTW; ++t) {
                        if constexpr (A16) {
                            // bf16 elements: the dword that holds channels (m16 & ~1, m16 | 1) - the 16 lanes of a pixel read one 32-byte run as
                            // eight dwords, two lanes each - and this lane's half of it
                            const unsigned wd = reinterpret_cast<const unsigned*>(p.a)[(off + 16 * t) >> 1];
                            araw[kq][t][j] = __builtin_bit_cast(float, (m16 & 1) ? (wd & 0xffff0000u) : (wd << 16));
                        } else {
                            araw[kq][t][j] = p.a[off + 16 * t];
                        }
                    }
                    ok_a |= (ok ? 1u : 0u) << (kq * 8 + j);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int pix = (tid >> 2) + 64 * it;
            const int iy = pix / IW, ix = pix - iy * IW;
            const int gy = oy0 * S - 1 + iy, gx = ox0 * S - 1 + ix;
            const bool ok = pix < IH * IW && gy >= 0 && gy < p.hi && gx >= 0 && gx < p.wi;
            const size_t off = ok ? (((size_t)img * p.hi + gy) * p.wi + gx) * p.b_pitch + cb0 + 4 * q : 0;
            if constexpr (B16) rb[it] = bf16x4_to_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.b_ptr) + off));
            else rb[it] = *reinterpret_cast<const f32x4*>(p.b_ptr + off);
            ok_b |= (ok ? 1u : 0u) << it;
        }
    };
    // the patch as [pixel][16 channels] 16-bit planes: thread = (pixel, channel quad), one 8-byte store per plane
    auto store_patch = [&]() {
        const float sl = (lazy && p.b_slope) ? slope : 0.f;
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int pix = (tid >> 2) + 64 * it;
            if (pix < IH * IW) {
                f32x4 v = rb[it];
                if (lazy) {
                    v.x = prelu_f(v.x * t0.x + t0.y, sl);
                    v.y = prelu_f(v.y * t0.z + t0.w, sl);
                    v.z = prelu_f(v.z * t1.x + t1.y, sl);
                    v.w = prelu_f(v.w * t1.z + t1.w, sl);
                }
                if (!((ok_b >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};     // zero padding AFTER the activation, as in the forward
                float e[4] = {v.x * sb, v.y * sb, v.z * sb, v.w * sb};
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    u32x2 h;
                    h.x = pack2<MODE>(e[0], e[1]);
                    h.y = pack2<MODE>(e[2], e[3]);
                    *reinterpret_cast<u32x2*>(smem + pl * PLANE + pix * PITCH + 8 * q) = h;
                }
            }
        }
        if constexpr (ALDS) {
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) {
                const int i = tid + 256 * it;
                const int px = i / (4 * MT), c4 = i - px * (4 * MT);
                u32x2 h{0u, 0u};
                if ((ok_a >> it) & 1u) {
                    if constexpr (A16) {
                        h.x = ra2[it].x;
                        h.y = ra2[it].y;
                    } else {
                        float e[4] = {ra4[it].x, ra4[it].y, ra4[it].z, ra4[it].w};
                        h.x = pack2<MODE>(e[0], e[1]);
                        h.y = pack2<MODE>(e[2], e[3]);
                    }
                }
                *reinterpret_cast<u32x2*>(a_smem + (c4 >> 2) * A_PLANE + px * 32 + 8 * (c4 & 3)) = h;
            }
        }
    };
    // this wave's A fragments of the staged tile: [K-step][co tile][plane], 8 pixels of channel m16 per lane
    u32x4 afr[ALDS ? 1 : KSW][ALDS ? 1 : MTW][NP];
    auto convert_a = [&]() {
        if constexpr (ALDS) return;
#pragma unroll
        for (int kq = 0; kq < KSW; ++kq)
#pragma unroll
            for (int t = 0; t < MTW; ++t) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ((ok_a >> (kq * 8 + j)) & 1u) ? araw[kq][t][j] * sa : 0.f;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int r = 0; r < 4; ++r) afr[kq][t][pl][r] = pack2<MODE>(v[2 * r], v[2 * r + 1]);
            }
    };

    // transposed-read address of this lane inside the patch (K-step 0, tap (0, 0), first half of the fragment): lane 4 q4 + p4 of a
    // 16-lane group supplies row q4 (= pixel 8 (gi & 1) + q4 of output row gi >> 1), channels 4 p4 .. 4 p4 + 3
    const int q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int lane_base = (((gi >> 1) * S * IW + (8 * (gi & 1) + q4) * S) * PITCH) + 8 * p4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto bfrag = [&](int base, int tap, int pl) -> u32x4 {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int off = (ky * IW + kx) * PITCH + pl * PLANE;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + base + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + base + off + 4 * S * PITCH));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        return u32x4{l2.x, l2.y, h2.x, h2.y};
    };

    // ALDS: the A fragment of (K-step kq, co tile t) out of its plane: pixel (row 2 (wk KSW + kq) + (gi >> 1), column 8 (gi & 1) + j)
    const int a_lane_base = (((gi >> 1) * TW + 8 * (gi & 1) + q4) * 32) + 8 * p4;
    auto afrag = [&](int kq, int t) -> u32x4 {
        const int base = (wm * MTW + t) * A_PLANE + a_lane_base + (2 * (wk * KSW + kq) * TW) * 32;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a_smem + base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a_smem + base + 4 * 32));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        return u32x4{l2.x, l2.y, h2.x, h2.y};
    };

    const int t_begin = (int)((int64_t)ks_id * p.total_tiles / p.split_k);
    const int t_end = (int)((int64_t)(ks_id + 1) * p.total_tiles / p.split_k);
    if (t_begin < t_end) {
        load_tile(t_begin);
        store_patch();
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        const bool more = tile + 1 < t_end;
        convert_a();                                   // frees the raw registers for the next tile's loads
        if (more) load_tile(tile + 1);
#pragma unroll
        for (int kq = 0; kq < KSW; ++kq) {
            const int base = lane_base + (2 * (wk * KSW + kq) * S * IW) * PITCH;
            u32x4 aq[MTW];                                  // (ALDS) this K-step's A fragments, read once for the nine taps
            if constexpr (ALDS) {
#pragma unroll
                for (int t = 0; t < MTW; ++t) aq[t] = afrag(kq, t);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {               // one kernel row = 3 taps x MTW independent accumulator chains
                u32x4 bq[3][NP];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) bq[kx][pl] = bfrag(base, ky * 3 + kx, pl);
                if constexpr (MODE == 0) {
                    // smallest terms first, the dominant h0 * h0 last; consecutive MFMAs never share an accumulator
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int t = 0; t < MTW; ++t) {
                                const int pa = term == 1 ? 1 : 0, pb = term == 0 ? 1 : 0;
                                acc[ky * 3 + kx][t] = mfma16<MODE>(afr[kq][t][pa], bq[kx][pb], acc[ky * 3 + kx][t]);
                            }
                } else {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int t = 0; t < MTW; ++t) {
                            if constexpr (ALDS) acc[ky * 3 + kx][t] = mfma16<MODE>(aq[t], bq[kx][0], acc[ky * 3 + kx][t]);
                            else acc[ky * 3 + kx][t] = mfma16<MODE>(afr[kq][t][0], bq[kx][0], acc[ky * 3 + kx][t]);
                        }
                }
            }
        }
        __syncthreads();
        if (more) {
            store_patch();
            __syncthreads();
        }
    }
    // add the K halves of the block: waves wk > 0 hand their tiles over through LDS (the patch is dead: every path above ends in a barrier)
    if constexpr (WK > 1) {
        f32x4* red = reinterpret_cast<f32x4*>(smem);
        if (wk > 0) {
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
                for (int t = 0; t < MTW; ++t) red[((((wk - 1) * WM + wm) * 9 + t9) * MTW + t) * 64 + lane] = acc[t9][t];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int o = 0; o < WK - 1; ++o)
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
                    for (int t = 0; t < MTW; ++t) acc[t9][t] += red[(((o * WM + wm) * 9 + t9) * MTW + t) * 64 + lane];
        }
    }
    if (wk == 0) {                                   // C layout of the 16x16 tile: row (co) = 4 * (lane >> 4) + r, column (ci) = lane & 15
        float* part = p.part + (size_t)ks_id * 9 * p.co_total * p.ci_total;
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
            for (int t = 0; t < MTW; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    part[((size_t)t9 * p.co_total + p.co_off + co0 + 16 * (wm * MTW + t) + gi * 4 + r) * p.ci_total + p.ci_off + cb0 + m16] = acc[t9][t][r];
    }
}

int wgrad16_split_k(int a_c, int b_c, int n, int ho, int wo) {
    const int mt = (a_c % 64 == 0) ? 4 : (a_c % 32 == 0) ? 2 : 1;
    const int combos = (a_c / (16 * mt)) * (b_c / 16);
    const int total_tiles = n * ceil_div(wo, 16) * ceil_div(ho, 8);
    int k = ceil_div(512, combos);
    if (k > total_tiles) k = total_tiles;
    return k < 1 ? 1 : k;
}

int launch_wgrad16(hipStream_t s, int mode, int stride, const float* a, int a_pitch, int a_c, const float* a_amax, const ConvSrc& b,
                   const float* b_amax, float* part, int split_k, int co_total, int ci_total, int co_off, int ci_off, int n, int ho, int wo,
                   int hi, int wi, bool a_h16) {
    CS_REQUIRE(a_c % 16 == 0 && a_c > 0 && b.c % 16 == 0 && b.c > 0, "wgrad16: %d x %d channels (multiples of 16)", a_c, b.c);
    CS_REQUIRE(stride == 1 || stride == 2, "wgrad16: stride %d", stride);
    CS_REQUIRE(mode == 0 || mode == 1, "wgrad16: mode %d", mode);
    CS_REQUIRE(!b.ptr2 && !b.q8, "wgrad16: fp32 or bf16 sources only");
    CS_REQUIRE(mode == 1 || (!b.h16 && !a_h16), "wgrad16: the bf16 tape belongs to the plain-bf16 mode");
    Wgrad16Params p{};
    p.a = a; p.a_pitch = a_pitch; p.a_amax = a_amax; p.a16 = a_h16 ? 1 : 0; p.b16 = b.h16 ? 1 : 0;
    p.b_ptr = b.ptr; p.b_tab = b.tab; p.b_slope = b.slope; p.b_pitch = b.pitch; p.b_tab_c = b.tab_c ? b.tab_c : b.c; p.b_tab_off = b.tab_off;
    p.b_amax = b_amax;
    p.dummy = a;
    p.part = part; p.co_total = co_total; p.ci_total = ci_total; p.co_off = co_off; p.ci_off = ci_off;
    p.n = n; p.ho = ho; p.wo = wo; p.hi = hi; p.wi = wi;
    p.tiles_x = ceil_div(wo, 16);
    p.tiles_y = ceil_div(ho, 8);
    p.total_tiles = n * p.tiles_x * p.tiles_y;
    CS_REQUIRE(split_k >= 1 && split_k <= p.total_tiles, "wgrad16: split_k %d out of range (%d tiles)", split_k, p.total_tiles);
    p.split_k = split_k;
    const int mt = (a_c % 64 == 0) ? 4 : (a_c % 32 == 0) ? 2 : 1;
    dim3 grid((unsigned)split_k, (unsigned)(a_c / (16 * mt)), (unsigned)(b.c / 16));
    const double flops = 2.0 * n * ho * wo * (double)a_c * b.c * 9;
    ProfScope prof(K_WGRAD, s, flops, 4.0 * n * ((double)ho * wo * a_c + (double)hi * wi * b.c));
#define CS_WG16(S_, MODE_, A_, B_, L_)                                                                                  \
    switch (mt) {                                                                                                      \
        case 4: hipLaunchKernelGGL((wgrad16_kernel<S_, 4, MODE_, A_, B_, L_>), grid, dim3(256), 0, s, p); break;       \
        case 2: hipLaunchKernelGGL((wgrad16_kernel<S_, 2, MODE_, A_, B_, L_>), grid, dim3(256), 0, s, p); break;       \
        default: hipLaunchKernelGGL((wgrad16_kernel<S_, 1, MODE_, A_, B_, L_>), grid, dim3(256), 0, s, p); break;      \
    }
    CS_REQUIRE(!(a_h16 && b.h16), "wgrad16: one operand is the gradient (fp32), at most one comes off the bf16 tape");
    CS_REQUIRE(!a_h16 || (a_pitch % 4 == 0 && a_c % 4 == 0), "wgrad16: a bf16 A operand needs a pixel pitch that is a multiple of 4");
    static const bool alds = !(getenv("CS_WGRAD_ALDS") && atoi(getenv("CS_WGRAD_ALDS")) == 0);      // A/B: 0 = A fragments straight from global memory
    const bool l = alds && mode == 1 && a_pitch % 4 == 0;
    if (mode == 0) {
        if (stride == 1) { CS_WG16(1, 0, false, false, false) } else { CS_WG16(2, 0, false, false, false) }
    } else if (a_h16) {
        if (l) { if (stride == 1) { CS_WG16(1, 1, true, false, true) } else { CS_WG16(2, 1, true, false, true) } }
        else { if (stride == 1) { CS_WG16(1, 1, true, false, false) } else { CS_WG16(2, 1, true, false, false) } }
    } else if (b.h16) {
        if (l) { if (stride == 1) { CS_WG16(1, 1, false, true, true) } else { CS_WG16(2, 1, false, true, true) } }
        else { if (stride == 1) { CS_WG16(1, 1, false, true, false) } else { CS_WG16(2, 1, false, true, false) } }
    } else {
        if (l) { if (stride == 1) { CS_WG16(1, 1, false, false, true) } else { CS_WG16(2, 1, false, false, true) } }
        else { if (stride == 1) { CS_WG16(1, 1, false, false, false) } else { CS_WG16(2, 1, false, false, false) } }
    }
#undef CS_WG16
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// max |v| of a dense tensor into a device float (bits of a non-negative float order like unsigned integers: atomicMax on the bits).
// `out` must be zero before the first launch that accumulates into it.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void amax_kernel(const float* src, int64_t n4, unsigned* out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    amax_commit(out, m);
}

int launch_amax(hipStream_t s, const float* src, int64_t numel, float* out) {
    CS_REQUIRE(numel % 4 == 0, "amax: %lld elements (multiple of 4)", (long long)numel);
    const int64_t n4 = numel / 4;
    int blocks = (int)((n4 + 1023) / 1024);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    ProfScope prof(K_COLSUM, s, 0, 4.0 * numel);
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, n4, reinterpret_cast<unsigned*>(out));
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
