// gfx950 split-bf16 convolution kernels: fp32-grade results at bf16 MFMA speed.
//
// Every fp32 operand x is split into NP bf16 "planes" (x = p0 + p1 + p2 + ..., each pk = bf16(x - p0 - ... - p(k-1)));
// a product a*b is evaluated as the NP = 3 : a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1   (6 MFMAs, error ~2^-24: fp32 grade)
//                               NP = 2 : a0b0 + a0b1 + a1b0                           (3 MFMAs, error ~2^-16)
//                               NP = 1 : a0b0                                          (plain bf16)
// on v_mfma_f32_16x16x32_bf16 (K = 32 per instruction, fp32 accumulate).  Every bf16 x bf16 product is exact in
// fp32, so NP = 3 differs from the fp32-MFMA kernel only by accumulation order - measured logit error ~1e-5 against
// the torch-CPU oracle, inside the 1e-4 budget - while costing 6/16 of its matrix-pipe cycles.
//
// Structure:
//   * K loop over 32-channel chunks; the input patch is staged ONCE per chunk as NP bf16 planes in LDS (lazy
//     InstanceNorm + PReLU applied before the split), next chunk's global loads are in flight under the MFMAs;
//   * weights are pre-split at load time and stored in MFMA fragment order; every wave owns one 16-channel output tile,
//     so its B fragments are private: they go L2 -> registers as lane-linear 16-byte loads, a whole chunk ahead of
//     their use (ring of taps x planes fragments) and never touch LDS;
//   * epilogue as in the fp32 kernel (bias, fp64 statistics, full-line stores through an LDS slab).
#include <type_traits>

#include "kernels.h"

namespace cs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// NP = 4 selects the split-FP16 mode ("fp16x3"): two fp16 planes h0 = fp16(x'), h1 = fp16(x' - h0) - 22 mantissa bits - and
// the three terms h0*h0 + h0*h1 + h1*h0 on v_mfma_f32_16x16x32_f16, all in ONE accumulator.  Operands are pre-scaled by exact
// powers of two so that the residual plane stays out of fp16's subnormal range for every value that matters: weights by 2^8
// (|w| ~ 1e-2 -> residual ~ 1e-3), staged activations by 2^4 (residual normal for |x| >= 8e-3; below that the absolute error
// is <= 2^-29); the epilogue multiplies the sum by 2^-12.  Every fp16 x fp16 product is exact in fp32.  Operands must stay
// inside fp16's range after scaling (|x| < 4094, |w| < 255), which normalised activations and trained weights do.  Half the
// MFMAs of bf16x6 at the same fp32-grade accuracy class (dropped term 2^-22, operand truncation 2^-23).
constexpr float F16_W_SCALE = 256.f /* == PACK_F16_W_SCALE in pack_jobs.h */, F16_X_SCALE = 16.f, F16_OUT_SCALE = 1.f / (256.f * 16.f);
template <int NP>
struct SplitCfg {
    static constexpr bool f16 = NP == 4;
    static constexpr int planes = NP == 4 ? 2 : NP;          // planes staged in LDS / loaded per weight fragment
    static constexpr int terms = NP == 4 ? 3 : NP == 3 ? 6 : NP == 2 ? 3 : 1;
};
// term -> (plane of A, plane of B, accumulator: 0 main, 1 cross); smallest terms first, the dominant a0*b0 last
__device__ __forceinline__ constexpr int term_a(int term) { return (term == 0) ? 0 : (term == 1) ? 0 : (term == 2) ? 1 : (term == 3) ? 0 : (term == 4) ? 2 : 1; }
__device__ __forceinline__ constexpr int term_b(int term) { return (term == 0) ? 0 : (term == 1) ? 1 : (term == 2) ? 0 : (term == 3) ? 2 : (term == 4) ? 0 : 1; }
template <bool F16>
__device__ __forceinline__ f32x4 mfma_split(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// split 4 fp32 values into the next 16-bit plane (bits returned in a bf16x4 container) and leave the remainder in v
template <bool F16>
__device__ __forceinline__ bf16x4 split_plane(f32x4& v, int pn, float xscale = F16_X_SCALE) {
    if constexpr (F16) {
        f16x4 h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (pn == 0) v[j] *= xscale;
            h[j] = (_Float16)v[j];                           // saturates at +-65504 (MODE.FP16_OVFL is set by the fp16 kernels)
            v[j] -= (float)h[j];
        }
        return __builtin_bit_cast(bf16x4, h);
    } else {
        bf16x4 h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = (__bf16)v[j];
            v[j] -= (float)h[j];
        }
        (void)pn;
        return h;
    }
}
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));   // 4 bf16 activations as loaded from HBM (bf16 datapath)
__device__ __forceinline__ f32x4 bf16x4_to_f32(u32x2 v) { return cs::bf16x4_to_f32(make_uint2(v.x, v.y)); }
__device__ __forceinline__ u32x2 f32x4_to_bf16v(f32x4 v) {
    const uint2 r = cs::f32x4_to_bf16(v);
    return u32x2{r.x, r.y};
}

struct SrcDevB {
    const float* ptr;
    const float2* tab;
    const float* slope;
    int c, pitch, tab_c, tab_off;
    const float* ptr2;             // second tensor of a SUM source (ConvSrc::ptr2; the transposed conv's `b` operand only)
    const float2* tab2;
    const float* slope2;
};

struct ConvBParams {
    SrcDevB a, b;
    const unsigned short* wpack;   // [chunk][tap][ntile_total][3 planes][64 lanes][8] bf16
    const float* bias;
    float* out;
    double* stats;
    const float* dummy;
    int pitch_out, cout, ntile_total;
    int n, hin, win, hout, wout, tiles_x, tiles_y;
    // data-gradient use of the stride-1 kernel (training on the 16-bit pipe): the input is a raw GRADIENT tensor whose power-of-two
    // scale is dynamic (AMAX_SLOTS device floats, see f16_dyn_scale; null: the static activation scale), and the epilogue can add a
    // tensor to the result (the other branch's gradient)
    const float* in_amax;
    const float* addend;
    int pitch_add;
    float* out2;                   // channel tiles >= split_nt go to out2 (pitch_out2), the data gradient of a transposed conv splits
    int pitch_out2, split_nt;      // into its skip and its deeper input this way (stride-2 kernel, fp32 outputs only)
};


// XCD-aware work decode.  Blocks are dealt round-robin over the 8 XCDs, so ids b and b + 8 share an L2.  The 1-D grid is
// laid out as [spatial group of 8 tiles][channel tile][8]: the blocks that read the SAME input tile for different output
// channels have ids 8 apart (same XCD, dispatched back to back) and hit L2 instead of re-fetching the tile from HBM.
struct WorkId {
    int spatial, ct;
    bool valid;
};
__device__ __forceinline__ WorkId decode_work(int n_spatial, int n_ct) {
    const int id = blockIdx.x, lane8 = id & 7, rest = id >> 3;
    const int ct = rest % n_ct, sg = rest / n_ct;
    const int spatial = sg * 8 + lane8;
    return WorkId{spatial, ct, spatial < n_spatial};
}
static inline unsigned xcd_grid(int n_spatial, int n_ct) { return (unsigned)(((n_spatial + 7) / 8) * 8 * n_ct); }

__device__ __forceinline__ float prelu_h(float v, float slope) { return v >= 0.f ? v : v * slope; }

// The split modes (fp32-grade results) sum every value in fp64.  The bf16 datapath, whose outputs are rounded to 8 mantissa bits
// anyway, sums one lane's values (<= 32 per channel) in fp32 and only these partial sums in fp64: the fp64 conversions and FMAs
// of the epilogue were 15 % of the stride-1 kernels' time (1.2 ms per slide), the fp32 partials carry a relative error <= 2e-6.
__device__ __forceinline__ void stats_commit_b(double s, double q, double* dst, int lane) {
    s += __shfl_xor(s, 16);
    q += __shfl_xor(q, 16);
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    if (lane < 16) {
        atomicAdd(dst, s);
        atomicAdd(dst + 1, q);
    }
}
// Block-level form for kernels whose WM waves hold different pixel rows of the SAME channels: every wave leaves its sums in LDS
// (red[wm][channel of the block][2]) and, after the block's next barrier, stats_flush_b() issues ONE pair of atomics per channel.
// The device-scope fp64 atomics themselves were 0.5 ms per slide on the bf16 datapath (4 M per 32-channel layer launch).
__device__ __forceinline__ void stats_stash_b(double s, double q, double* red_c, int lane) {
    s += __shfl_xor(s, 16);
    q += __shfl_xor(q, 16);
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    if (lane < 16) {
        red_c[0] = s;
        red_c[1] = q;
    }
}
template <int WM, int NCH>
__device__ __forceinline__ void stats_flush_b(const double* red, double* dst, int tid) {
    if (tid < NCH * 2) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) v += red[w * NCH * 2 + tid];
        atomicAdd(dst + tid, v);
    }
}

// block = 4 waves arranged WN (along output channels) x WM = 4 / WN (along pixel rows); each wave owns NTW 16-channel
// tiles and MTW pixel rows.  A (pixels) comes from the LDS planes shared by the block; B (weights) is private to the wave
// and goes global -> registers, one tap ROW ahead (ring of KS x NTW x NP fragments), so no weight ever touches LDS.
// A16: the bf16 datapath - inputs and output are bf16 tensors in HBM (half the activation traffic); only with NP == 1.
template <int S, int KS, int WN, int NTW, int NP, int THX, bool A16>
__global__ __launch_bounds__(256, 2) void conv_bf16s_kernel(ConvBParams p) {
    static_assert(!A16 || NP == 1, "bf16 activations in HBM only make sense for the plain-bf16 mode");
    typedef typename std::conditional<A16, u32x2, f32x4>::type raw4;
    constexpr bool F16 = SplitCfg<NP>::f16;
    constexpr int NPL = SplitCfg<NP>::planes, NTERM = SplitCfg<NP>::terms;
    constexpr int WM = 4 / WN;
    constexpr int TH = THX * ((S == 1) ? 8 : 4), TW = 16;    // THX = 2: double-height tile, half the weight traffic per MFMA
    constexpr int MTW = TH / WM;                            // pixel rows per wave
    constexpr int NTB = WN * NTW;                           // 16-channel tiles per block
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int PITCH = (S == 1) ? 96 : 80;               // bytes per staged pixel and plane: 32 bf16 + pad (conflict-free b128 reads)
    constexpr int PAD = KS / 2, TAPS = KS * KS;
    constexpr int IN_ITEMS = IH * IW * 8;                   // (pixel, 4-channel quad) items per chunk
    constexpr int IN_ITERS = (IN_ITEMS + 255) / 256;
    constexpr int PLANE_BYTES = IH * IW * PITCH;
    constexpr int IN_BYTES = (NPL * PLANE_BYTES + 15) / 16 * 16;
    constexpr int EPS = 16 * NTB + 4;
    constexpr int EP_BYTES = TH * 16 * EPS * 4;
    constexpr bool DB = 2 * IN_BYTES <= EP_BYTES + 4096;    // double-buffered input where the epilogue slab already pays for the LDS
    constexpr int SM_BYTES = (DB ? 2 : 1) * IN_BYTES > EP_BYTES ? (DB ? 2 : 1) * IN_BYTES : EP_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SM_BYTES];
    __shared__ double red[WM > 1 ? WM * NTB * 32 : 1];      // per-wave statistics of the block's channels (stats_stash_b)

    // fp16 mode: an out-of-range operand saturates at +-65504 instead of becoming inf (MODE.FP16_OVFL, bit 23 of HW_REG_MODE):
    // an activation beyond the documented domain then costs accuracy in its window, not a NaN from the next InstanceNorm
    if constexpr (F16) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, g = lane >> 4;
    const int wn = wave % WN, wm = wave / WN;
    const WorkId wk = decode_work(p.tiles_x * p.tiles_y * p.n, p.ntile_total / NTB);
    if (!wk.valid) return;
    int bid = wk.spatial;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int nt_w = wk.ct * NTB + wn * NTW;                 // this wave's first output-channel tile
    const int q = tid & 7;                                   // this thread's channel quad inside the 32-channel chunk
    const float xscale = (F16 && p.in_amax) ? f16_dyn_scale(p.in_amax) : F16_X_SCALE;
    const float out_scale = 1.f / (F16_W_SCALE * xscale);

    constexpr int NACC = 1;
    f32x4 acc[NACC][MTW][NTW];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[k][m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int goff[IN_ITERS];
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (tid >> 3) + 32 * it;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = oy0 * S - PAD + iy, gx = ox0 * S - PAD + ix;
        goff[it] = (pix < IH * IW && gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win) ? gy * p.win + gx : -1;
    }

    const int chunks_a = (p.a.c + 31) >> 5, chunks = chunks_a + ((p.b.c + 31) >> 5);
    raw4 rin[IN_ITERS];
    f32x4 t0, t1;
    float slope = 0.f;
    bool lazy = false, qvalid = false;

    auto load_input = [&](int ch) {
        const bool from_a = ch < chunks_a;
        const SrcDevB& src = from_a ? p.a : p.b;
        const int c0 = (from_a ? ch : ch - chunks_a) << 5;
        qvalid = c0 + 4 * q < src.c;                        // a 16-channel tail leaves the upper half of the chunk empty
        const int cq = qvalid ? c0 + 4 * q : 0;
        lazy = src.tab != nullptr;
        const f32x4* t = lazy ? reinterpret_cast<const f32x4*>(src.tab + (size_t)img * src.tab_c + src.tab_off + cq)
                              : reinterpret_cast<const f32x4*>(p.dummy);
        t0 = t[0];
        t1 = t[1];
        const float* sp = (lazy && src.slope) ? src.slope : p.dummy;
        slope = *sp;
        if (!(lazy && src.slope)) slope = 0.f;
        const size_t base_el = (size_t)img * p.hin * p.win * src.pitch + cq;
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int o = goff[it] >= 0 ? goff[it] : 0;
            if constexpr (A16) rin[it] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(src.ptr) + base_el + (size_t)o * src.pitch);
            else rin[it] = *reinterpret_cast<const f32x4*>(src.ptr + base_el + (size_t)o * src.pitch);
        }
    };
    auto store_input = [&](char* in_s) {
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int pix = (tid >> 3) + 32 * it;
            if (IN_ITEMS % 256 == 0 || pix < IH * IW) {
                f32x4 v;
                if constexpr (A16) v = bf16x4_to_f32(rin[it]);
                else v = rin[it];
                if (lazy) {
                    v.x = prelu_h(v.x * t0.x + t0.y, slope);
                    v.y = prelu_h(v.y * t0.z + t0.w, slope);
                    v.z = prelu_h(v.z * t1.x + t1.y, slope);
                    v.w = prelu_h(v.w * t1.z + t1.w, slope);
                }
                if (goff[it] < 0 || !qvalid) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pn = 0; pn < NPL; ++pn)
                    *reinterpret_cast<bf16x4*>(in_s + pn * PLANE_BYTES + pix * PITCH + q * 8) = split_plane<F16>(v, pn, xscale);
            }
        }
    };
    // weight fragments of one tap row: KS x NTW x NP lane-linear 16-byte loads (1 KiB per wave-instruction, from L2)
    bf16x8 bq[KS][NTW][NPL];
    const bf16x8* wbase = reinterpret_cast<const bf16x8*>(p.wpack) + (size_t)nt_w * 3 * 64 + lane;
    auto load_b = [&](int ch, int ky, int kx) {
        const bf16x8* w = wbase + ((size_t)ch * TAPS + ky * KS + kx) * p.ntile_total * 3 * 64;
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int pn = 0; pn < NPL; ++pn) bq[kx][t][pn] = w[(t * 3 + pn) * 64];
    };

    load_input(0);
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) load_b(0, 0, kx);
    store_input(smem);
    __syncthreads();

    const int last_row = chunks * KS - 1;
    for (int ch = 0; ch < chunks; ++ch) {
        const bool more = ch + 1 < chunks;
        // UNCONDITIONAL loads (the last trip re-reads its own chunk; ring refills are clamped to the last tap row): a load
        // group that is skipped on one path makes hipcc size the vmcnt waits of the weight ring for that path, and the
        // MFMA section then stalls on loads it has only just issued.  sched_barriers pin the issue points.
        load_input(more ? ch + 1 : ch);
        __builtin_amdgcn_sched_barrier(0);
        const char* in_s = smem + (DB ? (ch & 1) * IN_BYTES : 0);
        constexpr int MG = (MTW % 2 == 0) ? 2 : 1;               // pixel rows per group: MG x NTW independent MFMA chains
        constexpr int NGRP = MTW / MG;
        // single-plane modes: the A fragments of group n + 1 (of this tap, or the first group of the next tap) are requested BEFORE the MFMAs of
        // group n and pinned there.  Left to the compiler every group's reads sat right in front of its own MFMAs - rr [wait] MM [wait] MM -
        // and each 64 cycles of matrix work waited for a fresh LDS round trip (matrix pipe 40 % busy at two waves per SIMD).
        constexpr bool PRE = NPL == 1;
        auto a_read = [&](int ky, int kx, int m0, bf16x8 (&dst)[MG][NPL]) {
#pragma unroll
            for (int mm = 0; mm < MG; ++mm)
#pragma unroll
                for (int pn = 0; pn < NPL; ++pn)
                    dst[mm][pn] = *reinterpret_cast<const bf16x8*>(in_s + pn * PLANE_BYTES + (((wm * MTW + m0 + mm) * S + ky) * IW + pl * S + kx) * PITCH + g * 16);
        };
        bf16x8 apre[2][MG][NPL];
        if constexpr (PRE) a_read(0, 0, 0, apre[0]);
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int row_next = ch * KS + ky + 1 < last_row ? ch * KS + ky + 1 : last_row;
            const int ch_n = row_next / KS, ky_n = row_next - ch_n * KS;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                bf16x8 b[NTW][NPL];
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int pn = 0; pn < NPL; ++pn) b[t][pn] = bq[kx][t][pn];
                // refill this ring slot one tap row ahead (first row of the next chunk after the last row)
                load_b(ch_n, ky_n, kx);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m0 = 0; m0 < MTW; m0 += MG) {
                    const int step = (ky * KS + kx) * NGRP + m0 / MG;
                    bf16x8 a[MG][NPL];
                    if constexpr (PRE) {
                        if (step + 1 < TAPS * NGRP) {
                            const int nstep = step + 1, ntap = nstep / NGRP;
                            a_read(ntap / KS, ntap % KS, (nstep % NGRP) * MG, apre[nstep & 1]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int mm = 0; mm < MG; ++mm) a[mm][0] = apre[step & 1][mm][0];
                    } else {
                        a_read(ky, kx, m0, a);
                    }
                    // smallest terms first, the dominant a0*b0 last; rows and channel tiles alternate inside a term so that
                    // consecutive MFMAs never depend on each other (a dependent 4-pass MFMA cannot issue back to back)
#pragma unroll
                    for (int term = NTERM - 1; term >= 0; --term) {
                        const int ia = term_a(term), ib = term_b(term);
                        const int ka = 0;
#pragma unroll
                        for (int mm = 0; mm < MG; ++mm)
#pragma unroll
                            for (int t = 0; t < NTW; ++t)
                                acc[ka][m0 + mm][t] = mfma_split<F16>(a[mm][ia], b[t][ib], acc[ka][m0 + mm][t]);
                        // pin the chains here: LLVM may otherwise sink the (side-effect free) MFMAs below the staging code
                        // (kernels_conv_fp8.hip: 2.7 KB of spills per lane when that happened)
#pragma unroll
                        for (int mm = 0; mm < MG; ++mm)
#pragma unroll
                            for (int t = 0; t < NTW; ++t) asm volatile("" : "+v"(acc[0][m0 + mm][t]));
                    }
                    if constexpr (PRE) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) asm volatile("" ::"v"(rin[it]));   // keeps the input loads above the MFMA section
        if (!DB) __syncthreads();
        if (more) store_input(smem + (DB ? ((ch + 1) & 1) * IN_BYTES : 0));
        __syncthreads();
    }

    // ---- epilogue: bias, fp64 statistics, then full-line stores through a block-wide LDS slab ----
    float* ep = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int co = (nt_w + t) * 16 + pl;
        const float bias = p.bias ? p.bias[co] : 0.f;
        double s = 0.0, qq = 0.0;
        float sf = 0.f, qf = 0.f;                            // bf16 datapath: this lane's <= 32 values are summed in fp32 (see stats_commit_b)
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int row = wm * MTW + m, oy = oy0 + row;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = ox0 + g * 4 + r;
                float v = acc[0][m][t][r];
                if constexpr (F16) v *= out_scale;
                v += bias;
                ep[(row * 16 + g * 4 + r) * EPS + (wn * NTW + t) * 16 + pl] = v;
                if (p.stats && oy < p.hout && ox < p.wout) {
                    if constexpr (A16) {
                        sf += v;
                        qf = fmaf(v, v, qf);
                    } else {
                        s += (double)v;
                        qq += (double)v * (double)v;
                    }
                }
            }
        }
        if constexpr (A16) {
            s = (double)sf;
            qq = (double)qf;
        }
        if (p.stats) {
            if constexpr (WM > 1) stats_stash_b(s, qq, red + ((wm * NTB + wn * NTW + t) * 16 + pl) * 2, lane);
            else stats_commit_b(s, qq, p.stats + ((size_t)img * p.cout + co) * 2, lane);
        }
    }
    __syncthreads();
    if constexpr (WM > 1)
        if (p.stats) stats_flush_b<WM, NTB * 16>(red, p.stats + ((size_t)img * p.cout + wk.ct * NTB * 16) * 2, tid);
    {
        constexpr int F4 = NTB * 4;                          // float4 per pixel record of this block
        const size_t el0 = (((size_t)img * p.hout + oy0) * p.wout + ox0) * p.pitch_out + wk.ct * NTB * 16;
#pragma unroll
        for (int it_ = 0; it_ < (TH * 16 * F4 + 255) / 256; ++it_) {      // (trip count visible to the unroller: the LDS reads of all passes go out together)
            const int i = tid + 256 * it_;
            if ((TH * 16 * F4) % 256 != 0 && i >= TH * 16 * F4) break;
            const int pix = i / F4, c4 = i - pix * F4;
            const int row = pix >> 4, x = pix & 15;
            if (oy0 + row < p.hout && ox0 + x < p.wout) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&ep[pix * EPS + c4 * 4]);
                const size_t el = el0 + ((int64_t)row * p.wout + x) * p.pitch_out + c4 * 4;
                if constexpr (A16) *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(p.out) + el) = f32x4_to_bf16v(v);
                else {
                    const size_t gp = ((size_t)img * p.hout + oy0 + row) * p.wout + ox0 + x;
                    const int ch = wk.ct * NTB * 16 + c4 * 4;
                    if (p.addend) v += *reinterpret_cast<const f32x4*>(p.addend + gp * p.pitch_add + ch);
                    if (p.out2 && ch >= p.split_nt * 16) *reinterpret_cast<f32x4*>(p.out2 + gp * p.pitch_out2 + ch - p.split_nt * 16) = v;
                    else *reinterpret_cast<f32x4*>(p.out + el) = v;
                }
            }
        }
    }
}

static SrcDevB to_devb(const ConvSrc& s) { return SrcDevB{s.ptr, s.tab, s.slope, s.c, s.pitch, s.tab_c ? s.tab_c : s.c, s.tab_off, s.ptr2, s.tab2, s.slope2}; }

template <int S, int KS, int WN, int NTW, int NP, int THX = 1, bool A16 = false>
static int launch_b(hipStream_t s, ConvBParams p) {
    constexpr int TH = THX * ((S == 1) ? 8 : 4);
    p.tiles_x = ceil_div(p.wout, 16);
    p.tiles_y = ceil_div(p.hout, TH);
    dim3 grid(xcd_grid(p.tiles_x * p.tiles_y * p.n, p.ntile_total / (WN * NTW)));
    hipLaunchKernelGGL((conv_bf16s_kernel<S, KS, WN, NTW, NP, THX, A16>), grid, dim3(256), 0, s, p);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// bf16 datapath, 3x3 stride 1, 32 -> 32 channels (the two widest tensors of the 32-base net: 128 x 128 x 32 per window), inference.
// conv_bf16s_kernel<1, 3, 1, 2, 1, 2> runs these layers as blocks of ONE K chunk: gather -> barrier -> 72 MFMAs per wave -> epilogue,
// nothing of a block overlaps anything else of it (13 % matrix busy, 2.5 TB/s for a layer whose floor is its 537 MB of traffic).
// Here a WAVE streams a strip of 16 output columns down RW_RS rows:
//   * its 9 x 2 weight fragments (18 x 16 bytes per lane) stay in registers for the whole strip;
//   * a rolling window of input rows lives in wave-private LDS (4 slots of 18 pixels x 96 bytes): per output row ONE new input row is
//     loaded (three rows ahead, in registers), normalised + activated + rounded to bf16 and stored one step before it is needed (beside
//     the MFMAs of the current row: the slot it goes to is the one row oy - 2 left) - no halo rows are staged twice, only
//     the 2 halo columns (18 / 16).  Those two columns of EIGHT rows are one 64-lane load and one activation pass (lane = row of the
//     group x pixel x channel octet) into a side area of 16 row slots; a row step copies its two pixels from there into the window
//     (8 lanes, LDS to LDS) - a per-row halo pass of 8 active lanes cost as many instructions as the 64-lane main pass;
//   * 18 MFMAs per row, A fragments as 16-byte reads of the window (conflict-free at 96 bytes per pixel);
//   * bias, statistics (fp32 per lane over the segment's <= 32 rows, fp64 from the lane groups on), and the row leaves through a wave-private slab as one 16-byte store
//     per lane (64 lanes = 16 pixels x 64 bytes);
//   * no block barrier until the statistics of the block's four strips meet at the end.
// Same products in the same order as conv_bf16s_kernel (tap-major, one K = 32 step per tap): the conv results are bit-identical, the
// statistics are summed in another order.
// ------------------------------------------------------------------------------------------------
constexpr int RW_RS = 32;                                   // output rows of a wave's strip segment
constexpr int RW_PITCH = 96;                                // bytes per staged pixel: 32 bf16 + pad
constexpr int RW_ROW = 18 * RW_PITCH;
constexpr int RW_SLOTS = 4;
constexpr int RW_EPS = 36;                                  // floats per slab pixel: 32 channels + pad
constexpr int RW_HALO = 16 * 2 * RW_PITCH;                  // the halo columns of 16 input rows (two groups of eight)
constexpr int RW_WAVE_BYTES = RW_SLOTS * RW_ROW + RW_HALO + 16 * RW_EPS * 4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool LAZY>
__global__ __launch_bounds__(256, 3) void conv_bf16_rows32_kernel(ConvBParams p, int strips, int strip_groups, int segs) {
    __shared__ __attribute__((aligned(16))) char smem[4 * RW_WAVE_BYTES];
    __shared__ double red[4 * 32 * 2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    char* const ring = smem + wave * RW_WAVE_BYTES;
    char* const halo = ring + RW_SLOTS * RW_ROW;
    float* const ep = reinterpret_cast<float*>(halo + RW_HALO);
    int bid = blockIdx.x;
    const int sg = bid % strip_groups;
    bid /= strip_groups;
    const int seg = bid % segs;
    const int img = bid / segs;
    const int strip = sg * 4 + wave;
    const bool active = strip < strips;
    const int ox0 = strip * 16;
    const int oy_begin = seg * RW_RS, oy_end = oy_begin + RW_RS < p.hout ? oy_begin + RW_RS : p.hout;

    float sums[2][2] = {{0.f, 0.f}, {0.f, 0.f}};                  // a lane's <= 128 values per channel of the segment in fp32 (its outputs are rounded to bf16 anyway)
    if (active) {
        // weights: fragment (tap, channel tile) of chunk 0, plane 0 of packed[chunk][tap][ntile_total = 2][3][64][8]
        bf16x8 b[9][2];
        const bf16x8* wbase = reinterpret_cast<const bf16x8*>(p.wpack) + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int t = 0; t < 2; ++t) b[tap][t] = wbase[(size_t)(tap * 2 + t) * 3 * 64];
        float bias[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bias[t] = p.bias ? p.bias[t * 16 + pl] : 0.f;
        // staging items (pixel, channel octet c8 = lane & 3): the main pass takes pixels 0 ... 15 of one row (lane >> 2), the halo pass
        // pixels 16, 17 of eight rows (row lane >> 3, pixel (lane >> 2) & 1)
        const int c8 = lane & 3;
        const int px_a = lane >> 2;
        const int gx_a = ox0 - 1 + px_a, gx_h = ox0 + 15 + ((lane >> 2) & 1);
        const bool ok_a = gx_a >= 0 && gx_a < p.win, ok_h = gx_h < p.win;
        const int hrow = lane >> 3;
        f32x4 tb[4];                                               // (scale, shift) of channels 8 c8 ... 8 c8 + 7
        {
            const f32x4* t = LAZY ? reinterpret_cast<const f32x4*>(p.a.tab + (size_t)img * p.a.tab_c + p.a.tab_off + 8 * c8)
                                  : reinterpret_cast<const f32x4*>(p.dummy);
#pragma unroll
            for (int k = 0; k < 4; ++k) tb[k] = t[k];
        }
        float slope = *((LAZY && p.a.slope) ? p.a.slope : p.dummy);
        if (!(LAZY && p.a.slope)) slope = 0.f;
        const unsigned short* const src = reinterpret_cast<const unsigned short*>(p.a.ptr) + (size_t)img * p.hin * p.win * p.a.pitch + 8 * c8;
        const unsigned off_a = (unsigned)(ok_a ? gx_a : 0) * p.a.pitch, off_h = (unsigned)(ok_h ? gx_h : 0) * p.a.pitch;
        auto clamp_row = [&](int gy) { return gy < 0 ? 0 : (gy >= p.hin ? p.hin - 1 : gy); };     // rows outside the image are staged as zeros below
        auto load_row = [&](int gy, u32x4& ra) { ra = *reinterpret_cast<const u32x4*>(src + (size_t)clamp_row(gy) * p.win * p.a.pitch + off_a); };
        auto load_halo = [&](int gy0, u32x4& rh) { rh = *reinterpret_cast<const u32x4*>(src + (size_t)clamp_row(gy0 + hrow) * p.win * p.a.pitch + off_h); };
        // 0 <= slope <= 1 (ReLU, every PReLU near its 0.25 start): act(v) = max(v, slope v), one instruction less per value than the select form
        const bool slope01 = slope >= 0.f && slope <= 1.f;
        auto activate = [&](u32x4 r, bool ok) -> u32x4 {           // 8 bf16 -> norm + act -> 8 bf16
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float lo = __builtin_bit_cast(float, r[k] << 16), hi = __builtin_bit_cast(float, r[k] & 0xffff0000u);
                const f32x4 t = tb[k];                             // (scale, shift) of channels 2 k and 2 k + 1 of the octet
                if constexpr (LAZY) {
                    lo = __builtin_fmaf(lo, t.x, t.y);
                    hi = __builtin_fmaf(hi, t.z, t.w);
                    if (slope01) {
                        lo = __builtin_fmaxf(lo, lo * slope);
                        hi = __builtin_fmaxf(hi, hi * slope);
                    } else {
                        lo = prelu_h(lo, slope);
                        hi = prelu_h(hi, slope);
                    }
                }
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                const bf16x2 h{(__bf16)lo, (__bf16)hi};
                o[k] = ok ? __builtin_bit_cast(unsigned, h) : 0u;
            }
            return o;
        };
        const int r0 = oy_begin - 1;                               // first input row of the segment: halo slots count from it
        auto stage_halo = [&](int gy0, u32x4 rh) {                 // input rows gy0 ... gy0 + 7 -> halo slots (gy - r0) & 15
            const int gy = gy0 + hrow;
            *reinterpret_cast<u32x4*>(halo + ((((gy - r0) & 15) * 2 + ((lane >> 2) & 1)) * RW_PITCH) + c8 * 16) =
                activate(rh, gy >= 0 && gy < p.hin && ok_h);
        };
        auto stage = [&](int gy, u32x4 ra) {                       // input row gy -> ring slot (gy + 1) & 3
            char* const slot = ring + ((gy + 1) & 3) * RW_ROW;
            u32x4 hv;
            if (lane < 8) hv = *reinterpret_cast<const u32x4*>(halo + (((gy - r0) & 15) * 2 + (lane >> 2)) * RW_PITCH + c8 * 16);
            *reinterpret_cast<u32x4*>(slot + px_a * RW_PITCH + c8 * 16) = activate(ra, gy >= 0 && gy < p.hin && ok_a);
            if (lane < 8) *reinterpret_cast<u32x4*>(slot + (16 + (lane >> 2)) * RW_PITCH + c8 * 16) = hv;
        };

        constexpr int UNR = 3;                                     // rows per trip = registers of the row queue (it rotates without copies)
        u32x4 aq[UNR], hn;                                         // aq[j]: the input row that step j of a trip stages; reloaded UNR rows ahead
        {
            u32x4 h0, a0, a1, a2;
            load_halo(r0, h0);
            load_row(r0, a0);
            load_row(r0 + 1, a1);
            load_row(r0 + 2, a2);
            load_halo(r0 + 8, hn);
            stage_halo(r0, h0);
            __builtin_amdgcn_wave_barrier();
            stage(r0, a0);
            stage(r0 + 1, a1);
            stage(r0 + 2, a2);
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) load_row(oy_begin + 2 + j, aq[j]);
        const size_t out_row = ((size_t)img * p.hout) * p.wout + ox0;
        unsigned short* const out = reinterpret_cast<unsigned short*>(p.out);
        const int opix = lane >> 2;                                // the pixel (and channel octet c8) this lane stores
        const bool ostore = ox0 + opix < p.wout;
        const bool full = ox0 + 16 <= p.wout;
        // a row leaves one step late: its slab is read (other lanes' values: LDS operations of a wave complete in order) while the next
        // row's MFMAs run, so that nothing in a step waits for the matrix pipe but the slab writes at its end
        auto store_row = [&](int oy) {
            if (ostore) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(&ep[opix * RW_EPS + c8 * 8]);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(&ep[opix * RW_EPS + c8 * 8 + 4]);
                const bf16x8 h{(__bf16)v0.x, (__bf16)v0.y, (__bf16)v0.z, (__bf16)v0.w, (__bf16)v1.x, (__bf16)v1.y, (__bf16)v1.z, (__bf16)v1.w};
                *reinterpret_cast<bf16x8*>(out + (out_row + (size_t)oy * p.wout + opix) * p.pitch_out + c8 * 8) = h;
            }
        };
        // output row oy: window rows oy - 1 ... oy + 1 were staged by earlier steps; this step stages input row oy + 2 (areg) into the slot of
        // row oy - 2 and reloads areg UNR rows further - independent of its MFMAs, the scheduler interleaves the two
        auto row_step = [&](int oy, u32x4& areg) {
            __builtin_amdgcn_wave_barrier();
            if (oy > oy_begin) store_row(oy - 1);
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            // the A fragment of tap i + DEPTH is requested before the MFMAs of tap i (pinned: left alone the compiler keeps one in flight)
            constexpr int DEPTH = 2;
            auto a_ptr = [&](int i) { return ring + ((oy + i / 3) & 3) * RW_ROW + (pl + i % 3) * RW_PITCH + g * 16; };      // input row oy - 1 + ky
            bf16x8 af[DEPTH + 1];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a_ptr(i));
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                if (i + DEPTH < 9) af[(i + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(a_ptr(i + DEPTH));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i % (DEPTH + 1)], b[i][t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage(oy + 2, areg);
            load_row(oy + 2 + UNR, areg);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[t][r] + bias[t];
                    ep[(g * 4 + r) * RW_EPS + t * 16 + pl] = v;
                    const float vm = full ? v : (ox0 + g * 4 + r < p.wout ? v : 0.f);      // (wave-uniform: only the last strip of a row has columns beyond the image)
                    sums[t][0] += vm;
                    sums[t][1] = fmaf(vm, v, sums[t][1]);
                }
        };
#pragma unroll 1
        for (int oy = oy_begin; oy < oy_end; oy += UNR) {
#pragma unroll
            for (int j = 0; j < UNR; ++j) {
                if (oy + j >= oy_end) break;
                const int rel = oy + j - oy_begin;                 // (wave-uniform conditions: scalar unit)
                if (((rel + 3) & 7) == 0) {                        // input row oy + j + 2 = r0 + 8 k opens a group of eight: its halo columns, then the load of the group after it
                    stage_halo(oy + j + 2, hn);
                    load_halo(oy + j + 10, hn);
                }
                row_step(oy + j, aq[j]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        store_row(oy_end - 1);
    }
    if (p.stats) {
#pragma unroll
        for (int t = 0; t < 2; ++t) stats_stash_b((double)sums[t][0], (double)sums[t][1], red + ((wave * 2 + t) * 16 + pl) * 2, lane);
        __syncthreads();
        stats_flush_b<4, 32>(red, p.stats + (size_t)img * p.cout * 2, tid);
    }
}

// ------------------------------------------------------------------------------------------------
// The same streaming shape for 64 -> 64 channels (the 32-base net's 64 x 64 level, the widest tensors of the 64-base net of configs[4]).
// 9 taps x 2 K chunks x 4 channel tiles of weight fragments are 288 registers, so a strip belongs to a PAIR of waves (one 128-thread
// block): wave ct computes output channels 32 ct ... 32 ct + 31 (144 weight registers) and stages the 32-channel plane ct of every input
// row into the window the pair shares - half the staging arithmetic each, ONE block barrier per row, placed where the row staged in a step
// is first needed a step later (so neither wave waits long for the other).  Rows are two planes of 18 pixels x 80 bytes (conflict-free
// 16-byte reads at stride 1 as well).
// ------------------------------------------------------------------------------------------------
constexpr int R6_RS = 64;                                   // output rows of a pair's strip segment (64 x 64 images: one segment; 195.7 -> 185.7 us against 32)
constexpr int R6_PITCH = 80;
constexpr int R6_PLANE = 18 * R6_PITCH;                     // one 32-channel plane of a row
constexpr int R6_ROW = 2 * R6_PLANE;
constexpr int R6_HROW = 2 * 2 * R6_PITCH;                   // halo pixels of a row: 2 planes x 2 pixels
constexpr int R6_HALO = 16 * R6_HROW;
constexpr int R6_BLOCK_BYTES = RW_SLOTS * R6_ROW + R6_HALO + 2 * 16 * RW_EPS * 4;

template <bool LAZY>
__global__ __launch_bounds__(128, 2) void conv_bf16_rows64_kernel(ConvBParams p, int strips, int segs) {
    __shared__ __attribute__((aligned(16))) char smem[R6_BLOCK_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int ct = __builtin_amdgcn_readfirstlane(tid >> 6);    // this wave's half of the output channels = the plane it stages
    const int pl = lane & 15, g = lane >> 4;
    char* const ring = smem;
    char* const halo = ring + RW_SLOTS * R6_ROW;
    float* const ep = reinterpret_cast<float*>(halo + R6_HALO) + ct * 16 * RW_EPS;
    int bid = blockIdx.x;
    const int strip = bid % strips;
    bid /= strips;
    const int seg = bid % segs;
    const int img = bid / segs;
    const int ox0 = strip * 16;
    const int oy_begin = seg * R6_RS, oy_end = oy_begin + R6_RS < p.hout ? oy_begin + R6_RS : p.hout;

    // weights: fragment (K chunk, tap, channel tile 2 ct + t), plane 0 of packed[chunk][tap][ntile_total = 4][3][64][8]
    bf16x8 b[2][9][2];
    const bf16x8* wbase = reinterpret_cast<const bf16x8*>(p.wpack) + lane;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int t = 0; t < 2; ++t) b[kc][tap][t] = wbase[(size_t)((kc * 9 + tap) * 4 + 2 * ct + t) * 3 * 64];
    float bias[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) bias[t] = p.bias ? p.bias[(2 * ct + t) * 16 + pl] : 0.f;
    float slope = *((LAZY && p.a.slope) ? p.a.slope : p.dummy);
    if (!(LAZY && p.a.slope)) slope = 0.f;
    const bool slope01 = slope >= 0.f && slope <= 1.f;
    // staging items of plane ct (pixel, channel octet c8 = lane & 3): the main pass takes pixels 0 ... 15 of one row (lane >> 2), the halo
    // pass pixels 16, 17 of eight rows (row lane >> 3, pixel (lane >> 2) & 1)
    const int c8 = lane & 3;
    const int px_a = lane >> 2;
    const int gx_a = ox0 - 1 + px_a, gx_h = ox0 + 15 + ((lane >> 2) & 1);
    const bool ok_a = gx_a >= 0 && gx_a < p.win, ok_h = gx_h < p.win;
    const int hrow = lane >> 3;
    const unsigned short* const src = reinterpret_cast<const unsigned short*>(p.a.ptr) + (size_t)img * p.hin * p.win * p.a.pitch + 32 * ct + 8 * c8;
    const unsigned off_a = (unsigned)(ok_a ? gx_a : 0) * p.a.pitch, off_h = (unsigned)(ok_h ? gx_h : 0) * p.a.pitch;
    auto clamp_row = [&](int gy) { return gy < 0 ? 0 : (gy >= p.hin ? p.hin - 1 : gy); };
    auto load_row = [&](int gy, u32x4& ra) { ra = *reinterpret_cast<const u32x4*>(src + (size_t)clamp_row(gy) * p.win * p.a.pitch + off_a); };
    auto load_halo = [&](int gy0, u32x4& rh) { rh = *reinterpret_cast<const u32x4*>(src + (size_t)clamp_row(gy0 + hrow) * p.win * p.a.pitch + off_h); };
    f32x4 tb[4];                                                   // this lane's octet of plane ct: 4 x (scale, shift, scale, shift)
    {
        const f32x4* t = LAZY ? reinterpret_cast<const f32x4*>(p.a.tab + (size_t)img * p.a.tab_c + p.a.tab_off + 32 * ct + 8 * c8)
                              : reinterpret_cast<const f32x4*>(p.dummy);
#pragma unroll
        for (int k = 0; k < 4; ++k) tb[k] = t[k];
    }
    auto activate = [&](u32x4 r, bool ok) -> u32x4 {               // 8 bf16 -> norm + act -> 8 bf16
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float lo = __builtin_bit_cast(float, r[k] << 16), hi = __builtin_bit_cast(float, r[k] & 0xffff0000u);
            if constexpr (LAZY) {
                const f32x4 t = tb[k];
                lo = __builtin_fmaf(lo, t.x, t.y);
                hi = __builtin_fmaf(hi, t.z, t.w);
                if (slope01) {
                    lo = __builtin_fmaxf(lo, lo * slope);
                    hi = __builtin_fmaxf(hi, hi * slope);
                } else {
                    lo = prelu_h(lo, slope);
                    hi = prelu_h(hi, slope);
                }
            }
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            const bf16x2 h{(__bf16)lo, (__bf16)hi};
            o[k] = ok ? __builtin_bit_cast(unsigned, h) : 0u;
        }
        return o;
    };
    const int r0 = oy_begin - 1;
    char* const my_halo = halo + ct * 2 * R6_PITCH + c8 * 16;     // + slot * R6_HROW + pixel * R6_PITCH
    auto stage_halo = [&](int gy0, u32x4 rh) {                     // plane ct of input rows gy0 ... gy0 + 7 -> halo slots (gy - r0) & 15
        const int gy = gy0 + hrow;
        *reinterpret_cast<u32x4*>(my_halo + ((gy - r0) & 15) * R6_HROW + ((lane >> 2) & 1) * R6_PITCH) = activate(rh, gy >= 0 && gy < p.hin && ok_h);
    };
    auto stage = [&](int gy, u32x4 ra) {                           // plane ct of input row gy -> ring slot (gy + 1) & 3
        char* const slot = ring + ((gy + 1) & 3) * R6_ROW + ct * R6_PLANE + c8 * 16;
        u32x4 hv;
        if (lane < 8) hv = *reinterpret_cast<const u32x4*>(my_halo + ((gy - r0) & 15) * R6_HROW + (lane >> 2) * R6_PITCH);
        *reinterpret_cast<u32x4*>(slot + px_a * R6_PITCH) = activate(ra, gy >= 0 && gy < p.hin && ok_a);
        if (lane < 8) *reinterpret_cast<u32x4*>(slot + (16 + (lane >> 2)) * R6_PITCH) = hv;
    };

    constexpr int UNR = 3;
    u32x4 aq[UNR], hn;
    {
        u32x4 h0, a0, a1, a2;
        load_halo(r0, h0);
        load_row(r0, a0);
        load_row(r0 + 1, a1);
        load_row(r0 + 2, a2);
        load_halo(r0 + 8, hn);
        stage_halo(r0, h0);
        __builtin_amdgcn_wave_barrier();
        stage(r0, a0);
        stage(r0 + 1, a1);
        stage(r0 + 2, a2);
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) load_row(oy_begin + 2 + j, aq[j]);
    const size_t out_row = ((size_t)img * p.hout) * p.wout + ox0;
    unsigned short* const out = reinterpret_cast<unsigned short*>(p.out) + 32 * ct;
    const int opix = lane >> 2;
    const bool ostore = ox0 + opix < p.wout;
    const bool full = ox0 + 16 <= p.wout;
    float sums[2][2] = {{0.f, 0.f}, {0.f, 0.f}};                  // a lane's <= 256 values per channel of the segment in fp32 (its outputs are rounded to bf16 anyway)
    auto store_row = [&](int oy) {
        if (ostore) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&ep[opix * RW_EPS + c8 * 8]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&ep[opix * RW_EPS + c8 * 8 + 4]);
            const bf16x8 h{(__bf16)v0.x, (__bf16)v0.y, (__bf16)v0.z, (__bf16)v0.w, (__bf16)v1.x, (__bf16)v1.y, (__bf16)v1.z, (__bf16)v1.w};
            *reinterpret_cast<bf16x8*>(out + (out_row + (size_t)oy * p.wout + opix) * p.pitch_out + c8 * 8) = h;
        }
    };
    // output row oy reads window rows oy - 1 ... oy + 1 (both planes) and stages plane ct of input row oy + 2 into the slot row oy - 2 left
    auto row_step = [&](int oy, u32x4& areg) {
        if (oy > oy_begin) store_row(oy - 1);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        // 18 steps (K chunk, tap row, tap column) in conv_bf16s_kernel's order of the products; the A fragment of step i + DEPTH is requested
        // before the MFMAs of step i (pinned: left alone the compiler keeps one fragment in flight and every MFMA pair waits ~100 cycles for LDS)
        constexpr int DEPTH = 3;
        auto a_ptr = [&](int i) {
            const int kc = i / 9, ky = (i % 9) / 3, kx = i % 3;
            return ring + ((oy + ky) & 3) * R6_ROW + kc * R6_PLANE + (pl + kx) * R6_PITCH + g * 16;
        };
        bf16x8 af[DEPTH + 1];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a_ptr(i));      // rows oy - 1, oy: complete since the previous step's barrier
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            // the pair's ONE barrier per row sits in front of the first REQUEST for the newest row (oy + 1, staged during the previous step): the
            // slower wave of the pair has the first steps of the matrix phase to finish its share of the staging.  Past it both waves have left
            // step oy - 1 for good: the slot of row oy - 2, written below, is free.
            if (i + DEPTH == 6) __syncthreads();
            if (i + DEPTH < 18) af[(i + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(a_ptr(i + DEPTH));
            __builtin_amdgcn_sched_barrier(0);
            const int kc = i / 9, tap = i % 9;
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i % (DEPTH + 1)], b[kc][tap][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        stage(oy + 2, areg);
        load_row(oy + 2 + UNR, areg);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[t][r] + bias[t];
                ep[(g * 4 + r) * RW_EPS + t * 16 + pl] = v;
                const float vm = full ? v : (ox0 + g * 4 + r < p.wout ? v : 0.f);
                sums[t][0] += vm;
                sums[t][1] = fmaf(vm, v, sums[t][1]);
            }
    };
    __syncthreads();                                               // rows r0 ... r0 + 2 of both planes
#pragma unroll 1
    for (int oy = oy_begin; oy < oy_end; oy += UNR) {
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            if (oy + j >= oy_end) break;
            const int rel = oy + j - oy_begin;
            if (((rel + 3) & 7) == 0) {
                stage_halo(oy + j + 2, hn);
                load_halo(oy + j + 10, hn);
            }
            row_step(oy + j, aq[j]);
        }
    }
    store_row(oy_end - 1);
    if (p.stats) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            stats_commit_b((double)sums[t][0], (double)sums[t][1], p.stats + ((size_t)img * p.cout + (2 * ct + t) * 16 + pl) * 2, lane);
    }
}

// ------------------------------------------------------------------------------------------------
// The streaming shape for the fused stride-2 pair [unit0 || residual] of the first encoder level below the image (round 5): 3x3 stride 2, 32 input
// channels (KC = cin / 32 = 1; KC = 2 compiles, but at 256 registers and 94 KB of LDS it ran slower than the block kernel), output channels in groups of 128.  conv_bf16s_kernel<2, 3, 4, 2, 1> runs the 32 -> 128 layer as blocks of ONE K
// chunk - gather 9 x 33 pixels, barrier, 72 MFMAs per wave, epilogue: 21 % matrix busy, 410 us for a launch whose traffic is worth 200.
// Here the FOUR waves of a block walk the same strip of 16 output columns down RS2 rows, wave w computing output channels 32 w ... 32 w + 31
// of the block's 128 (18 KC weight fragments in registers for the whole strip); each keeps its own rolling window of four input rows (33
// pixels x KC planes) in wave-private LDS - the inputs of these layers are materialised sums, so staging is a plain copy and the three
// re-reads of a row by the other waves are L2 hits; two new input rows per output row, requested two steps ahead; NO block barrier at all
// (every wave owns its channels' statistics).  Same products in the same order as the block kernel (tap-major, KC k-steps per tap).
// ------------------------------------------------------------------------------------------------
constexpr int S2_RS = 32;                                    // output rows of a block's strip segment
constexpr int S2_PITCH = 80;                                 // bytes per staged pixel and plane: 32 bf16 + pad (stride-2 A reads)
constexpr int S2_NPX = 33;                                   // input columns 2 ox0 - 1 ... 2 ox0 + 31
template <int KC>
struct S2Cfg {
    static constexpr int PLANE = S2_NPX * S2_PITCH;
    static constexpr int ROW = KC * PLANE;
    static constexpr int WAVE_BYTES = 4 * ROW + 16 * RW_EPS * 4;
};

template <int KC>
__global__ __launch_bounds__(256, 2) void conv_bf16_rows_s2_kernel(ConvBParams p, int strips, int segs, int cgroups) {
    typedef S2Cfg<KC> C;
    __shared__ __attribute__((aligned(16))) char smem[4 * C::WAVE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    char* const ring = smem + wave * C::WAVE_BYTES;
    float* const ep = reinterpret_cast<float*>(ring + 4 * C::ROW);
    int bid = blockIdx.x;
    const int cg = bid % cgroups;
    bid /= cgroups;
    const int strip = bid % strips;
    bid /= strips;
    const int seg = bid % segs;
    const int img = bid / segs;
    const int ox0 = strip * 16;
    const int oy_begin = seg * S2_RS, oy_end = oy_begin + S2_RS < p.hout ? oy_begin + S2_RS : p.hout;
    const int nt0 = cg * 8 + wave * 2;                          // this wave's first 16-channel tile

    // weights: fragment (K chunk, tap, channel tile nt0 + t), plane 0 of packed[chunk][tap][ntile_total][3][64][8]
    bf16x8 b[KC][9][2];
    const bf16x8* wbase = reinterpret_cast<const bf16x8*>(p.wpack) + lane;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int t = 0; t < 2; ++t) b[kc][tap][t] = wbase[(size_t)((kc * 9 + tap) * p.ntile_total + nt0 + t) * 3 * 64];
    float bias[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) bias[t] = p.bias ? p.bias[(nt0 + t) * 16 + pl] : 0.f;

    // staging items (pixel, channel octet c8 = lane & 3) of a plane: passes 0 / 1 take pixels 0 ... 15 / 16 ... 31 (lane >> 2), pass 2 pixel 32 (4 lanes)
    const int c8 = lane & 3;
    const int gx0 = 2 * ox0 - 1;                                // image column of staged pixel 0
    int offx[3];
    bool okx[3];
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
        const int px = ps < 2 ? 16 * ps + (lane >> 2) : 32;
        const int gx = gx0 + px;
        okx[ps] = gx >= 0 && gx < p.win && (ps < 2 || lane < 4);
        offx[ps] = (okx[ps] ? gx : 0) * p.a.pitch;
    }
    const unsigned short* const src = reinterpret_cast<const unsigned short*>(p.a.ptr) + (size_t)img * p.hin * p.win * p.a.pitch + 8 * c8;
    auto clamp_row = [&](int gy) { return gy < 0 ? 0 : (gy >= p.hin ? p.hin - 1 : gy); };
    struct RowRegs { u32x4 v[KC][3]; };
    auto load_row = [&](int gy, RowRegs& r) {                   // unconditional, clamped loads
        const unsigned short* rowp = src + (size_t)clamp_row(gy) * p.win * p.a.pitch;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int ps = 0; ps < 3; ++ps) r.v[kc][ps] = *reinterpret_cast<const u32x4*>(rowp + offx[ps] + 32 * kc);
    };
    auto stage = [&](int gy, const RowRegs& r) {                // input row gy -> ring slot (gy + 1) & 3; rows / columns outside the image are zeros
        char* const slot = ring + ((gy + 1) & 3) * C::ROW;
        const bool oky = gy >= 0 && gy < p.hin;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps)
                *reinterpret_cast<u32x4*>(slot + kc * C::PLANE + (16 * ps + (lane >> 2)) * S2_PITCH + c8 * 16) =
                    (oky && okx[ps]) ? r.v[kc][ps] : u32x4{0u, 0u, 0u, 0u};
            if (lane < 4)
                *reinterpret_cast<u32x4*>(slot + kc * C::PLANE + 32 * S2_PITCH + c8 * 16) = (oky && okx[2]) ? r.v[kc][2] : u32x4{0u, 0u, 0u, 0u};
        }
    };

    float sums[2][2] = {{0.f, 0.f}, {0.f, 0.f}};                 // a lane's <= 128 values per channel of the segment in fp32 (the outputs are rounded to bf16 anyway)
    RowRegs q0, q1;                                             // the two input rows the NEXT step needs (2 oy + 2, 2 oy + 3)
    {
        RowRegs a0, a1, a2;
        load_row(2 * oy_begin - 1, a0);
        load_row(2 * oy_begin, a1);
        load_row(2 * oy_begin + 1, a2);
        load_row(2 * oy_begin + 2, q0);
        load_row(2 * oy_begin + 3, q1);
        stage(2 * oy_begin - 1, a0);
        stage(2 * oy_begin, a1);
        stage(2 * oy_begin + 1, a2);
    }
    const size_t out_row = ((size_t)img * p.hout) * p.wout + ox0;
    unsigned short* const out = reinterpret_cast<unsigned short*>(p.out) + nt0 * 16;
    const int opix = lane >> 2;                                 // the pixel (and channel octet c8) this lane stores
    const bool ostore = ox0 + opix < p.wout;
    const bool full = ox0 + 16 <= p.wout;
    // a row leaves through a wave-private slab as ONE 16-byte store per lane, one step late (its slab is read while the next row's MFMAs run).
    // (Measured against it: the MFMA operands swapped - weights as A, pixels as B - so that a lane holds four consecutive channels of one pixel
    // and stores 8 bytes straight from the accumulators, no slab: 1.75 vs 1.57 ms for the stride-2 group per slide - the 8-byte stores lose
    // more than the slab costs.)
    auto store_row = [&](int oy) {
        if (ostore) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&ep[opix * RW_EPS + c8 * 8]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&ep[opix * RW_EPS + c8 * 8 + 4]);
            const bf16x8 h{(__bf16)v0.x, (__bf16)v0.y, (__bf16)v0.z, (__bf16)v0.w, (__bf16)v1.x, (__bf16)v1.y, (__bf16)v1.z, (__bf16)v1.w};
            *reinterpret_cast<bf16x8*>(out + (out_row + (size_t)oy * p.wout + opix) * p.pitch_out + c8 * 8) = h;
        }
    };
#pragma unroll 1
    for (int oy = oy_begin; oy < oy_end; ++oy) {
        __builtin_amdgcn_wave_barrier();
        if (oy > oy_begin) store_row(oy - 1);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        // input row 2 oy - 1 + ky sits in slot (2 oy + ky) & 3; output pixel pl reads staged pixel 2 pl + kx
        constexpr int NA = 9 * KC, DEPTH = 2;
        auto a_ptr = [&](int i) {
            const int kc = i / 9, tap = i - kc * 9, ky = tap / 3, kx = tap - ky * 3;
            return ring + ((2 * oy + ky) & 3) * C::ROW + kc * C::PLANE + (2 * pl + kx) * S2_PITCH + g * 16;
        };
        bf16x8 af[DEPTH + 1];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a_ptr(i));
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i + DEPTH < NA) af[(i + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(a_ptr(i + DEPTH));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i % (DEPTH + 1)], b[i / 9][i % 9][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the two rows of the next step: into the slots rows 2 oy - 2 and 2 oy - 1 leave (LDS operations of a wave complete in order: the
        // reads of row 2 oy - 1 above are done before these writes)
        stage(2 * oy + 2, q0);
        stage(2 * oy + 3, q1);
        load_row(2 * oy + 4, q0);
        load_row(2 * oy + 5, q1);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[t][r] + bias[t];
                ep[(g * 4 + r) * RW_EPS + t * 16 + pl] = v;
                const float vm = full ? v : (ox0 + g * 4 + r < p.wout ? v : 0.f);
                sums[t][0] += vm;
                sums[t][1] = fmaf(vm, v, sums[t][1]);
            }
    }
    __builtin_amdgcn_wave_barrier();
    store_row(oy_end - 1);
    if (p.stats) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            stats_commit_b((double)sums[t][0], (double)sums[t][1], p.stats + ((size_t)img * p.cout + (nt0 + t) * 16 + pl) * 2, lane);
    }
}

// ------------------------------------------------------------------------------------------------
// The stride-2 pair with 64 input channels (64 -> 2 x 128 at the 32-base net's second level, 64 -> 2 x 128 of the 64-base net's first: the
// widest stride-2 launch of configs[4]).  Two K steps per tap are 36 weight fragments = 144 registers for a wave's 32 output channels, so the
// four waves of a block cannot each keep a private copy of the rows as conv_bf16_rows_s2_kernel<1> does (94 KB of LDS, measured slower than
// the block kernel): here they SHARE one window of six row slots (two 32-channel planes of 33 pixels each) that they stage together - two
// new input rows per output row are 528 sixteen-byte items for 256 lanes - with ONE block barrier per output row.  Six slots because the
// rows staged in a step (2 oy + 2, 2 oy + 3) must not land on a row another wave may still be reading (2 oy - 1 ... 2 oy + 1).
// ------------------------------------------------------------------------------------------------
constexpr int S2X_PLANE = S2_NPX * S2_PITCH;
constexpr int S2X_ROW = 2 * S2X_PLANE;
constexpr int S2X_SLOTS = 6;
constexpr int S2X_ITEMS = 2 * S2_NPX * 4;                    // 16-byte items of one input row: plane x pixel x channel octet
constexpr int S2X_BLOCK_BYTES = S2X_SLOTS * S2X_ROW + 4 * 16 * RW_EPS * 4;

__global__ __launch_bounds__(256, 2) void conv_bf16_rows_s2x_kernel(ConvBParams p, int strips, int segs, int cgroups) {
    __shared__ __attribute__((aligned(16))) char smem[S2X_BLOCK_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    char* const ring = smem;
    float* const ep = reinterpret_cast<float*>(smem + S2X_SLOTS * S2X_ROW) + wave * 16 * RW_EPS;
    int bid = blockIdx.x;
    const int cg = bid % cgroups;
    bid /= cgroups;
    const int strip = bid % strips;
    bid /= strips;
    const int seg = bid % segs;
    const int img = bid / segs;
    const int ox0 = strip * 16;
    const int oy_begin = seg * S2_RS, oy_end = oy_begin + S2_RS < p.hout ? oy_begin + S2_RS : p.hout;
    const int nt0 = cg * 8 + wave * 2;

    bf16x8 b[2][9][2];
    const bf16x8* wbase = reinterpret_cast<const bf16x8*>(p.wpack) + lane;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int t = 0; t < 2; ++t) b[kc][tap][t] = wbase[(size_t)((kc * 9 + tap) * p.ntile_total + nt0 + t) * 3 * 64];
    float bias[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) bias[t] = p.bias ? p.bias[(nt0 + t) * 16 + pl] : 0.f;

    // staging: the two input rows of a step are 2 x S2X_ITEMS = 528 items (row, plane, pixel, octet) dealt to the block's 256 lanes in three passes
    const int gx0 = 2 * ox0 - 1;
    int it_row[3], it_lds[3], it_goff[3];
    bool it_ok[3];
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
        const int it = tid + 256 * ps;
        const bool live = it < 2 * S2X_ITEMS;
        const int r = it / S2X_ITEMS, rem = it - r * S2X_ITEMS;
        const int plane = rem / (S2_NPX * 4), px = (rem - plane * S2_NPX * 4) >> 2, c8 = rem & 3;
        const int gx = gx0 + px;
        it_row[ps] = r;
        it_lds[ps] = plane * S2X_PLANE + px * S2_PITCH + c8 * 16;
        it_ok[ps] = live && gx >= 0 && gx < p.win;
        it_goff[ps] = (it_ok[ps] ? gx : 0) * p.a.pitch + 32 * plane + 8 * c8;
        if (!live) it_row[ps] = -1;
    }
    const unsigned short* const src = reinterpret_cast<const unsigned short*>(p.a.ptr) + (size_t)img * p.hin * p.win * p.a.pitch;
    auto clamp_row = [&](int gy) { return gy < 0 ? 0 : (gy >= p.hin ? p.hin - 1 : gy); };
    struct PairRegs { u32x4 v[3]; };
    auto load_pair = [&](int gy0, PairRegs& r) {                // input rows gy0, gy0 + 1 (unconditional, clamped loads)
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) {
            const int gy = gy0 + (it_row[ps] > 0 ? 1 : 0);
            r.v[ps] = *reinterpret_cast<const u32x4*>(src + (size_t)clamp_row(gy) * p.win * p.a.pitch + it_goff[ps]);
        }
    };
    auto slot_of = [&](int gy) { return ((gy + 7) % S2X_SLOTS) * S2X_ROW; };      // gy >= -1
    auto stage_pair = [&](int gy0, const PairRegs& r) {
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) {
            if (it_row[ps] < 0) continue;
            const int gy = gy0 + it_row[ps];
            const bool ok = it_ok[ps] && gy >= 0 && gy < p.hin;
            *reinterpret_cast<u32x4*>(ring + slot_of(gy) + it_lds[ps]) = ok ? r.v[ps] : u32x4{0u, 0u, 0u, 0u};
        }
    };

    float sums[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    PairRegs q;
    {
        PairRegs a0, a1;
        load_pair(2 * oy_begin - 2, a0);                         // rows 2 oy - 2 (unused, staged into a free slot) and 2 oy - 1
        load_pair(2 * oy_begin, a1);
        load_pair(2 * oy_begin + 2, q);
        stage_pair(2 * oy_begin - 2, a0);                        // (rows above the image are staged as zeros)
        stage_pair(2 * oy_begin, a1);
    }
    const size_t out_row = ((size_t)img * p.hout) * p.wout + ox0;
    unsigned short* const out = reinterpret_cast<unsigned short*>(p.out) + nt0 * 16;
    const int opix = lane >> 2, c8o = lane & 3;
    const bool ostore = ox0 + opix < p.wout;
    const bool full = ox0 + 16 <= p.wout;
    auto store_row = [&](int oy) {
        if (ostore) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&ep[opix * RW_EPS + c8o * 8]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&ep[opix * RW_EPS + c8o * 8 + 4]);
            const bf16x8 h{(__bf16)v0.x, (__bf16)v0.y, (__bf16)v0.z, (__bf16)v0.w, (__bf16)v1.x, (__bf16)v1.y, (__bf16)v1.z, (__bf16)v1.w};
            *reinterpret_cast<bf16x8*>(out + (out_row + (size_t)oy * p.wout + opix) * p.pitch_out + c8o * 8) = h;
        }
    };
#pragma unroll 1
    for (int oy = oy_begin; oy < oy_end; ++oy) {
        __syncthreads();                                         // rows 2 oy - 1 ... 2 oy + 1 are complete in the shared window
        if (oy > oy_begin) store_row(oy - 1);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        constexpr int NA = 18, DEPTH = 2;
        const int so[3] = {slot_of(2 * oy - 1), slot_of(2 * oy), slot_of(2 * oy + 1)};
        auto a_ptr = [&](int i) {
            const int kc = i / 9, tap = i - kc * 9, ky = tap / 3, kx = tap - ky * 3;
            return ring + so[ky] + kc * S2X_PLANE + (2 * pl + kx) * S2_PITCH + g * 16;
        };
        bf16x8 af[DEPTH + 1];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a_ptr(i));
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i + DEPTH < NA) af[(i + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(a_ptr(i + DEPTH));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i % (DEPTH + 1)], b[i / 9][i % 9][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        stage_pair(2 * oy + 2, q);                               // into the slots of rows 2 oy - 4, 2 oy - 3: nobody reads those any more
        load_pair(2 * oy + 4, q);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[t][r] + bias[t];
                ep[(g * 4 + r) * RW_EPS + t * 16 + pl] = v;
                const float vm = full ? v : (ox0 + g * 4 + r < p.wout ? v : 0.f);
                sums[t][0] += vm;
                sums[t][1] = fmaf(vm, v, sums[t][1]);
            }
    }
    __builtin_amdgcn_wave_barrier();
    store_row(oy_end - 1);
    if (p.stats) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            stats_commit_b((double)sums[t][0], (double)sums[t][1], p.stats + ((size_t)img * p.cout + (nt0 + t) * 16 + pl) * 2, lane);
    }
}

constexpr long long STREAM_MIN_BLOCKS = 1024;   // the round-5 streaming kernels (stride-2 pair, transposed conv) below this many blocks: the block kernels
                                                // (32 tiles of 256 x 256 give them 256 blocks for 256 CUs: forward 0.758 vs 0.735 ms)
// CS_BF16_ROWS: 0 = block kernels, 1 = the 32 -> 32 layers streamed by waves, 2 (default) = the 64 -> 64 layers and the 32-channel stride-2 pair too
static int g_bf16_rows = [] { const char* e = getenv("CS_BF16_ROWS"); return e ? atoi(e) : 2; }();

// conv_bf16_rows64_kernel launches two-wave blocks, one per (strip, 64-row segment, image): below ~3 blocks per CU the block kernel is
// faster (32 tiles of 256 x 256 = 128 blocks: forward 0.718 -> 0.640 ms, 8 tiles 0.531 -> 0.448; 240 tiles = 960 blocks: 2.938 vs 2.965 with
// the streaming kernel; profiles/rows64_threshold_r05.txt)
constexpr long long ROWS64_MIN_BLOCKS = 768;
int launch_conv_bf16s(hipStream_t s, int ksize, int stride, int planes, const ConvSrc& a, const ConvSrc& b,
                      const unsigned short* wpack, const float* bias, float* out, int pitch_out, int cout, double* stats, int n,
                      int hin, int win, bool act16, const float* in_amax, const ConvOutExtra* ex) {
    CS_REQUIRE(ksize == 3 || ksize == 1, "conv(bf16): ksize %d unsupported", ksize);
    CS_REQUIRE(!ex || (!ex->addend2 && !act16 && !(ex->out2 && ex->addend)), "conv(bf16): a plain fp32 addend or an (out, out2) channel split");
    CS_REQUIRE(!ex || !ex->out2 || (ex->split % 16 == 0 && ex->split > 0 && ex->split < cout), "conv(bf16): bad output split %d", ex ? ex->split : 0);
    CS_REQUIRE(!act16 || planes == 1, "conv(bf16): bf16 activations in HBM go with the plain-bf16 mode only");
    CS_REQUIRE(a.h16 == act16 && (b.c == 0 || b.h16 == act16), "conv(bf16): activation element type does not match the mode");
    CS_REQUIRE((stride == 1 || stride == 2) && !(ksize == 1 && stride != 1), "conv(bf16): stride %d unsupported", stride);
    CS_REQUIRE(planes >= 1 && planes <= 4, "conv(bf16): %d planes", planes);
    CS_REQUIRE(a.c % 16 == 0 && b.c % 16 == 0 && a.c > 0 && cout % 16 == 0, "conv(bf16): channels must be multiples of 16");
    CS_REQUIRE(!a.ptr2 && !b.ptr2, "conv(bf16): sum sources are only understood by the transposed convolution");
    ConvBParams p{};
    p.a = to_devb(a);
    p.b = to_devb(b);
    p.wpack = wpack; p.bias = bias; p.out = out; p.stats = stats; p.dummy = reinterpret_cast<const float*>(wpack);
    p.pitch_out = pitch_out; p.cout = cout; p.ntile_total = cout / 16;
    p.n = n; p.hin = hin; p.win = win;
    p.in_amax = in_amax;
    p.addend = ex ? ex->addend : nullptr;
    p.pitch_add = ex ? ex->pitch_add : 0;
    p.out2 = ex ? ex->out2 : nullptr;
    p.pitch_out2 = ex ? ex->pitch_out2 : 0;
    p.split_nt = ex && ex->out2 ? ex->split / 16 : 0;
    const int pad = ksize / 2;
    p.hout = (hin + 2 * pad - ksize) / stride + 1;
    p.wout = (win + 2 * pad - ksize) / stride + 1;
    const int wn = (cout % 64 == 0) ? 4 : (cout % 32 == 0) ? 2 : 1;
    // Tile shapes that cut the per-wave weight streaming from L2 (the limiter of these kernels): double-height tiles for stride 1
    // (two-plane modes), stride-2 layers as 128-channel blocks in which every wave owns all 4 rows, stride-1 layers with >= 128
    // channels as 128-channel blocks in which every wave owns all 8 rows.  (A persistent form with cross-tile prefetch was built
    // in round 2 and measured slower - 6.73 vs 6.05 ms per slide for the stride-1 group: the next tile's patch held in registers
    // across the epilogue costs 40-70 VGPRs and a block per CU - it lives in the history only.)
    const bool tall = stride == 1 && p.hout >= 16;
    const bool wide128 = cout % 128 == 0;
    const double flops = 2.0 * n * p.hout * p.wout * (double)cout * (a.c + b.c) * ksize * ksize;
    const double esz = act16 ? 2.0 : 4.0;
    const double bytes = esz * n * ((double)hin * win * (a.c + b.c) + (double)p.hout * p.wout * cout);
    ProfScope prof(ksize == 1 ? K_CONV1 : (stride == 1 ? K_CONV3_S1 : K_CONV3_S2), s, flops, bytes);
    // the fused stride-2 pairs of the shallow encoder levels (raw 32- / 64-channel inputs, output channels in groups of 128): four waves stream a strip
    if (g_bf16_rows >= 2 && act16 && planes == 1 && ksize == 3 && stride == 2 && !a.tab && b.c == 0 && a.c == 32 && cout % 128 == 0 &&
        !ex && !in_amax && a.pitch % 8 == 0 && pitch_out % 8 == 0 && p.hout >= 8 && (size_t)win * a.pitch < ((size_t)1 << 31) &&
        (long long)(cout / 128) * ceil_div(p.wout, 16) * ceil_div(p.hout, S2_RS) * n >= STREAM_MIN_BLOCKS) {      // (small launches: too few strips to fill the chip)
        const int strips = ceil_div(p.wout, 16), segs = ceil_div(p.hout, S2_RS), cgroups = cout / 128;
        const dim3 grid((unsigned)(cgroups * strips * segs * n));
        hipLaunchKernelGGL(conv_bf16_rows_s2_kernel<1>, grid, dim3(256), 0, s, p, strips, segs, cgroups);      // (KC = 2, 64 input channels: 256 registers, one
                                                                                                              //  block per CU - measured slower than the block kernel)
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
    // ... and the one with 64 input channels: four waves, a shared window (conv_bf16_rows_s2x_kernel)
    if (g_bf16_rows >= 2 && act16 && planes == 1 && ksize == 3 && stride == 2 && !a.tab && b.c == 0 && a.c == 64 && cout % 128 == 0 &&
        !ex && !in_amax && a.pitch % 8 == 0 && pitch_out % 8 == 0 && p.hout >= 8 && (size_t)win * a.pitch < ((size_t)1 << 31) &&
        (long long)(cout / 128) * ceil_div(p.wout, 16) * ceil_div(p.hout, S2_RS) * n >= STREAM_MIN_BLOCKS) {
        const int strips = ceil_div(p.wout, 16), segs = ceil_div(p.hout, S2_RS), cgroups = cout / 128;
        hipLaunchKernelGGL(conv_bf16_rows_s2x_kernel, dim3((unsigned)(cgroups * strips * segs * n)), dim3(256), 0, s, p, strips, segs, cgroups);
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
    // the 32 -> 32 stride-1 layers of the bf16 datapath: a wave streams a 16-column strip (conv_bf16_rows32_kernel)
    if (g_bf16_rows >= 2 && act16 && planes == 1 && ksize == 3 && stride == 1 && a.c == 64 && b.c == 0 && cout == 64 && !ex && !in_amax &&
        a.pitch % 8 == 0 && pitch_out % 8 == 0 && hin >= 8 && (size_t)win * a.pitch < ((size_t)1 << 31) && (!a.tab || (a.tab_c ? a.tab_c : a.c) >= 64) &&
        (long long)ceil_div(p.wout, 16) * ceil_div(p.hout, R6_RS) * n >= ROWS64_MIN_BLOCKS) {
        const int strips = ceil_div(p.wout, 16), segs = ceil_div(p.hout, R6_RS);
        if (a.tab) hipLaunchKernelGGL(conv_bf16_rows64_kernel<true>, dim3((unsigned)(strips * segs * n)), dim3(128), 0, s, p, strips, segs);
        else hipLaunchKernelGGL(conv_bf16_rows64_kernel<false>, dim3((unsigned)(strips * segs * n)), dim3(128), 0, s, p, strips, segs);
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
    if (g_bf16_rows && act16 && planes == 1 && ksize == 3 && stride == 1 && a.c == 32 && b.c == 0 && cout == 32 && !ex && !in_amax &&
        a.pitch % 8 == 0 && pitch_out % 8 == 0 && hin >= 8 && (size_t)win * a.pitch < ((size_t)1 << 31)) {
        const int strips = ceil_div(p.wout, 16), strip_groups = ceil_div(strips, 4), segs = ceil_div(p.hout, RW_RS);
        if (a.tab) hipLaunchKernelGGL(conv_bf16_rows32_kernel<true>, dim3((unsigned)(strip_groups * segs * n)), dim3(256), 0, s, p, strips, strip_groups, segs);
        else hipLaunchKernelGGL(conv_bf16_rows32_kernel<false>, dim3((unsigned)(strip_groups * segs * n)), dim3(256), 0, s, p, strips, strip_groups, segs);
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
#define CS_B3(S_, KS_, NP_, A16_)                                                   \
    switch (wn) {                                                                   \
        case 4: if (S_ == 1 && KS_ == 3 && NP_ != 3 && cout % 128 == 0) return launch_b<S_, KS_, 4, 2, (NP_ == 3 ? 4 : NP_), 1, A16_>(s, p); \
                if (tall && (NP_ == 4 || NP_ <= 2) && KS_ == 3) return launch_b<S_, KS_, 2, 2, (NP_ == 3 ? 4 : NP_), 2, A16_>(s, p); \
                if (wide128 && S_ == 2) return launch_b<S_, KS_, 4, 2, NP_, 1, A16_>(s, p);    \
                return launch_b<S_, KS_, 2, 2, NP_, 1, A16_>(s, p);                 \
        case 2: if (tall && KS_ == 3 && (NP_ == 1 || NP_ == 4)) return launch_b<S_, KS_, 1, 2, NP_, 2, A16_>(s, p); /* 32-channel layers: each A fragment feeds both tiles */ \
                return launch_b<S_, KS_, 2, 1, NP_, 1, A16_>(s, p);                 \
        default: return launch_b<S_, KS_, 1, 1, NP_, 1, A16_>(s, p);                \
    }
#define CS_B2(S_, KS_)                                                              \
    if (act16) { CS_B3(S_, KS_, 1, true) }                                          \
    switch (planes) {                                                               \
        case 4: CS_B3(S_, KS_, 4, false)                                            \
        case 3: CS_B3(S_, KS_, 3, false)                                            \
        case 2: CS_B3(S_, KS_, 2, false)                                            \
        default: CS_B3(S_, KS_, 1, false)                                           \
    }
    if (ksize == 1) { CS_B2(1, 1) }
    if (stride == 1) { CS_B2(1, 3) }
    CS_B2(2, 3)
#undef CS_B2
#undef CS_B3
}

// ------------------------------------------------------------------------------------------------
// transposed conv k3 s2 p1 op1, split-bf16: same operand handling as above, 4 output phases per input pixel
//   out[2i+a][2j+b] += in[i+di][j+dj] * W[ky][kx],  ky: 0->(a=1,di=1) 1->(a=0,di=0) 2->(a=1,di=0); same for kx.
// ------------------------------------------------------------------------------------------------
// SUM2 (bf16 datapath): the `b` source is the decoder's residual sum act(norm(V)) + act(norm(U)), read as its two bf16 tensors.
// The loads stay unconditional (chunks of `a` fetch their own item twice, an L1 hit) so that the waits keep their counts.
// THX = 2 (round 6): a block owns EIGHT input rows instead of four.  A wave reads every weight fragment from L2 itself (never through LDS),
// and at four rows a fragment feeds only MTW = 2 MFMAs per phase tap: the wide-K levels (128 + 256 -> 64 channels of configs[4]: 432
// fragments = 442 KB per block for 64 input pixels) were bound by that stream, 17 % matrix-busy at 3.4 x the layer's HBM floor.  Eight
// rows halve the weight bytes per pixel; the epilogue slab stays the four-row one and is used twice (rows 0-3, then 4-7).
template <int WN, int NTW, int NP, bool A16, bool SUM2 = false, int THX = 1>
__global__ __launch_bounds__(256, 2) void convt_bf16s_kernel(ConvBParams p) {
    static_assert(!A16 || NP == 1, "bf16 activations in HBM only make sense for the plain-bf16 mode");
    static_assert(!SUM2 || A16, "sum sources are implemented for the bf16 datapath");
    static_assert(THX == 1 || (THX == 2 && A16), "the eight-row tile exists for the bf16 datapath");
    typedef typename std::conditional<A16, u32x2, f32x4>::type raw4;
    constexpr bool F16 = SplitCfg<NP>::f16;
    constexpr int NPL = SplitCfg<NP>::planes, NTERM = SplitCfg<NP>::terms, NACC = 1;
    constexpr int WM = 4 / WN, TH = 4 * THX, TW = 16, MTW = TH / WM, NTB = WN * NTW;
    constexpr int IH = TH + 1, IW = TW + 1, PITCH = 96;
    constexpr int IN_ITEMS = IH * IW * 8, IN_ITERS = (IN_ITEMS + 255) / 256;
    constexpr int PLANE_BYTES = IH * IW * PITCH;
    constexpr int IN_BYTES = (NPL * PLANE_BYTES + 15) / 16 * 16;
    constexpr int EPS = 16 * NTB + 4;
    constexpr int EPH = 4;                                   // input rows per epilogue pass
    constexpr int EP_BYTES = 2 * EPH * 32 * EPS * 4;         // 8 output rows x 32 output pixels
    constexpr int SM_BYTES = IN_BYTES > EP_BYTES ? IN_BYTES : EP_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SM_BYTES];
    __shared__ double red[WM > 1 ? WM * NTB * 32 : 1];      // per-wave statistics of the block's channels (stats_stash_b)
    char* const in_s = smem;

    // fp16 mode: an out-of-range operand saturates at +-65504 instead of becoming inf (MODE.FP16_OVFL, bit 23 of HW_REG_MODE):
    // an activation beyond the documented domain then costs accuracy in its window, not a NaN from the next InstanceNorm
    if constexpr (F16) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, g = lane >> 4;
    const int wn = wave % WN, wm = wave / WN;
    const WorkId wk = decode_work(p.tiles_x * p.tiles_y * p.n, p.ntile_total / NTB);
    if (!wk.valid) return;
    int bid = wk.spatial;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int iy0 = ty * TH, ix0 = tx * TW;
    const int nt_w = wk.ct * NTB + wn * NTW;
    const int q = tid & 7;

    f32x4 acc[NACC][MTW][4][NTW];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[k][m][ph][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int goff[IN_ITERS];
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (tid >> 3) + 32 * it;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        goff[it] = (pix < IH * IW && gy < p.hin && gx < p.win) ? gy * p.win + gx : -1;
    }
    const int chunks_a = (p.a.c + 31) >> 5, chunks = chunks_a + ((p.b.c + 31) >> 5);
    raw4 rin[IN_ITERS];
    u32x2 rin2[SUM2 ? IN_ITERS : 1];
    f32x4 t0, t1, t0b, t1b;
    float slope = 0.f, slope_b = 0.f;
    bool lazy = false, qvalid = false, two = false;
    auto load_input = [&](int ch) {
        const bool from_a = ch < chunks_a;
        const SrcDevB& src = from_a ? p.a : p.b;
        const int c0 = (from_a ? ch : ch - chunks_a) << 5;
        qvalid = c0 + 4 * q < src.c;
        const int cq = qvalid ? c0 + 4 * q : 0;
        lazy = src.tab != nullptr;
        const f32x4* t = lazy ? reinterpret_cast<const f32x4*>(src.tab + (size_t)img * src.tab_c + src.tab_off + cq)
                              : reinterpret_cast<const f32x4*>(p.dummy);
        t0 = t[0];
        t1 = t[1];
        const float* sp = (lazy && src.slope) ? src.slope : p.dummy;
        slope = *sp;
        if (!(lazy && src.slope)) slope = 0.f;
        const size_t base_el = (size_t)img * p.hin * p.win * src.pitch + cq;
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int o = goff[it] >= 0 ? goff[it] : 0;
            if constexpr (A16) rin[it] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(src.ptr) + base_el + (size_t)o * src.pitch);
            else rin[it] = *reinterpret_cast<const f32x4*>(src.ptr + base_el + (size_t)o * src.pitch);
        }
        if constexpr (SUM2) {
            two = !from_a;
            const float2* tab2 = two ? p.b.tab2 : src.tab;                  // (not `two`: the same table / item again, never used)
            const f32x4* tb = tab2 ? reinterpret_cast<const f32x4*>(tab2 + (size_t)img * src.tab_c + src.tab_off + cq)
                                   : reinterpret_cast<const f32x4*>(p.dummy);
            t0b = tb[0];
            t1b = tb[1];
            const float* sp2 = (two && p.b.slope2) ? p.b.slope2 : p.dummy;
            slope_b = *sp2;
            if (!(two && p.b.slope2)) slope_b = 0.f;
            const float* ptr2 = two ? p.b.ptr2 : src.ptr;
#pragma unroll
            for (int it = 0; it < IN_ITERS; ++it) {
                const int o = goff[it] >= 0 ? goff[it] : 0;
                rin2[it] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(ptr2) + base_el + (size_t)o * src.pitch);
            }
        }
    };
    auto store_input = [&]() {
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int pix = (tid >> 3) + 32 * it;
            if (IN_ITEMS % 256 == 0 || pix < IH * IW) {
                f32x4 v;
                if constexpr (A16) v = bf16x4_to_f32(rin[it]);
                else v = rin[it];
                if (lazy) {
                    v.x = prelu_h(v.x * t0.x + t0.y, slope);
                    v.y = prelu_h(v.y * t0.z + t0.w, slope);
                    v.z = prelu_h(v.z * t1.x + t1.y, slope);
                    v.w = prelu_h(v.w * t1.z + t1.w, slope);
                }
                if constexpr (SUM2) {
                    if (two) {
                        const f32x4 u = bf16x4_to_f32(rin2[it]);
                        v.x += prelu_h(u.x * t0b.x + t0b.y, slope_b);
                        v.y += prelu_h(u.y * t0b.z + t0b.w, slope_b);
                        v.z += prelu_h(u.z * t1b.x + t1b.y, slope_b);
                        v.w += prelu_h(u.w * t1b.z + t1b.w, slope_b);
                    }
                }
                if (goff[it] < 0 || !qvalid) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pn = 0; pn < NPL; ++pn)
                    *reinterpret_cast<bf16x4*>(in_s + pn * PLANE_BYTES + pix * PITCH + q * 8) = split_plane<F16>(v, pn);
            }
        }
    };
    bf16x8 bq[3][NTW][NPL];
    const bf16x8* wbase = reinterpret_cast<const bf16x8*>(p.wpack) + (size_t)nt_w * 3 * 64 + lane;
    auto load_b = [&](int ch, int tap, int slot) {
        const bf16x8* w = wbase + ((size_t)ch * 9 + tap) * p.ntile_total * 3 * 64;
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int pn = 0; pn < NPL; ++pn) bq[slot][t][pn] = w[(t * 3 + pn) * 64];
    };

    load_input(0);
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) load_b(0, tap, tap);
    store_input();
    __syncthreads();
    const int last_tap = chunks * 9 - 1;
    for (int ch = 0; ch < chunks; ++ch) {
        const bool more = ch + 1 < chunks;
        load_input(more ? ch + 1 : ch);                     // unconditional loads, pinned (see conv_bf16s_kernel)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int pa = (ky == 1) ? 0 : 1, di = (ky == 0) ? 1 : 0;
            const int pb = (kx == 1) ? 0 : 1, dj = (kx == 0) ? 1 : 0;
            const int ph = pa * 2 + pb;
            bf16x8 b[NTW][NPL];
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int pn = 0; pn < NPL; ++pn) b[t][pn] = bq[tap % 3][t][pn];
            {
                const int nxt = ch * 9 + tap + 3 < last_tap ? ch * 9 + tap + 3 : last_tap;     // clamped: unconditional refill
                load_b(nxt / 9, nxt % 9, tap % 3);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int r = wm * MTW + m;
                bf16x8 a[NPL];
#pragma unroll
                for (int pn = 0; pn < NPL; ++pn)
                    a[pn] = *reinterpret_cast<const bf16x8*>(in_s + pn * PLANE_BYTES + ((r + di) * IW + pl + dj) * PITCH + g * 16);
#pragma unroll
                for (int term = NTERM - 1; term >= 0; --term) {
                    const int ia = term_a(term), ib = term_b(term);
                    const int ka = 0;
#pragma unroll
                    for (int t = 0; t < NTW; ++t) acc[ka][m][ph][t] = mfma_split<F16>(a[ia], b[t][ib], acc[ka][m][ph][t]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) asm volatile("" ::"v"(rin[it]));
        __syncthreads();
        if (more) {
            store_input();
            __syncthreads();
        }
    }

    float* ep = reinterpret_cast<float*>(smem);
    double s_[NTW], qq_[NTW];
    float sf_[NTW], qf_[NTW];                                // bf16 datapath: fp32 partial sums per lane (see stats_commit_b)
    float bias_[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        s_[t] = 0.0; qq_[t] = 0.0; sf_[t] = 0.f; qf_[t] = 0.f;
        bias_[t] = p.bias ? p.bias[(nt_w + t) * 16 + pl] : 0.f;
    }
    // the slab holds EPH = 4 input rows (8 output rows): one pass for the four-row tile, two for the eight-row one
#pragma unroll
    for (int pass = 0; pass < THX; ++pass) {
        if (pass > 0) __syncthreads();                       // the stores of the previous pass have read the slab
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int row = wm * MTW + m, iy = iy0 + row;
                if (THX > 1 && row / EPH != pass) continue;
                const int lrow = row - pass * EPH;
#pragma unroll
                for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ix = ix0 + g * 4 + r;
                        float v = acc[0][m][ph][t][r];
                        if constexpr (F16) v *= F16_OUT_SCALE;
                        v += bias_[t];
                        ep[((2 * lrow + (ph >> 1)) * 32 + 2 * (g * 4 + r) + (ph & 1)) * EPS + (wn * NTW + t) * 16 + pl] = v;
                        if (p.stats && iy < p.hin && ix < p.win) {
                            if constexpr (A16) {
                                sf_[t] += v;
                                qf_[t] = fmaf(v, v, qf_[t]);
                            } else {
                                s_[t] += (double)v;
                                qq_[t] += (double)v * (double)v;
                            }
                        }
                    }
            }
        }
        __syncthreads();
        {
            constexpr int F4 = NTB * 4;
            const int oy_base = 2 * (iy0 + pass * EPH);
            const size_t el0 = (((size_t)img * p.hout + oy_base) * p.wout + 2 * ix0) * p.pitch_out + wk.ct * NTB * 16;
#pragma unroll
            for (int it_ = 0; it_ < (2 * EPH * 32 * F4 + 255) / 256; ++it_) {      // (trip count visible to the unroller: the LDS reads of all passes go out together)
                const int i = tid + 256 * it_;
                if ((2 * EPH * 32 * F4) % 256 != 0 && i >= 2 * EPH * 32 * F4) break;
                const int pix = i / F4, c4 = i - pix * F4;
                const int row = pix >> 5, x = pix & 31;
                if (oy_base + row < p.hout && 2 * ix0 + x < p.wout) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&ep[pix * EPS + c4 * 4]);
                    const size_t el = el0 + ((int64_t)row * p.wout + x) * p.pitch_out + c4 * 4;
                    if constexpr (A16) *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(p.out) + el) = f32x4_to_bf16v(v);
                    else {
                        if (p.addend)
                            v += *reinterpret_cast<const f32x4*>(p.addend + (((size_t)img * p.hout + oy_base + row) * p.wout + 2 * ix0 + x) * p.pitch_add +
                                                                 wk.ct * NTB * 16 + c4 * 4);
                        *reinterpret_cast<f32x4*>(p.out + el) = v;
                    }
                }
            }
        }
    }
    if (p.stats) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int co = (nt_w + t) * 16 + pl;
            double s = s_[t], qq = qq_[t];
            if constexpr (A16) {
                s = (double)sf_[t];
                qq = (double)qf_[t];
            }
            if constexpr (WM > 1) stats_stash_b(s, qq, red + ((wm * NTB + wn * NTW + t) * 16 + pl) * 2, lane);
            else stats_commit_b(s, qq, p.stats + ((size_t)img * p.cout + co) * 2, lane);
        }
    }
    if constexpr (WM > 1) {
        __syncthreads();
        if (p.stats) stats_flush_b<WM, NTB * 16>(red, p.stats + ((size_t)img * p.cout + wk.ct * NTB * 16) * 2, tid);
    }
}

// ------------------------------------------------------------------------------------------------
// The streaming shape for the decoder's widest transposed conv (round 5): 64 skip + 64 deeper channels -> 32 channels, 64 x 64 -> 128 x 128
// per window (up1 of the 32-base net), bf16 datapath.  convt_bf16s_kernel<2, 1, 1> runs it as 30 720 blocks of 4 x 16 input pixels with four
// chunk barriers each: 556 us per 481-window launch for 1.26 GB of traffic (250 us at the HBM rate), matrix pipe 11 % busy - latency, not work.
// Here a PAIR of waves (one 128-thread block) walks a strip of 16 input columns down CT_RS input rows; wave ct computes output channels
// 16 ct ... 16 ct + 15 of all four output phases (36 weight fragments = 144 registers for the whole strip) and stages planes ct and ct + 2
// of every input row - one 32-channel plane of the skip tensor (a plain copy) and one of the deeper source, which is either a materialised
// tensor or (SUM2) the residual sum act(norm(V)) + act(norm(U)) computed in fp32 and rounded to bf16 once, exactly what norm_act_add16x8
// stores - into the window the pair shares; ONE block barrier per input row.  Per row and K step the nine taps read only FOUR distinct A
// fragments (input pixel offsets (0,0), (0,1), (1,0), (1,1)).  The two output rows of an input row leave through a wave-private slab as one
// 16-byte store per lane and row.  Same products in the same order as the block kernel (chunk-major, tap-major).
// ------------------------------------------------------------------------------------------------
constexpr int CT_RS = 32;                                    // input rows of a pair's strip segment
constexpr int CT_PITCH = 80;                                 // bytes per staged pixel and plane
constexpr int CT_NPX = 17;                                   // input columns ix0 ... ix0 + 16 (one halo column on the right)
constexpr int CT_PLANE = CT_NPX * CT_PITCH;
constexpr int CT_ROW = 4 * CT_PLANE;                         // planes: skip 0..31, skip 32..63, deeper 0..31, deeper 32..63
constexpr int CT_EPS = 20;                                   // floats per slab pixel: 16 channels + pad
constexpr int CT_SLAB = 2 * 32 * CT_EPS * 4;                 // two output rows of 32 pixels
constexpr int CT_BLOCK_BYTES = 4 * CT_ROW + 2 * CT_SLAB;

template <bool SUM2>
__global__ __launch_bounds__(128, 2) void convt_bf16_rows_kernel(ConvBParams p, int strips, int segs) {
    __shared__ __attribute__((aligned(16))) char smem[CT_BLOCK_BYTES];
    __shared__ f32x4 tab_s[2][2][16];                           // SUM2: [V | U][wave][octet c8 x 4] (scale, shift) pairs - 32 registers per lane otherwise
    const int tid = threadIdx.x, lane = tid & 63;
    const int ct = __builtin_amdgcn_readfirstlane(tid >> 6);    // this wave's channel tile = the pair of planes it stages
    const int pl = lane & 15, g = lane >> 4;
    char* const ring = smem;
    float* const ep = reinterpret_cast<float*>(smem + 4 * CT_ROW + ct * CT_SLAB);
    int bid = blockIdx.x;
    const int strip = bid % strips;
    bid /= strips;
    const int seg = bid % segs;
    const int img = bid / segs;
    const int ix0 = strip * 16;
    const int iy_begin = seg * CT_RS, iy_end = iy_begin + CT_RS < p.hin ? iy_begin + CT_RS : p.hin;

    // weights: fragment (K chunk kc, tap, channel tile ct), plane 0 of packed[chunk][tap][ntile_total = 2][3][64][8]; chunks 0, 1 = skip, 2, 3 = deeper
    bf16x8 b[4][9];
    const bf16x8* wbase = reinterpret_cast<const bf16x8*>(p.wpack) + lane;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) b[kc][tap] = wbase[(size_t)((kc * 9 + tap) * 2 + ct) * 3 * 64];
    const float bias = p.bias ? p.bias[ct * 16 + pl] : 0.f;

    // staging items (pixel, channel octet c8 = lane & 3) of a plane: the main pass takes pixels 0 ... 15 (lane >> 2), the halo pass pixel 16 (4 lanes)
    const int c8 = lane & 3;
    int offx[2];
    bool okx[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int gx = ix0 + (ps == 0 ? (lane >> 2) : 16);
        okx[ps] = gx < p.win && (ps == 0 || lane < 4);
        offx[ps] = okx[ps] ? gx : 0;
    }
    const size_t img_px = (size_t)img * p.hin * p.win;
    const unsigned short* const src_a = reinterpret_cast<const unsigned short*>(p.a.ptr) + img_px * p.a.pitch + 32 * ct + 8 * c8;
    const unsigned short* const src_b = reinterpret_cast<const unsigned short*>(p.b.ptr) + img_px * p.b.pitch + 32 * ct + 8 * c8;
    const unsigned short* const src_b2 = SUM2 ? reinterpret_cast<const unsigned short*>(p.b.ptr2) + img_px * p.b.pitch + 32 * ct + 8 * c8 : src_b;
    float slope_v = 0.f, slope_u = 0.f;
    if constexpr (SUM2) {                                        // the tables of this wave's 32 channels: 16 x 16 bytes per tensor, read back per item
        if (lane < 32) {
            const float2* t = (lane < 16 ? p.b.tab : p.b.tab2) + (size_t)img * p.b.tab_c + p.b.tab_off + 32 * ct;
            tab_s[lane >> 4][ct][lane & 15] = reinterpret_cast<const f32x4*>(t)[lane & 15];
        }
        slope_v = p.b.slope ? *p.b.slope : 0.f;
        slope_u = p.b.slope2 ? *p.b.slope2 : 0.f;
    }
    auto clamp_row = [&](int gy) { return gy >= p.hin ? p.hin - 1 : gy; };
    struct RowRegs { u32x4 a[2], v[2], u[2]; };
    auto load_row = [&](int gy, RowRegs& r) {                   // unconditional, clamped loads
        const size_t row = (size_t)clamp_row(gy) * p.win;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            r.a[ps] = *reinterpret_cast<const u32x4*>(src_a + (row + offx[ps]) * p.a.pitch);
            r.v[ps] = *reinterpret_cast<const u32x4*>(src_b + (row + offx[ps]) * p.b.pitch);
            if constexpr (SUM2) r.u[ps] = *reinterpret_cast<const u32x4*>(src_b2 + (row + offx[ps]) * p.b.pitch);
        }
    };
    // act(norm(V)) + act(norm(U)) on 8 channels: the operations and their order as in norm_act_add16x8_kernel (load_lazy8), one bf16 rounding
    auto sum8 = [&](const u32x4& v, const u32x4& u) {
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 tvk = tab_s[0][ct][4 * c8 + k], tuk = tab_s[1][ct][4 * c8 + k];
            const float v0 = __builtin_bit_cast(float, v[k] << 16), v1 = __builtin_bit_cast(float, v[k] & 0xffff0000u);
            const float u0 = __builtin_bit_cast(float, u[k] << 16), u1 = __builtin_bit_cast(float, u[k] & 0xffff0000u);
            const float s0 = prelu_h(v0 * tvk.x + tvk.y, slope_v) + prelu_h(u0 * tuk.x + tuk.y, slope_u);
            const float s1 = prelu_h(v1 * tvk.z + tvk.w, slope_v) + prelu_h(u1 * tuk.z + tuk.w, slope_u);
            o[k] = f32_to_bf16_bits(s0) | (f32_to_bf16_bits(s1) << 16);
        }
        return o;
    };
    auto stage = [&](int gy, const RowRegs& r) {                // input row gy -> ring slot gy & 3, planes ct and ct + 2; outside the image: zeros
        char* const slot = ring + (gy & 3) * CT_ROW;
        const bool oky = gy < p.hin;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            if (ps == 1 && lane >= 4) break;
            const int px = ps == 0 ? (lane >> 2) : 16;
            const bool ok = oky && okx[ps];
            const u32x4 zero{0u, 0u, 0u, 0u};
            u32x4 d = r.v[ps];
            if constexpr (SUM2) d = sum8(r.v[ps], r.u[ps]);
            *reinterpret_cast<u32x4*>(slot + ct * CT_PLANE + px * CT_PITCH + c8 * 16) = ok ? r.a[ps] : zero;
            *reinterpret_cast<u32x4*>(slot + (2 + ct) * CT_PLANE + px * CT_PITCH + c8 * 16) = ok ? d : zero;
        }
    };

    float sum = 0.f, sq = 0.f;                                   // this lane's channel: <= 32 rows x 16 values in fp32 (the outputs are rounded to bf16 anyway)
    // Rows are requested ONE step before they are staged (a second register set spilled: 144 weight registers); the HBM round trip is taken
    // earlier by TOUCHING the main-pass lines of the row three steps ahead with one 4-byte load per tensor, so that the real loads are L2 hits
    auto touch_row = [&](int gy) {
        const size_t row = (size_t)clamp_row(gy) * p.win;
        unsigned ta = *reinterpret_cast<const unsigned*>(src_a + (row + offx[0]) * p.a.pitch);
        unsigned tv2 = *reinterpret_cast<const unsigned*>(src_b + (row + offx[0]) * p.b.pitch);
        asm volatile("" ::"v"(ta), "v"(tv2));
        if constexpr (SUM2) {
            unsigned tu2 = *reinterpret_cast<const unsigned*>(src_b2 + (row + offx[0]) * p.b.pitch);
            asm volatile("" ::"v"(tu2));
        }
    };
    RowRegs q;
    {
        RowRegs a1;
        load_row(iy_begin, q);
        load_row(iy_begin + 1, a1);
        touch_row(iy_begin + 3);
        touch_row(iy_begin + 4);
        __syncthreads();                                         // (the tables are in LDS)
        stage(iy_begin, q);
        load_row(iy_begin + 2, q);
        stage(iy_begin + 1, a1);
    }
    unsigned short* const out = reinterpret_cast<unsigned short*>(p.out) + ((size_t)img * p.hout * p.wout + 2 * ix0) * p.pitch_out + 16 * ct;
    const int opix = lane >> 1, oh = lane & 1;                   // the output pixel (of 32) and channel octet this lane stores
    const bool ostore = 2 * ix0 + opix < p.wout;
    auto store_rows = [&](int iy) {                              // the two output rows of input row iy, one step late
        if (ostore) {
#pragma unroll
            for (int pa = 0; pa < 2; ++pa) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(&ep[(pa * 32 + opix) * CT_EPS + oh * 8]);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(&ep[(pa * 32 + opix) * CT_EPS + oh * 8 + 4]);
                const bf16x8 h{(__bf16)v0.x, (__bf16)v0.y, (__bf16)v0.z, (__bf16)v0.w, (__bf16)v1.x, (__bf16)v1.y, (__bf16)v1.z, (__bf16)v1.w};
                *reinterpret_cast<bf16x8*>(out + ((size_t)(2 * iy + pa) * p.wout + opix) * p.pitch_out + oh * 8) = h;
            }
        }
    };
#pragma unroll 1
    for (int iy = iy_begin; iy < iy_end; ++iy) {
        __syncthreads();                                         // rows iy and iy + 1 are complete in the window (both waves' planes)
        if (iy > iy_begin) store_rows(iy - 1);
        f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            bf16x8 a[2][2];
#pragma unroll
            for (int di = 0; di < 2; ++di)
#pragma unroll
                for (int dj = 0; dj < 2; ++dj)
                    a[di][dj] = *reinterpret_cast<const bf16x8*>(ring + ((iy + di) & 3) * CT_ROW + kc * CT_PLANE + (pl + dj) * CT_PITCH + g * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const int pa = (ky == 1) ? 0 : 1, di = (ky == 0) ? 1 : 0;
                const int pb = (kx == 1) ? 0 : 1, dj = (kx == 0) ? 1 : 0;
                acc[pa * 2 + pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[di][dj], b[kc][tap], acc[pa * 2 + pb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // row iy + 2 into the slot row iy - 2 left (both waves finished reading it before this step's barrier), then the load of row iy + 3
        stage(iy + 2, q);
        load_row(iy + 3, q);
        touch_row(iy + 5);
        const bool row_ok = true;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[ph][r] + bias;
                ep[((ph >> 1) * 32 + 2 * (g * 4 + r) + (ph & 1)) * CT_EPS + pl] = v;
                const float vm = (row_ok && ix0 + g * 4 + r < p.win) ? v : 0.f;
                sum += vm;
                sq = fmaf(vm, v, sq);
            }
    }
    __syncthreads();
    store_rows(iy_end - 1);
    if (p.stats) stats_commit_b((double)sum, (double)sq, p.stats + ((size_t)img * p.cout + ct * 16 + pl) * 2, lane);
}

template <int WN, int NTW, int NP, bool A16 = false>
static int launch_tb(hipStream_t s, ConvBParams p) {
    if constexpr (A16 && WN == 2 && NTW == 2) {
        // 64-channel blocks on maps of >= 16 rows with a wide K: the eight-row tile (CS_CONVT_TH8=0: the four-row one everywhere)
        static const int th8_env = getenv("CS_CONVT_TH8") ? atoi(getenv("CS_CONVT_TH8")) : 1;
        if (th8_env && p.hin >= 16 && p.hin % 8 == 0 && p.a.c + p.b.c >= th8_env * 128) {
            p.tiles_y = ceil_div(p.hin, 8);
            dim3 grid8(xcd_grid(p.tiles_x * p.tiles_y * p.n, p.ntile_total / (WN * NTW)));
            if (p.b.ptr2) hipLaunchKernelGGL((convt_bf16s_kernel<WN, NTW, NP, A16, true, 2>), grid8, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((convt_bf16s_kernel<WN, NTW, NP, A16, false, 2>), grid8, dim3(256), 0, s, p);
            CS_HIP(hipGetLastError());
            return CS_OK;
        }
    }
    dim3 grid(xcd_grid(p.tiles_x * p.tiles_y * p.n, p.ntile_total / (WN * NTW)));
    if constexpr (A16) {
        if (p.b.ptr2) {
            hipLaunchKernelGGL((convt_bf16s_kernel<WN, NTW, NP, A16, true>), grid, dim3(256), 0, s, p);
            CS_HIP(hipGetLastError());
            return CS_OK;
        }
    }
    hipLaunchKernelGGL((convt_bf16s_kernel<WN, NTW, NP, A16>), grid, dim3(256), 0, s, p);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int launch_convt_bf16s(hipStream_t s, int planes, const ConvSrc& a, const ConvSrc& b, const unsigned short* wpack, const float* bias,
                       float* out, int pitch_out, int cout, double* stats, int n, int hin, int win, bool act16, const ConvOutExtra* ex) {
    CS_REQUIRE(planes >= 1 && planes <= 4, "convT(bf16): %d planes", planes);
    CS_REQUIRE(!ex || (!ex->out2 && !ex->addend2 && !act16), "convT(bf16): only a plain fp32 addend is understood");
    CS_REQUIRE(!act16 || planes == 1, "convT(bf16): bf16 activations in HBM go with the plain-bf16 mode only");
    CS_REQUIRE(a.h16 == act16 && (b.c == 0 || b.h16 == act16), "convT(bf16): activation element type does not match the mode");
    CS_REQUIRE(a.c % 16 == 0 && b.c % 16 == 0 && a.c > 0 && cout % 16 == 0, "convT(bf16): channels must be multiples of 16");
    CS_REQUIRE(!a.ptr2 && (!b.ptr2 || (act16 && b.c % 32 == 0)), "convT(bf16): a sum source must be the second operand of the bf16 datapath, in whole 32-channel chunks");
    ConvBParams p{};
    p.a = to_devb(a);
    p.b = to_devb(b);
    p.wpack = wpack; p.bias = bias; p.out = out; p.stats = stats; p.dummy = reinterpret_cast<const float*>(wpack);
    p.pitch_out = pitch_out; p.cout = cout; p.ntile_total = cout / 16;
    p.n = n; p.hin = hin; p.win = win; p.hout = 2 * hin; p.wout = 2 * win;
    p.addend = ex ? ex->addend : nullptr;
    p.pitch_add = ex ? ex->pitch_add : 0;
    p.tiles_x = ceil_div(win, 16);
    p.tiles_y = ceil_div(hin, 4);
    const int wn = (cout % 64 == 0) ? 4 : (cout % 32 == 0) ? 2 : 1;
    const double flops = 2.0 * n * hin * win * (double)cout * (a.c + b.c) * 9;
    const double bytes = (act16 ? 2.0 : 4.0) * n * ((double)hin * win * (a.c + (b.ptr2 ? 2 : 1) * b.c) + 4.0 * hin * win * cout);
    ProfScope prof(K_CONVT, s, flops, bytes);
    // the widest decoder level of the 32-base net (64 + 64 -> 32 channels): a pair of waves streams a strip (convt_bf16_rows_kernel)
    if (g_bf16_rows >= 2 && act16 && planes == 1 && a.c == 64 && b.c == 64 && cout == 32 && !a.tab && (b.ptr2 ? (b.tab && b.tab2) : !b.tab) && !ex &&
        a.pitch % 8 == 0 && b.pitch % 8 == 0 && pitch_out % 8 == 0 && b.tab_off % 8 == 0 && hin >= 4 && (size_t)hin * win * (a.pitch > b.pitch ? a.pitch : b.pitch) < ((size_t)1 << 31) &&
        (long long)ceil_div(win, 16) * ceil_div(hin, CT_RS) * n >= STREAM_MIN_BLOCKS) {
        const int strips = ceil_div(win, 16), segs = ceil_div(hin, CT_RS);
        const dim3 grid((unsigned)(strips * segs * n));
        if (b.ptr2) hipLaunchKernelGGL(convt_bf16_rows_kernel<true>, grid, dim3(128), 0, s, p, strips, segs);
        else hipLaunchKernelGGL(convt_bf16_rows_kernel<false>, grid, dim3(128), 0, s, p, strips, segs);
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
#define CS_TB(NP_, A16_)                                             \
    switch (wn) {                                                    \
        case 4: return launch_tb<2, 2, NP_, A16_>(s, p);             \
        case 2: return launch_tb<2, 1, NP_, A16_>(s, p);             \
        default: return launch_tb<1, 1, NP_, A16_>(s, p);            \
    }
    if (act16) { CS_TB(1, true) }
    switch (planes) {
        case 4: CS_TB(4, false)
        case 3: CS_TB(3, false)
        case 2: CS_TB(2, false)
        default: CS_TB(1, false)
    }
#undef CS_TB
}

// ---- weight packing: fp32 PyTorch layout -> 3 bf16 planes in MFMA (16x16x32) fragment order --------------------
size_t packed_bf16_elems(int cout_total, int cin_a, int cin_b, int ksize) {
    const size_t chunks = (size_t)(cin_a + 31) / 32 + (size_t)(cin_b + 31) / 32;
    return chunks * ksize * ksize * (cout_total / 16) * 3 * 512;
}

int pack_conv_weights_bf16(PackCtx* pc, hipStream_t s, const float* w, bool transposed, int cout, int cin, int ksize,
                           unsigned short* packed, int cout_off, int cout_total, int cin_split, bool f16, bool flip, int chunk_off) {
    CS_REQUIRE(cout % 16 == 0 && cin % 16 == 0 && cout_off % 16 == 0 && cout_total % 16 == 0, "pack(bf16): bad channel counts");
    const int taps = ksize * ksize;
    const int chunks_total = cin_split > 0 ? (cin_split + 31) / 32 + (cin - cin_split + 31) / 32 : (cin + 31) / 32;
    return pack_bf16_submit(pc, s, w, transposed, cout, cin, taps, packed, cout_off, cout_total / 16, cin_split, chunks_total, f16, flip, chunk_off);
}

}  // namespace cs
