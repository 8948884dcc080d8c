// gfx950 fp8 convolution kernels: BASELINE.json configs[4] ("mixed fp8 weights ... on CDNA4 MFMA").
//
// Weights are stored as OCP fp8 (e4m3fn) with ONE fp32 scale per output channel (computed at pack time: amax / 448).
// Conv OUTPUTS are bf16 (raw, pre-norm); every conv INPUT is an activated e4m3 tensor that was materialised once - by
// launch_act_quant (norm table + PReLU + conversion, 2 bytes in / 1 byte out per element) or by the residual add - so the
// values are O(1) and need no scale, and a conv stages its input patch with plain 16-byte copies: no arithmetic, one byte per
// element and channel-tile block.  (The first version converted bf16 -> e4m3 while staging: ~45 VALU instructions per 8 channels,
// repeated by each of the up to 8 output-channel blocks that read a pixel - the conversion, not the matrix pipe, set the pace,
// and the bf16 kernels were faster on the configs[4] net.)  The contraction runs on the
// block-scaled matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (E8M0 0x7f): K = 128 channels per
// instruction at twice the bf16 rate (~5 PFLOP/s dense), fp32 accumulate.  The per-channel weight scale is applied to the
// fp32 accumulator in the epilogue; statistics for the next InstanceNorm come from those fp32 values; the output is bf16.
//
// Operand layout (probed with exact integer data, tests/diagnostics/mfma_fp8_layout_probe.hip): lane l holds
// A[row l & 15][k = 32 (l >> 4) + j] and B[k = 32 (l >> 4) + j][col l & 15] in byte j = 0..31 of its 8-VGPR fragment;
// C/D as every 16x16 MFMA (col = l & 15, row = 4 (l >> 4) + reg).
//
// Kernel frame = conv_bf16s_kernel (kernels_conv_bf16.hip): 4 waves as WN x WM, weights L2 -> registers one tap row ahead
// (never through LDS), input patch double-buffered in LDS, unconditional clamped loads pinned with sched_barriers.
#include <type_traits>

#include "kernels.h"
#include "pack_jobs.h"

namespace cs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int FP8_UNIT_SCALE = 0x7f7f7f7f;      // E8M0 1.0 in every byte

__device__ __forceinline__ f32x4 mfma_fp8(i32x8 a, i32x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, FP8_UNIT_SCALE, 0, FP8_UNIT_SCALE);
}
__device__ __forceinline__ float prelu_8(float v, float slope) { return v >= 0.f ? v : v * slope; }
__device__ __forceinline__ float bf16lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf16hi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
// two floats -> two e4m3 bytes in the low / high half of a dword (saturating: v_cvt_pk_fp8_f32 clamps to +-448)
template <bool HIGH>
__device__ __forceinline__ unsigned pk_fp8(float a, float b, unsigned old) {
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, HIGH);
}

struct SrcDev8 {
    const unsigned char* ptr;    // activated e4m3 elements
    int c, pitch;
};

struct Conv8Params {
    SrcDev8 a, b;
    const unsigned char* wpack;  // [chunk][tap][ntile_total][half][64 lanes][16] e4m3 (pack_conv_weights_fp8)
    const float* wscale;         // [cout] per-output-channel weight scale
    const float* bias;
    float* out;                  // bf16 elements
    double* stats;
    int pitch_out, cout, ntile_total;
    int n, hin, win, hout, wout, tiles_x, tiles_y;
};

struct WorkId8 {
    int spatial, ct;
    bool valid;
};
__device__ __forceinline__ WorkId8 decode_work8(int n_spatial, int n_ct) {   // XCD-aware: see kernels_conv.hip
    const int id = blockIdx.x, lane8 = id & 7, rest = id >> 3;
    const int ct = rest % n_ct, sg = rest / n_ct;
    const int spatial = sg * 8 + lane8;
    return WorkId8{spatial, ct, spatial < n_spatial};
}
static inline unsigned xcd_grid8(int n_spatial, int n_ct) { return (unsigned)(((n_spatial + 7) / 8) * 8 * n_ct); }

__device__ __forceinline__ void stats_commit8(double s, double q, double* dst, int lane) {
    s += __shfl_xor(s, 16);
    q += __shfl_xor(q, 16);
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    if (lane < 16) {
        atomicAdd(dst, s);
        atomicAdd(dst + 1, q);
    }
}

// As on the bf16 datapath (kernels_conv_bf16.hip): a lane's values are summed in fp32 (the outputs are stored as bf16 anyway), the
// waves of a block that hold the same channels leave their sums in LDS and ONE fp64 atomic pair per block and channel follows.
__device__ __forceinline__ void stats_stash8(double s, double q, double* red_c, int lane) {
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
__device__ __forceinline__ void stats_flush8(const double* red, double* dst, int tid) {
    if (tid < NCH * 2) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) v += red[w * NCH * 2 + tid];
        atomicAdd(dst + tid, v);
    }
}

// Staging of one 128-channel chunk: item = (pixel, 16 channels) = one 16-byte load and one 16-byte LDS store.

// MODE 0: conv (S in {1,2}, KS in {3,1});  the transposed conv has its own kernel below.
template <int S, int KS, int WN, int NTW, int THX>
__global__ __launch_bounds__(256, 2) void conv_fp8_kernel(Conv8Params p) {
    constexpr int WM = 4 / WN;
    constexpr int TH = THX * ((S == 1) ? 8 : 4), TW = 16;
    constexpr int MTW = TH / WM;
    constexpr int NTB = WN * NTW;
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
    constexpr int PITCH = 144;                              // bytes per staged pixel: 128 e4m3 + 16 pad (conflict-free b128 reads at S = 1)
    constexpr int PAD = KS / 2, TAPS = KS * KS;
    constexpr int IN_ITEMS = IH * IW * 8;
    constexpr int IN_ITERS = (IN_ITEMS + 255) / 256;
    constexpr int IN_BYTES = (IH * IW * PITCH + 15) / 16 * 16;
    constexpr int EPS = 16 * NTB + 4;
    constexpr int EP_BYTES = TH * 16 * EPS * 4;
    // the input patch is double-buffered only where the epilogue slab already pays for the LDS: a 2 x 43 KB patch (stride 2) left
    // ONE block per CU - one wave per SIMD, every load / barrier latency exposed: 20 us per block for 1 us of matrix work
    constexpr bool DB = 2 * IN_BYTES <= EP_BYTES + 4096;
    constexpr int SM_BYTES = (DB ? 2 : 1) * IN_BYTES > EP_BYTES ? (DB ? 2 : 1) * IN_BYTES : EP_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SM_BYTES];
    __shared__ double red8[WM > 1 ? WM * NTB * 32 : 1];     // per-wave statistics of the block's channels (stats_stash8)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, g = lane >> 4;
    const int wn = wave % WN, wm = wave / WN;
    const WorkId8 wk = decode_work8(p.tiles_x * p.tiles_y * p.n, p.ntile_total / NTB);
    if (!wk.valid) return;
    int bid = wk.spatial;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int nt_w = wk.ct * NTB + wn * NTW;
    const int q = tid & 7;                                   // 16-channel group of this thread inside the 128-channel chunk

    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int goff[IN_ITERS];
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (tid >> 3) + 32 * it;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = oy0 * S - PAD + iy, gx = ox0 * S - PAD + ix;
        goff[it] = (pix < IH * IW && gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win) ? gy * p.win + gx : -1;
    }

    const int chunks_a = (p.a.c + 127) >> 7, chunks = chunks_a + ((p.b.c + 127) >> 7);
    u32x4 rin[IN_ITERS];
    bool qvalid = false;

    auto load_input = [&](int ch) {
        const bool from_a = ch < chunks_a;
        const SrcDev8& src = from_a ? p.a : p.b;
        const int c0 = (from_a ? ch : ch - chunks_a) << 7;
        qvalid = c0 + 16 * q < src.c;                       // channel tails (cin % 128 != 0) are zero-filled
        const int cq = qvalid ? c0 + 16 * q : 0;
        const unsigned char* base = src.ptr + (size_t)img * p.hin * p.win * src.pitch + cq;
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int o = goff[it] >= 0 ? goff[it] : 0;
            rin[it] = *reinterpret_cast<const u32x4*>(base + (size_t)o * src.pitch);
        }
    };
    auto store_input = [&](char* in_s) {
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int pix = (tid >> 3) + 32 * it;
            if (IN_ITEMS % 256 == 0 || pix < IH * IW)
                *reinterpret_cast<u32x4*>(in_s + pix * PITCH + q * 16) = (goff[it] >= 0 && qvalid) ? rin[it] : u32x4{0u, 0u, 0u, 0u};
        }
    };
    // weight fragments of one tap row: KS x NTW fragments of 32 bytes per lane (two lane-linear 16-byte loads each, from L2)
    i32x8 bq[KS][NTW];
    const i32x4* wbase = reinterpret_cast<const i32x4*>(p.wpack) + (size_t)nt_w * 128 + lane;      // [channel tile][half 0 / 1][lane][16 bytes]
    auto load_b = [&](int ch, int ky, int kx) {
        const i32x4* w = wbase + ((size_t)ch * TAPS + ky * KS + kx) * p.ntile_total * 128;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const i32x4 lo = w[t * 128], hi = w[t * 128 + 64];        // each a lane-linear kilobyte: full 128-byte lines per instruction
            bq[kx][t] = i32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        }
    };

    load_input(0);
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) load_b(0, 0, kx);
    store_input(smem);
    __syncthreads();

    const int last_row = chunks * KS - 1;
    for (int ch = 0; ch < chunks; ++ch) {
        const bool more = ch + 1 < chunks;
        load_input(more ? ch + 1 : ch);                     // unconditional (see conv_bf16s_kernel)
        __builtin_amdgcn_sched_barrier(0);
        const char* in_s = smem + (DB ? (ch & 1) * IN_BYTES : 0);
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int row_next = ch * KS + ky + 1 < last_row ? ch * KS + ky + 1 : last_row;
            const int ch_n = row_next / KS, ky_n = row_next - ch_n * KS;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                i32x8 b[NTW];
#pragma unroll
                for (int t = 0; t < NTW; ++t) b[t] = bq[kx][t];
                load_b(ch_n, ky_n, kx);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int MG = (MTW % 2 == 0) ? 2 : 1;
#pragma unroll
                for (int m0 = 0; m0 < MTW; m0 += MG) {
                    i32x8 a[MG];
#pragma unroll
                    for (int mm = 0; mm < MG; ++mm) {
                        const char* ap = in_s + (((wm * MTW + m0 + mm) * S + ky) * IW + pl * S + kx) * PITCH + g * 32;
                        const i32x4 lo = *reinterpret_cast<const i32x4*>(ap), hi = *reinterpret_cast<const i32x4*>(ap + 16);
                        a[mm] = i32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    }
#pragma unroll
                    for (int mm = 0; mm < MG; ++mm)
#pragma unroll
                        for (int t = 0; t < NTW; ++t) {
                            acc[m0 + mm][t] = mfma_fp8(a[mm], b[t], acc[m0 + mm][t]);
                            // pins the MFMA at this point of the program: without it LLVM sinks the whole (side-effect free)
                            // accumulation chain below the staging code, first loads the A fragments of all 9 taps (144+
                            // registers) and spills 2.7 KB per lane
                            asm volatile("" : "+v"(acc[m0 + mm][t]));
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) asm volatile("" ::"v"(rin[it]));
        if (!DB) __syncthreads();
        if (more) store_input(smem + (DB ? ((ch + 1) & 1) * IN_BYTES : 0));
        __syncthreads();
    }

    // ---- epilogue: weight scale, bias, fp64 statistics, bf16 stores through a block-wide LDS slab ----
    float* ep = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int co = (nt_w + t) * 16 + pl;
        const float bias = p.bias ? p.bias[co] : 0.f;
        const float ws = p.wscale[co];
        float sf = 0.f, qf = 0.f;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int row = wm * MTW + m, oy = oy0 + row;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = ox0 + g * 4 + r;
                const float v = acc[m][t][r] * ws + bias;
                ep[(row * 16 + g * 4 + r) * EPS + (wn * NTW + t) * 16 + pl] = v;
                if (p.stats && oy < p.hout && ox < p.wout) {
                    sf += v;
                    qf = fmaf(v, v, qf);
                }
            }
        }
        if (p.stats) {
            if constexpr (WM > 1) stats_stash8((double)sf, (double)qf, red8 + ((wm * NTB + wn * NTW + t) * 16 + pl) * 2, lane);
            else stats_commit8((double)sf, (double)qf, p.stats + ((size_t)img * p.cout + co) * 2, lane);
        }
    }
    __syncthreads();
    if constexpr (WM > 1)
        if (p.stats) stats_flush8<WM, NTB * 16>(red8, p.stats + ((size_t)img * p.cout + wk.ct * NTB * 16) * 2, tid);
    {
        constexpr int F4 = NTB * 4;
        const size_t el0 = (((size_t)img * p.hout + oy0) * p.wout + ox0) * p.pitch_out + wk.ct * NTB * 16;
#pragma unroll
        for (int it_ = 0; it_ < (TH * 16 * F4 + 255) / 256; ++it_) {      // (trip count visible to the unroller: the LDS reads of all passes go out together)
            const int i = tid + 256 * it_;
            if ((TH * 16 * F4) % 256 != 0 && i >= TH * 16 * F4) break;
            const int pix = i / F4, c4 = i - pix * F4;
            const int row = pix >> 4, x = pix & 15;
            if (oy0 + row < p.hout && ox0 + x < p.wout) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&ep[pix * EPS + c4 * 4]);
                const uint2 o = f32x4_to_bf16(v);
                *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(p.out) + el0 + ((int64_t)row * p.wout + x) * p.pitch_out + c4 * 4) =
                    u32x2{o.x, o.y};
            }
        }
    }
}

static SrcDev8 to_dev8(const ConvSrc& s) { return SrcDev8{reinterpret_cast<const unsigned char*>(s.ptr), s.c, s.pitch}; }

template <int S, int KS, int WN, int NTW, int THX = 1>
static int launch_8(hipStream_t s, Conv8Params p) {
    constexpr int TH = THX * ((S == 1) ? 8 : 4);
    p.tiles_x = ceil_div(p.wout, 16);
    p.tiles_y = ceil_div(p.hout, TH);
    dim3 grid(xcd_grid8(p.tiles_x * p.tiles_y * p.n, p.ntile_total / (WN * NTW)));
    hipLaunchKernelGGL((conv_fp8_kernel<S, KS, WN, NTW, THX>), grid, dim3(256), 0, s, p);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int launch_conv_fp8(hipStream_t s, int ksize, int stride, const ConvSrc& a, const ConvSrc& b, const unsigned char* wpack,
                    const float* wscale, const float* bias, float* out, int pitch_out, int cout, double* stats, int n, int hin, int win) {
    CS_REQUIRE(ksize == 3 || ksize == 1, "conv(fp8): ksize %d unsupported", ksize);
    CS_REQUIRE((stride == 1 || stride == 2) && !(ksize == 1 && stride != 1), "conv(fp8): stride %d unsupported", stride);
    CS_REQUIRE(a.c % 16 == 0 && b.c % 16 == 0 && a.c > 0 && cout % 16 == 0, "conv(fp8): channels must be multiples of 16");
    CS_REQUIRE(a.q8 && !a.tab && (b.c == 0 || (b.q8 && !b.tab)), "conv(fp8): inputs must be activated e4m3 tensors (launch_act_quant)");
    CS_REQUIRE(a.pitch % 16 == 0 && (b.c == 0 || b.pitch % 16 == 0), "conv(fp8): pixel pitch must be a multiple of 16 bytes");
    Conv8Params p{};
    p.a = to_dev8(a);
    p.b = to_dev8(b);
    p.wpack = wpack; p.wscale = wscale; p.bias = bias; p.out = out; p.stats = stats;
    p.pitch_out = pitch_out; p.cout = cout; p.ntile_total = cout / 16;
    p.n = n; p.hin = hin; p.win = win;
    const int pad = ksize / 2;
    p.hout = (hin + 2 * pad - ksize) / stride + 1;
    p.wout = (win + 2 * pad - ksize) / stride + 1;
    const int wn = (cout % 64 == 0) ? 4 : (cout % 32 == 0) ? 2 : 1;
    // FLOPs are the ALGORITHMIC ones (2 * cin * ...), not the zero-padded K the matrix pipe executes
    const double flops = 2.0 * n * p.hout * p.wout * (double)cout * (a.c + b.c) * ksize * ksize;
    const double bytes = n * (1.0 * hin * win * (a.c + b.c) + 2.0 * p.hout * p.wout * cout);
    ProfScope prof(ksize == 1 ? K_CONV1_FP8 : (stride == 1 ? K_CONV3_S1_FP8 : K_CONV3_S2_FP8), s, flops, bytes);
    const bool tall = stride == 1 && p.hout >= 16;
#define CS_8(S_, KS_)                                                               \
    switch (wn) {                                                                   \
        case 4: if (S_ == 1 && KS_ == 3 && cout % 128 == 0) return launch_8<S_, KS_, 4, 2>(s, p);  \
                if (tall && KS_ == 3) return launch_8<S_, KS_, 2, 2, 2>(s, p);      \
                if (S_ == 2 && cout % 128 == 0) return launch_8<S_, KS_, 4, 2>(s, p);  \
                return launch_8<S_, KS_, 2, 2>(s, p);                               \
        case 2: if (tall && KS_ == 3) return launch_8<S_, KS_, 1, 2, 2>(s, p);      \
                return launch_8<S_, KS_, 2, 1>(s, p);                               \
        default: return launch_8<S_, KS_, 1, 1>(s, p);                              \
    }
    if (ksize == 1) { CS_8(1, 1) }
    if (stride == 1) { CS_8(1, 3) }
    CS_8(2, 3)
#undef CS_8
}

// ------------------------------------------------------------------------------------------------
// transposed conv k3 s2 p1 op1 in fp8: 4 output phases per input pixel (see convt_mfma_kernel)
// ------------------------------------------------------------------------------------------------
// THX = 2 (round 6, as convt_bf16s_kernel): eight input rows per block - every weight fragment a wave fetches from L2 feeds twice the
// MFMAs; the epilogue slab stays the four-row one and is used twice.
template <int WN, int NTW, int THX = 1>
__global__ __launch_bounds__(256, 2) void convt_fp8_kernel(Conv8Params p) {
    constexpr int WM = 4 / WN, TH = 4 * THX, TW = 16, MTW = TH / WM, NTB = WN * NTW;
    constexpr int IH = TH + 1, IW = TW + 1, PITCH = 144;
    constexpr int IN_ITEMS = IH * IW * 8, IN_ITERS = (IN_ITEMS + 255) / 256;
    constexpr int IN_BYTES = (IH * IW * PITCH + 15) / 16 * 16;
    constexpr int EPS = 16 * NTB + 4;
    constexpr int EPH = 4;                                   // input rows per epilogue pass
    constexpr int EP_BYTES = 2 * EPH * 32 * EPS * 4;
    constexpr int SM_BYTES = IN_BYTES > EP_BYTES ? IN_BYTES : EP_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SM_BYTES];
    __shared__ double red8[WM > 1 ? WM * NTB * 32 : 1];     // per-wave statistics of the block's channels (stats_stash8)
    char* const in_s = smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, g = lane >> 4;
    const int wn = wave % WN, wm = wave / WN;
    const WorkId8 wk = decode_work8(p.tiles_x * p.tiles_y * p.n, p.ntile_total / NTB);
    if (!wk.valid) return;
    int bid = wk.spatial;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int img = bid / p.tiles_y;
    const int iy0 = ty * TH, ix0 = tx * TW;
    const int nt_w = wk.ct * NTB + wn * NTW;
    const int q = tid & 7;

    f32x4 acc[MTW][4][NTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[m][ph][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int goff[IN_ITERS];
#pragma unroll
    for (int it = 0; it < IN_ITERS; ++it) {
        const int pix = (tid >> 3) + 32 * it;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        goff[it] = (pix < IH * IW && gy < p.hin && gx < p.win) ? gy * p.win + gx : -1;
    }
    const int chunks_a = (p.a.c + 127) >> 7, chunks = chunks_a + ((p.b.c + 127) >> 7);
    u32x4 rin[IN_ITERS];
    bool qvalid = false;
    auto load_input = [&](int ch) {
        const bool from_a = ch < chunks_a;
        const SrcDev8& src = from_a ? p.a : p.b;
        const int c0 = (from_a ? ch : ch - chunks_a) << 7;
        qvalid = c0 + 16 * q < src.c;
        const int cq = qvalid ? c0 + 16 * q : 0;
        const unsigned char* base = src.ptr + (size_t)img * p.hin * p.win * src.pitch + cq;
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int o = goff[it] >= 0 ? goff[it] : 0;
            rin[it] = *reinterpret_cast<const u32x4*>(base + (size_t)o * src.pitch);
        }
    };
    auto store_input = [&]() {
#pragma unroll
        for (int it = 0; it < IN_ITERS; ++it) {
            const int pix = (tid >> 3) + 32 * it;
            if (IN_ITEMS % 256 == 0 || pix < IH * IW)
                *reinterpret_cast<u32x4*>(in_s + pix * PITCH + q * 16) = (goff[it] >= 0 && qvalid) ? rin[it] : u32x4{0u, 0u, 0u, 0u};
        }
    };
    i32x8 bq[3][NTW];
    const i32x4* wbase = reinterpret_cast<const i32x4*>(p.wpack) + (size_t)nt_w * 128 + lane;      // [channel tile][half 0 / 1][lane][16 bytes]
    auto load_b = [&](int ch, int tap, int slot) {
        const i32x4* w = wbase + ((size_t)ch * 9 + tap) * p.ntile_total * 128;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const i32x4 lo = w[t * 128], hi = w[t * 128 + 64];        // each a lane-linear kilobyte: full 128-byte lines per instruction
            bq[slot][t] = i32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        }
    };

    load_input(0);
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) load_b(0, tap, tap);
    store_input();
    __syncthreads();
    const int last_tap = chunks * 9 - 1;
    for (int ch = 0; ch < chunks; ++ch) {
        const bool more = ch + 1 < chunks;
        load_input(more ? ch + 1 : ch);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int pa = (ky == 1) ? 0 : 1, di = (ky == 0) ? 1 : 0;
            const int pb = (kx == 1) ? 0 : 1, dj = (kx == 0) ? 1 : 0;
            const int ph = pa * 2 + pb;
            i32x8 b[NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t) b[t] = bq[tap % 3][t];
            {
                const int nxt = ch * 9 + tap + 3 < last_tap ? ch * 9 + tap + 3 : last_tap;
                load_b(nxt / 9, nxt % 9, tap % 3);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int r = wm * MTW + m;
                const char* ap = in_s + ((r + di) * IW + pl + dj) * PITCH + g * 32;
                const i32x4 lo = *reinterpret_cast<const i32x4*>(ap), hi = *reinterpret_cast<const i32x4*>(ap + 16);
                const i32x8 a = i32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    acc[m][ph][t] = mfma_fp8(a, b[t], acc[m][ph][t]);
                    asm volatile("" : "+v"(acc[m][ph][t]));   // pinned: see conv_fp8_kernel
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
    float sf_[NTW], qf_[NTW], bias_[NTW], ws_[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int co = (nt_w + t) * 16 + pl;
        sf_[t] = 0.f; qf_[t] = 0.f;
        bias_[t] = p.bias ? p.bias[co] : 0.f;
        ws_[t] = p.wscale[co];
    }
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
                        const float v = acc[m][ph][t][r] * ws_[t] + bias_[t];
                        ep[((2 * lrow + (ph >> 1)) * 32 + 2 * (g * 4 + r) + (ph & 1)) * EPS + (wn * NTW + t) * 16 + pl] = v;
                        if (p.stats && iy < p.hin && ix < p.win) {
                            sf_[t] += v;
                            qf_[t] = fmaf(v, v, qf_[t]);
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
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&ep[pix * EPS + c4 * 4]);
                    const uint2 o = f32x4_to_bf16(v);
                    *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(p.out) + el0 + ((int64_t)row * p.wout + x) * p.pitch_out + c4 * 4) =
                        u32x2{o.x, o.y};
                }
            }
        }
    }
    if (p.stats) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int co = (nt_w + t) * 16 + pl;
            if constexpr (WM > 1) stats_stash8((double)sf_[t], (double)qf_[t], red8 + ((wm * NTB + wn * NTW + t) * 16 + pl) * 2, lane);
            else stats_commit8((double)sf_[t], (double)qf_[t], p.stats + ((size_t)img * p.cout + co) * 2, lane);
        }
    }
    if constexpr (WM > 1) {
        __syncthreads();
        if (p.stats) stats_flush8<WM, NTB * 16>(red8, p.stats + ((size_t)img * p.cout + wk.ct * NTB * 16) * 2, tid);
    }
}

template <int WN, int NTW>
static int launch_t8(hipStream_t s, Conv8Params p) {
    if constexpr (WN == 2 && NTW == 2) {
        static const int th8_env = getenv("CS_CONVT_TH8") ? atoi(getenv("CS_CONVT_TH8")) : 1;     // 0: the four-row tile everywhere
        if (th8_env && p.hin >= 16 && p.hin % 8 == 0) {
            p.tiles_y = ceil_div(p.hin, 8);
            dim3 grid8(xcd_grid8(p.tiles_x * p.tiles_y * p.n, p.ntile_total / (WN * NTW)));
            hipLaunchKernelGGL((convt_fp8_kernel<WN, NTW, 2>), grid8, dim3(256), 0, s, p);
            CS_HIP(hipGetLastError());
            return CS_OK;
        }
    }
    dim3 grid(xcd_grid8(p.tiles_x * p.tiles_y * p.n, p.ntile_total / (WN * NTW)));
    hipLaunchKernelGGL((convt_fp8_kernel<WN, NTW>), grid, dim3(256), 0, s, p);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int launch_convt_fp8(hipStream_t s, const ConvSrc& a, const ConvSrc& b, const unsigned char* wpack, const float* wscale, const float* bias,
                     float* out, int pitch_out, int cout, double* stats, int n, int hin, int win) {
    CS_REQUIRE(a.c % 16 == 0 && b.c % 16 == 0 && a.c > 0 && cout % 16 == 0, "convT(fp8): channels must be multiples of 16");
    CS_REQUIRE(a.q8 && !a.tab && (b.c == 0 || (b.q8 && !b.tab)), "convT(fp8): inputs must be activated e4m3 tensors (launch_act_quant)");
    CS_REQUIRE(a.pitch % 16 == 0 && (b.c == 0 || b.pitch % 16 == 0), "convT(fp8): pixel pitch must be a multiple of 16 bytes");
    Conv8Params p{};
    p.a = to_dev8(a);
    p.b = to_dev8(b);
    p.wpack = wpack; p.wscale = wscale; p.bias = bias; p.out = out; p.stats = stats;
    p.pitch_out = pitch_out; p.cout = cout; p.ntile_total = cout / 16;
    p.n = n; p.hin = hin; p.win = win; p.hout = 2 * hin; p.wout = 2 * win;
    p.tiles_x = ceil_div(win, 16);
    p.tiles_y = ceil_div(hin, 4);
    const int wn = (cout % 64 == 0) ? 4 : (cout % 32 == 0) ? 2 : 1;
    const double flops = 2.0 * n * hin * win * (double)cout * (a.c + b.c) * 9;
    const double bytes = n * (1.0 * hin * win * (a.c + b.c) + 2.0 * 4.0 * hin * win * cout);
    ProfScope prof(K_CONVT_FP8, s, flops, bytes);
    switch (wn) {
        case 4: return launch_t8<2, 2>(s, p);
        case 2: return launch_t8<2, 1>(s, p);
        default: return launch_t8<1, 1>(s, p);
    }
}

// ------------------------------------------------------------------------------------------------
// weight packing: fp32 PyTorch layout -> per-output-channel scale + e4m3 fragments
//   packed[chunk][tap][ntile_total][half][lane][16]: cout = ntile*16 + (lane & 15), cin = chunk*128 + 32 (lane >> 4) + 16 half + j
//   (round 6: the two 16-byte halves of a lane's 32-byte fragment are lane-linear kilobytes of their own, so that each load instruction
//   of a wave reads whole 128-byte lines; before, lane l read bytes 32 l ... 32 l + 15 and 32 l + 16 ... + 31 in two half-line passes)
//   input channels >= cin_split (second concat source) start a fresh 128-channel chunk, as the kernels stage them
// ------------------------------------------------------------------------------------------------
size_t packed_fp8_bytes(int cout_total, int cin_a, int cin_b, int ksize) {
    const size_t chunks = (size_t)(cin_a + 127) / 128 + (size_t)(cin_b + 127) / 128;
    return chunks * ksize * ksize * (cout_total / 16) * 2048;
}

// one block per output channel: amax over (cin, taps) -> scale = amax / 448 (1 for an all-zero channel)
__global__ __launch_bounds__(256) void fp8_wscale_kernel(const float* w, int transposed, int cout, int cin, int taps, float* wscale) {
    __shared__ float red[4];
    const int co = blockIdx.x;
    float m = 0.f;
    for (int i = threadIdx.x; i < cin * taps; i += 256) {
        const int ci = i / taps, tp = i - ci * taps;
        const float v = transposed ? w[((size_t)ci * cout + co) * taps + tp] : w[((size_t)co * cin + ci) * taps + tp];
        m = fmaxf(m, fabsf(v));
    }
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float a = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        wscale[co] = a > 0.f ? a * (1.f / 448.f) : 1.f;
    }
}

// one thread per packed dword (4 consecutive k of one lane)
__global__ __launch_bounds__(256) void fp8_pack_kernel(const float* w, int transposed, int cout, int cin, int taps, int cin_split, int chunks_total,
                                                       const float* wscale, unsigned* packed, int cout_off, int ntile_total, long long total) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int jd = e & 7, lane = (e >> 3) & 63;
    long long r = e >> 9;
    const int ntl = r % (cout / 16);
    r /= (cout / 16);
    const int tap = r % taps;
    const int chunk = (int)(r / taps);
    const int chunks_a = cin_split > 0 ? (cin_split + 127) / 128 : chunks_total;
    const int co = ntl * 16 + (lane & 15);
    const float inv = 1.f / wscale[cout_off + co];
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = (chunk < chunks_a ? chunk * 128 : (chunk - chunks_a) * 128) + 32 * (lane >> 4) + 4 * jd + j;
        const int ci = chunk < chunks_a ? k : cin_split + k;
        const bool valid = chunk < chunks_a ? k < (cin_split > 0 ? cin_split : cin) : ci < cin;
        v[j] = valid ? (transposed ? w[((size_t)ci * cout + co) * taps + tap] : w[((size_t)co * cin + ci) * taps + tap]) * inv : 0.f;
    }
    unsigned o = pk_fp8<false>(v[0], v[1], 0u);
    o = pk_fp8<true>(v[2], v[3], o);
    packed[(((((size_t)chunk * taps + tap) * ntile_total + cout_off / 16 + ntl) * 2 + (jd >> 2)) * 64 + lane) * 4 + (jd & 3)] = o;
}

// Not batched: the fp8 packing is a load-time (inference) step, two small kernels per tensor on the caller's stream.
int pack_conv_weights_fp8(hipStream_t s, const float* w, bool transposed, int cout, int cin, int ksize, unsigned char* packed, float* wscale,
                          int cout_off, int cout_total, int cin_split) {
    CS_REQUIRE(cout % 16 == 0 && cin % 16 == 0 && cout_off % 16 == 0 && cout_total % 16 == 0, "pack(fp8): bad channel counts");
    const int taps = ksize * ksize;
    const int chunks_total = cin_split > 0 ? (cin_split + 127) / 128 + (cin - cin_split + 127) / 128 : (cin + 127) / 128;
    const long long total = (long long)chunks_total * taps * (cout / 16) * 512;
    ProfScope prof(K_PACK, s, 0, 4.0 * cout * cin * taps + 4.0 * total);
    hipLaunchKernelGGL(fp8_wscale_kernel, dim3((unsigned)cout), dim3(256), 0, s, w, transposed ? 1 : 0, cout, cin, taps, wscale + cout_off);
    hipLaunchKernelGGL(fp8_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, transposed ? 1 : 0, cout, cin, taps, cin_split,
                       chunks_total, wscale, reinterpret_cast<unsigned*>(packed), cout_off, cout_total / 16, total);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
