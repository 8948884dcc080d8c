// gfx950 (MI355X): the first layer of the U-Net (3 -> c conv, 3x3 stride 2 pad 1, and the first ResidualUnit's residual conv of the same
// shape) read from PIXEL RECORDS of the image - the sliding-window gather of baseline/predict.py:109 (monai sliding_window_inference's
// "cat slices" step) fused into the layer as in kernels_conv.hip: conv_first_kernel, inference only.
//
// Why a second form.  conv_first_kernel gathers a 17 x 33 x 3 patch of the planar image per 8 x 16 output pixels with one 4-byte load
// per element (~45 VALU instructions of index arithmetic each), meets at two block barriers and retires after 56 MFMAs per wave: PMC
// (profiles/pmc_mfma_util_r04*.json) shows its waves waiting 41-48 % of their cycles, and the bf16 form ran at a fifth of its HBM
// roofline.  Here
//   * the image is first rewritten ONCE per slide as records - (r, g, b, 0) per pixel, 16 bytes in fp32 / 8 bytes in bf16
//     (image_records_kernel; 0.05 ms for 4096^2) - so a patch pixel is ONE load with two compares;
//   * a wave owns 2 output rows x 16 columns and walks `tiles_per_block` such tiles along the row: its 5 x 33 patch, its im2col reads and
//     its epilogue slab live in wave-private LDS - NO block barrier in the loop, the records of the next tile (and, in the ADD form,
//     the addend of the next tile) are in flight under the MFMAs and the epilogue of the current one;
//   * weights live in registers for the whole walk (fp32: 7 K-steps x NT fragments of the fp32 pack; bf16: three K = 16 steps with the
//     k axis re-ordered as tap x (r, g, b, 0), so that an A fragment is ONE 8-byte LDS read of a record - built from the same fp32 pack);
//   * statistics: fp32 partial sums per tile, fp64 per lane across the walk, one atomic pair per block and channel at the end.
// ADD form: out = conv + bias + act(norm(y)) - the first ResidualUnit's sum (engine.hip: fuse_first), y and out fp32 or bf16.
#include "conv_common.h"

namespace cs {

namespace {

struct FirstRecParams {
    const void* rec;           // float4 (fp32 path) or 4 x bf16 (bf16 datapath) per pixel: [img_h][img_w], or [n][img_h][img_w] without origins
    int64_t rec_sn;            // records between batch items (plain batched forward), 0 with origins
    const int2* origins;       // [n] (y0, x0) of every window, or nullptr
    int img_h, img_w, roi_h, roi_w;
    const float* wpack;        // fp32 pack of the layer: [7 K-steps][nt_pack][64 lanes] (pack_first_elem)
    const float* bias;
    void* out;
    double* stats;
    int pitch_out, stats_stride, stats_c;
    int n, hout, wout, tiles_x, tiles_y, xsplit, tiles_per_block;
    int nt_pack, nt_off;
    SrcDev add;                // ADD: y (raw), its lazy table and slope; add.ptr really points at bf16 in the bf16 form
};

constexpr int R_IW = 33, R_IH = 5, R_NPIX = R_IH * R_IW;        // a wave's patch: 2 output rows x 16 columns at stride 2, 3x3
constexpr int R_ITERS = (R_NPIX + 63) / 64;                     // 3 record loads per lane and tile

// geometry of a wave's walk: which records its lanes load for tile tx (clamped address + validity)
struct RecGeom {
    int row_off[R_ITERS];     // record offset of (row, column of tile 0) inside the image (may be negative left of the image)
    bool row_ok[R_ITERS];     // the row lies inside the window and the image
    int wx0[R_ITERS];         // window-local column of tile 0
    int xlo, xhi;             // valid window-local columns [xlo, xhi)
    __device__ __forceinline__ void init(const FirstRecParams& p, int lane, int oy0, int y0, int x0) {
        xlo = x0 < 0 ? -x0 : 0;
        xhi = p.roi_w < p.img_w - x0 ? p.roi_w : p.img_w - x0;
#pragma unroll
        for (int it = 0; it < R_ITERS; ++it) {
            const int pix = lane + 64 * it;
            const int iy = pix / R_IW, ix = pix - iy * R_IW;
            const int wy = oy0 * 2 - 1 + iy, gy = y0 + wy;
            wx0[it] = ix - 1;
            row_ok[it] = pix < R_NPIX && wy >= 0 && wy < p.roi_h && gy >= 0 && gy < p.img_h;
            row_off[it] = row_ok[it] ? gy * p.img_w + x0 + ix - 1 : 0;
        }
    }
    __device__ __forceinline__ bool ok(int it, int tx) const {
        const int wx = wx0[it] + 32 * tx;
        return row_ok[it] && wx >= xlo && wx < xhi;
    }
};

}  // namespace

// ------------------------------------------------------------------------------------------------ records
// planar (strided) fp32 image -> (r, g, b, 0) records; AS16: bf16 (round to nearest even, as the staging of conv_first16_kernel did)
template <bool AS16>
__global__ __launch_bounds__(256) void image_records_kernel(const float* image, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int h, int w,
                                                            void* rec, int row_begin) {
    // a thread converts 8 pixels of one row, 256 apart (one pixel per thread - 262 k waves of four instructions for a 4096^2 slide - ran
    // at 1 TB/s: wave launches, not bytes, set its pace)
    constexpr int PER = 8;
    const int x0 = blockIdx.x * (256 * PER) + threadIdx.x, y = row_begin + blockIdx.y, img = blockIdx.z;      // rows [row_begin, row_begin + gridDim.y)
    const float* const row = image + img * sn + (int64_t)y * sh;
    const int64_t i0 = ((int64_t)img * h + y) * w;
    f32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int x = x0 + 256 * k;
        const float* src = row + (int64_t)(x < w ? x : 0) * sw;
        v[k] = f32x4{src[0], src[sc], src[2 * sc], 0.f};
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int x = x0 + 256 * k;
        if (x < w) {
            if constexpr (AS16) reinterpret_cast<uint2*>(rec)[i0 + x] = f32x4_to_bf16(v[k]);
            else reinterpret_cast<f32x4*>(rec)[i0 + x] = v[k];
        }
    }
}

int launch_image_records(hipStream_t s, const float* image, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int n, int h, int w, void* rec,
                         bool as16, int row_begin, int row_end) {
    CS_REQUIRE(image && rec && n > 0 && h > 0 && w > 0 && h < 65536 && n < 65536, "image records: bad argument");
    if (row_end < 0) row_end = h;                                // (default: every row)
    CS_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= h, "image records: rows [%d, %d) of %d", row_begin, row_end, h);
    if (row_begin == row_end) return CS_OK;
    const int rows = row_end - row_begin;                        // the record buffer keeps the full image's layout: only these rows are written
    const dim3 grid((unsigned)ceil_div(w, 256 * 8), (unsigned)rows, (unsigned)n);
    ProfScope prof(K_CONV_FIRST, s, 0.0, (double)n * rows * w * (12.0 + (as16 ? 8.0 : 16.0)));
    if (as16) hipLaunchKernelGGL(image_records_kernel<true>, grid, dim3(256), 0, s, image, sn, sc, sh, sw, h, w, rec, row_begin);
    else hipLaunchKernelGGL(image_records_kernel<false>, grid, dim3(256), 0, s, image, sn, sc, sh, sw, h, w, rec, row_begin);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------ shared epilogue pieces
namespace {

// block decode shared by both kernels
struct RecBlock {
    int img, oy0, tx_begin, tx_end, y0, x0;
    const char* rec;
};
template <int REC_BYTES>
__device__ __forceinline__ RecBlock rec_block(const FirstRecParams& p, int wave) {
    RecBlock b;
    int bid = blockIdx.x;
    const int xb = bid % p.xsplit;
    bid /= p.xsplit;
    const int ty = bid % p.tiles_y;
    b.img = bid / p.tiles_y;
    b.oy0 = ty * 8 + wave * 2;
    b.tx_begin = xb * p.tiles_per_block;
    b.tx_end = b.tx_begin + p.tiles_per_block < p.tiles_x ? b.tx_begin + p.tiles_per_block : p.tiles_x;
    b.y0 = 0;
    b.x0 = 0;
    b.rec = reinterpret_cast<const char*>(p.rec);
    if (p.origins) {
        const int2 o = p.origins[b.img];
        b.y0 = o.x;
        b.x0 = o.y;
    } else {
        b.rec += (int64_t)b.img * p.rec_sn * REC_BYTES;
    }
    return b;
}

// the wave's fp64 sums of NT channel tiles -> LDS -> one atomic pair per block and channel
template <int NT>
__device__ __forceinline__ void commit_stats(const FirstRecParams& p, double* st_s, double (&s)[NT], double (&q)[NT], int img, int tid) {
    const int lane = tid & 63, wave = tid >> 6, pl = lane & 15;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        double a = s[nt], b = q[nt];
        a += __shfl_xor(a, 16);
        b += __shfl_xor(b, 16);
        a += __shfl_xor(a, 32);
        b += __shfl_xor(b, 32);
        if (lane < 16) {
            st_s[(wave * 16 * NT + nt * 16 + pl) * 2] = a;
            st_s[(wave * 16 * NT + nt * 16 + pl) * 2 + 1] = b;
        }
    }
    __syncthreads();
    if (tid < 32 * NT && (tid >> 1) < p.stats_c) {
        const int c = tid >> 1, k = tid & 1;
        const double v = st_s[(0 * 16 * NT + c) * 2 + k] + st_s[(1 * 16 * NT + c) * 2 + k] + st_s[(2 * 16 * NT + c) * 2 + k] +
                         st_s[(3 * 16 * NT + c) * 2 + k];
        atomicAdd(p.stats + ((size_t)img * p.stats_stride + c) * 2 + k, v);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ fp32
template <int NT, bool ADD>
__global__ __launch_bounds__(256) void conv_first_rec_kernel(FirstRecParams p) {
    constexpr int MT = 2, KSTEPS = 7;
    constexpr int INW = 512;                                   // floats of a wave's patch (165 x 3, padded)
    constexpr int EPS = 16 * NT + 4, EPW = MT * 16 * EPS;      // the wave's epilogue slab
    constexpr int F4_PER_PIX = NT * 4, PASSES = MT * 16 * F4_PER_PIX / 64;
    static_assert(64 % F4_PER_PIX == 0, "a lane keeps its channel quad through the passes");
    __shared__ __attribute__((aligned(16))) float smem[4 * (INW + EPW) + (ADD ? 0 : 4 * 16 * NT * 2 * 2)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    float* const in_s = smem + wave * (INW + EPW);
    float* const ep = in_s + INW;
    const RecBlock bk = rec_block<16>(p, wave);
    const int rows_valid = p.hout - bk.oy0;                    // <= 0: this wave only joins the statistics barrier

    // weights of the whole walk in registers
    float b[KSTEPS][NT];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[ks][nt] = p.wpack[((size_t)ks * p.nt_pack + p.nt_off + nt) * 64 + lane];
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = p.bias ? p.bias[(p.nt_off + nt) * 16 + pl] : 0.f;
    // im2col offsets of this lane's k = 4 ks + g inside the patch
    int koff[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const int k = 4 * ks + g;
        const int tap = k < 27 ? k / 3 : 0, c = k < 27 ? k - tap * 3 : 0;
        koff[ks] = ((tap / 3) * R_IW + tap % 3) * 3 + c + pl * 6;
    }
    RecGeom geo;
    geo.init(p, lane, bk.oy0, bk.y0, bk.x0);
    const f32x4* const rec = reinterpret_cast<const f32x4*>(bk.rec);
    f32x4 rv[R_ITERS];
    auto load_rec = [&](int tx) {
#pragma unroll
        for (int it = 0; it < R_ITERS; ++it) {
            const bool ok = geo.ok(it, tx);
            rv[it] = rec[ok ? geo.row_off[it] + 32 * tx : 0];
            if (!ok) rv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // ADD: the addend of the same output pixels (raw y + lazy table), one tile ahead
    const int c4 = lane % F4_PER_PIX;
    Lazy4 lz;
    f32x4 yv[ADD ? PASSES : 1];
    const int64_t y_row0 = ((int64_t)bk.img * p.hout + bk.oy0) * p.wout;
    auto load_y = [&](int tx) {
        if constexpr (ADD) {
#pragma unroll
            for (int it = 0; it < PASSES; ++it) {
                const int pix = (lane + 64 * it) / F4_PER_PIX;
                const int m = pix >> 4, x = pix & 15;
                const bool ok = m < rows_valid && tx * 16 + x < p.wout;
                yv[it] = *reinterpret_cast<const f32x4*>(p.add.ptr + (ok ? (y_row0 + (int64_t)m * p.wout + tx * 16 + x) * p.add.pitch + c4 * 4 : 0));
            }
        }
    };
    if constexpr (ADD) lz.load(p.add, bk.img, c4 * 4, p.wpack);
    double ssum[NT], qsum[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum[nt] = qsum[nt] = 0.0;

    if (rows_valid > 0) {
        load_rec(bk.tx_begin);
        load_y(bk.tx_begin);
#pragma unroll 1
        for (int tx = bk.tx_begin; tx < bk.tx_end; ++tx) {
            // patch -> wave-private LDS (LDS operations of one wave complete in order: no wait between this tile's reads and the next stores)
#pragma unroll
            for (int it = 0; it < R_ITERS; ++it) {
                const int pix = lane + 64 * it;
                if (pix < R_NPIX) {
                    in_s[pix * 3 + 0] = rv[it].x;
                    in_s[pix * 3 + 1] = rv[it].y;
                    in_s[pix * 3 + 2] = rv[it].z;
                }
            }
            f32x4 yc[ADD ? PASSES : 1];
            if constexpr (ADD) {
#pragma unroll
                for (int it = 0; it < PASSES; ++it) yc[it] = yv[it];
            }
            if (tx + 1 < bk.tx_end) {
                load_rec(tx + 1);
                load_y(tx + 1);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the patch is wave-private, no block barrier
            __builtin_amdgcn_wave_barrier();
            f32x4 acc[MT][NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float a = in_s[koff[ks] + m * (2 * R_IW * 3)];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma4(a, b[ks][nt], acc[m][nt]);
                }
            }
            const int cols_valid = p.wout - tx * 16;
            // bias, statistics, slab.  The statistics are fp64 from the first addend on (round 6): behind a blank or nearly blank window -
            // the reference leaves zero background at exactly 0, baseline/predict.py:14-21 - this layer's map is the bias plus a few
            // pixels, var << mean^2, and a lane's <= 8 values summed in fp32 (rounds 1-5) put 1e-7 x mean^2 of rounding into a
            // variance of 1e-8: the InstanceNorm scale came out 2e-4 off (tests/test_gpu_blank_windows.py)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[m][nt][r] + bias[nt];
                        ep[(m * 16 + g * 4 + r) * EPS + nt * 16 + pl] = v;
                        if (!ADD && m < rows_valid && g * 4 + r < cols_valid) {
                            const double vd = (double)v;
                            ssum[nt] += vd;
                            qsum[nt] = __builtin_fma(vd, vd, qsum[nt]);
                        }
                    }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            float* const out = reinterpret_cast<float*>(p.out);
#pragma unroll
            for (int it = 0; it < PASSES; ++it) {
                const int pix = (lane + 64 * it) / F4_PER_PIX;
                const int m = pix >> 4, x = pix & 15;
                if (m < rows_valid && x < cols_valid) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&ep[pix * EPS + c4 * 4]);
                    if constexpr (ADD) v = lz.apply(yc[it]) + v;               // (a + b as norm_act_add forms it)
                    *reinterpret_cast<f32x4*>(out + (y_row0 + (int64_t)m * p.wout + tx * 16 + x) * p.pitch_out + c4 * 4) = v;
                }
            }
        }
    }
    if constexpr (!ADD) {
        if (p.stats) commit_stats<NT>(p, reinterpret_cast<double*>(smem + 4 * (INW + EPW)), ssum, qsum, bk.img, tid);
    }
}

// ------------------------------------------------------------------------------------------------ bf16 datapath
typedef short bf16x4s __attribute__((ext_vector_type(4)));
template <int NT, bool ADD>
__global__ __launch_bounds__(256) void conv_first16_rec_kernel(FirstRecParams p) {
    constexpr int MT = 2;
    constexpr int INW = 384;                                   // floats (= 8-byte records / 2) of a wave's patch: 165 records, padded
    constexpr int EPS = 16 * NT + 4, EPW = MT * 16 * EPS;
    constexpr int F4_PER_PIX = NT * 4, PASSES = MT * 16 * F4_PER_PIX / 64;
    static_assert(64 % F4_PER_PIX == 0, "a lane keeps its channel quad through the passes");
    __shared__ __attribute__((aligned(16))) float smem[4 * (INW + EPW) + (ADD ? 0 : 4 * 16 * NT * 2 * 2)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    uint2* const in_s = reinterpret_cast<uint2*>(smem + wave * (INW + EPW));
    float* const ep = smem + wave * (INW + EPW) + INW;
    const RecBlock bk = rec_block<8>(p, wave);
    const int rows_valid = p.hout - bk.oy0;

    // B fragments: K-step i holds taps 4 i ... 4 i + 3, this lane's group g the tap 4 i + g as (r, g, b, 0) - from the fp32 pack
    // (element k = 3 tap + c of channel co sits at [k >> 2][nt][16 (k & 3) + co]), rounded to bf16 as pack_first16_elem rounds
    bf16x4s b[3][NT];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int tap = 4 * i + g;
            unsigned h[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int k = tap < 9 ? 3 * tap + c : 0;
                const float w = p.wpack[((size_t)(k >> 2) * p.nt_pack + p.nt_off + nt) * 64 + 16 * (k & 3) + pl];
                h[c] = tap < 9 ? f32_to_bf16_bits(w) : 0u;
            }
            b[i][nt] = bf16x4s{(short)h[0], (short)h[1], (short)h[2], (short)0};
        }
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = p.bias ? p.bias[(p.nt_off + nt) * 16 + pl] : 0.f;
    int aoff[3];                                               // record of tap 4 i + g for output column pl, row 0 (taps >= 9 meet zero weights)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int tap = 4 * i + g < 9 ? 4 * i + g : 0;
        aoff[i] = (tap / 3) * R_IW + tap % 3 + 2 * pl;
    }
    RecGeom geo;
    geo.init(p, lane, bk.oy0, bk.y0, bk.x0);
    const uint2* const rec = reinterpret_cast<const uint2*>(bk.rec);
    uint2 rv[R_ITERS];
    auto load_rec = [&](int tx) {
#pragma unroll
        for (int it = 0; it < R_ITERS; ++it) {
            const bool ok = geo.ok(it, tx);
            rv[it] = rec[ok ? geo.row_off[it] + 32 * tx : 0];
            if (!ok) rv[it] = make_uint2(0u, 0u);
        }
    };
    const int c4 = lane % F4_PER_PIX;
    Lazy4 lz;
    uint2 yv[ADD ? PASSES : 1];
    const int64_t y_row0 = ((int64_t)bk.img * p.hout + bk.oy0) * p.wout;
    auto load_y = [&](int tx) {
        if constexpr (ADD) {
            const unsigned short* const y16 = reinterpret_cast<const unsigned short*>(p.add.ptr);
#pragma unroll
            for (int it = 0; it < PASSES; ++it) {
                const int pix = (lane + 64 * it) / F4_PER_PIX;
                const int m = pix >> 4, x = pix & 15;
                const bool ok = m < rows_valid && tx * 16 + x < p.wout;
                yv[it] = *reinterpret_cast<const uint2*>(y16 + (ok ? (y_row0 + (int64_t)m * p.wout + tx * 16 + x) * p.add.pitch + c4 * 4 : 0));
            }
        }
    };
    if constexpr (ADD) lz.load(p.add, bk.img, c4 * 4, p.wpack);
    double ssum[NT], qsum[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum[nt] = qsum[nt] = 0.0;

    if (rows_valid > 0) {
        load_rec(bk.tx_begin);
        load_y(bk.tx_begin);
#pragma unroll 1
        for (int tx = bk.tx_begin; tx < bk.tx_end; ++tx) {
#pragma unroll
            for (int it = 0; it < R_ITERS; ++it) {
                const int pix = lane + 64 * it;
                if (pix < R_NPIX) in_s[pix] = rv[it];
            }
            uint2 yc[ADD ? PASSES : 1];
            if constexpr (ADD) {
#pragma unroll
                for (int it = 0; it < PASSES; ++it) yc[it] = yv[it];
            }
            if (tx + 1 < bk.tx_end) {
                load_rec(tx + 1);
                load_y(tx + 1);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the patch is wave-private, no block barrier
            __builtin_amdgcn_wave_barrier();
            f32x4 acc[MT][NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bf16x4s a = __builtin_bit_cast(bf16x4s, in_s[aoff[i] + m * 2 * R_IW]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b[i][nt], acc[m][nt], 0, 0, 0);
                }
            const int cols_valid = p.wout - tx * 16;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {          // statistics in fp64 from the first addend, as in the fp32 kernel above (blank windows)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[m][nt][r] + bias[nt];
                        ep[(m * 16 + g * 4 + r) * EPS + nt * 16 + pl] = v;
                        if (!ADD && m < rows_valid && g * 4 + r < cols_valid) {
                            const double vd = (double)v;
                            ssum[nt] += vd;
                            qsum[nt] = __builtin_fma(vd, vd, qsum[nt]);
                        }
                    }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            unsigned short* const out = reinterpret_cast<unsigned short*>(p.out);
#pragma unroll
            for (int it = 0; it < PASSES; ++it) {
                const int pix = (lane + 64 * it) / F4_PER_PIX;
                const int m = pix >> 4, x = pix & 15;
                if (m < rows_valid && x < cols_valid) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&ep[pix * EPS + c4 * 4]);
                    if constexpr (ADD) v = lz.apply(bf16x4_to_f32(yc[it])) + v;
                    *reinterpret_cast<uint2*>(out + (y_row0 + (int64_t)m * p.wout + tx * 16 + x) * p.pitch_out + c4 * 4) = f32x4_to_bf16(v);
                }
            }
        }
    }
    if constexpr (!ADD) {
        if (p.stats) commit_stats<NT>(p, reinterpret_cast<double*>(smem + 4 * (INW + EPW)), ssum, qsum, bk.img, tid);
    }
}

// ------------------------------------------------------------------------------------------------ launcher
static int g_first_rec = [] { const char* e = getenv("CS_FIRST_REC"); return e ? atoi(e) : 1; }();
bool first_rec_enabled() { return g_first_rec != 0; }

int launch_conv_first_rec(hipStream_t s, const WindowSrc& src, const float* wpack, const float* bias, void* out, int pitch_out, int cout,
                          double* stats, int n, int stats_c, int cout_pack, int cout_off, const ConvSrc* add) {
    const bool a16 = src.rec16;
    CS_REQUIRE(src.rec && wpack && out && n > 0, "first conv(records): null argument");
    CS_REQUIRE(cout == 16 || cout == 32 || cout == 64, "first conv(records): %d output channels per launch unsupported", cout);
    if (cout_pack <= 0) cout_pack = cout;
    CS_REQUIRE(cout_pack % 16 == 0 && cout_off % 16 == 0 && cout_off >= 0 && cout_off + cout <= cout_pack,
               "first conv(records): channels [%d, %d) of a %d-channel pack", cout_off, cout_off + cout, cout_pack);
    CS_REQUIRE(!add || (!stats && add->ptr && add->h16 == a16 && !add->q8 && !add->ptr2 && add->c == cout && add->pitch % 4 == 0),
               "first conv(records): the fused residual sum takes a plain tensor of %d channels in the datapath's own format", cout);
    CS_REQUIRE(pitch_out % 4 == 0, "first conv(records): pixel pitch must be a multiple of 4 elements");
    CS_REQUIRE((int64_t)src.img_h * src.img_w < ((int64_t)1 << 31), "first conv(records): image too large for 32-bit record offsets");
    FirstRecParams p{};
    p.rec = src.rec;
    p.rec_sn = src.origins ? 0 : (int64_t)src.img_h * src.img_w;
    p.origins = src.origins;
    p.img_h = src.img_h; p.img_w = src.img_w; p.roi_h = src.roi_h; p.roi_w = src.roi_w;
    p.wpack = wpack; p.bias = bias; p.out = out; p.stats = stats;
    p.pitch_out = pitch_out; p.stats_stride = cout_pack; p.stats_c = stats_c > 0 ? stats_c : cout;
    p.n = n;
    p.hout = (src.roi_h + 2 - 3) / 2 + 1;
    p.wout = (src.roi_w + 2 - 3) / 2 + 1;
    p.tiles_x = ceil_div(p.wout, 16);
    p.tiles_y = ceil_div(p.hout, 8);
    // tiles a wave walks: 8 (128 output columns = one block per row group) unless the launch would not cover the chip twice
    p.tiles_per_block = 8;                                 // (4 measured 104 / 116 us against 90 / 110 us for the bf16 pair at 256 windows)
    while (p.tiles_per_block > 1 && (int64_t)ceil_div(p.tiles_x, p.tiles_per_block) * p.tiles_y * n < 2 * SMALL_GRID_BLOCKS) p.tiles_per_block /= 2;
    p.xsplit = ceil_div(p.tiles_x, p.tiles_per_block);
    p.nt_pack = cout_pack / 16;
    p.nt_off = cout_off / 16;
    if (add) p.add = SrcDev{add->ptr, add->tab, add->slope, add->c, add->pitch, add->tab_c ? add->tab_c : add->c, add->tab_off, nullptr, nullptr, nullptr};
    const dim3 grid((unsigned)(p.xsplit * p.tiles_y * n));
    const double eb = a16 ? 2.0 : 4.0;
    const double flops = 2.0 * n * p.hout * p.wout * (double)cout * 27;
    const double bytes = n * (eb * 4.0 * src.roi_h * src.roi_w + eb * p.hout * p.wout * cout * (add ? 2.0 : 1.0));
    ProfScope prof(K_CONV_FIRST, s, flops, bytes);
#define CS_FIRST_REC(KERNEL, NT_)                                                                     \
    if (add) hipLaunchKernelGGL((KERNEL<NT_, true>), grid, dim3(256), 0, s, p);                     \
    else hipLaunchKernelGGL((KERNEL<NT_, false>), grid, dim3(256), 0, s, p);
    if (a16) {
        switch (cout / 16) {
            case 1: CS_FIRST_REC(conv_first16_rec_kernel, 1) break;
            case 2: CS_FIRST_REC(conv_first16_rec_kernel, 2) break;
            default: CS_FIRST_REC(conv_first16_rec_kernel, 4) break;
        }
    } else {
        switch (cout / 16) {
            case 1: CS_FIRST_REC(conv_first_rec_kernel, 1) break;
            case 2: CS_FIRST_REC(conv_first_rec_kernel, 2) break;
            default: CS_FIRST_REC(conv_first_rec_kernel, 4) break;
        }
    }
#undef CS_FIRST_REC
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
