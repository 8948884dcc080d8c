// Weight-packing jobs: element functions of every packed layout, run by ONE batched kernel (kernels_conv.hip).
#pragma once
#include <vector>

#include "cs_common.h"

namespace cs {

enum PackKind { PK_CONV = 0, PK_FIRST = 1, PK_BF16 = 2, PK_COPY = 3, PK_FIRST16 = 4, PK_WINO = 5, PK_WINO4 = 6, PK_CONVTW = 7 };
struct PackJob {
    const float* src;
    void* dst;
    void* dst2;                 // PK_CONV: second packing of the same tensor for the data-gradient launch (roles of the two channel
                                // axes swapped), or null
    long long begin;            // first flattened element (element-wise kinds) / first block (PK_CONV) of this job
    int kind, transposed, cout, cin, taps, cout_off, ntile_total, flip, chunk_off, cin_split, chunks_total, f16, ksteps;
    int flip2, chunk_off2, ntile_total2;
    void* dstw;                 // PK_CONV, 3x3 stride-1 layers: the same tile also written as Winograd weights G g G^T (forward) ...
    void* dstw2;                // ... and as the Winograd weights of the data-gradient convolution; either may be null
};

// Per-handle packing state (no process-global state): the job list of the batch being recorded, the pinned staging
// copy of the job table and its device twin.  Both tables are allocated ONCE, when the handle is created, on the handle's device.
struct PackCtx {
    std::vector<PackJob> jobs;
    bool recording = false;
    PackJob* host = nullptr;
    PackJob* dev = nullptr;
    size_t cap = 0;
    hipEvent_t copied = nullptr;   // the last H2D copy out of `host` (the next flush rewrites it)
    bool copied_pending = false;
    int device = -1;
    // layout of the job table that was uploaded last (element-wise jobs first, tile jobs behind them): what pack_replay re-runs
    size_t n_elem = 0, n_jobs = 0;
    long long total_elems = 0, total_blocks = 0;
    double bytes = 0.0;
    bool frozen = false;           // pack_batch_freeze: the device table is final, pack_replay launches the kernels only
};

__device__ __forceinline__ void pack_first_elem(const PackJob& jb, int64_t e) {
    const int cout = jb.cout, cin = jb.cin;
    const int lane = e & 63;
    int r = (int)(e >> 6);
    const int ntl = r % (cout / 16);
    const int ks = r / (cout / 16);
    const int k = 4 * ks + (lane >> 4);
    float v = 0.f;
    if (k < 9 * cin) {
        const int tap = k / cin, c = k - tap * cin, co = ntl * 16 + (lane & 15);
        v = jb.src[((size_t)co * cin + c) * 9 + tap];
    }
    reinterpret_cast<float*>(jb.dst)[((size_t)ks * jb.ntile_total + jb.cout_off / 16 + ntl) * 64 + lane] = v;
}


// first layer for the bf16 / fp8 datapaths: ONE 16x16x32 bf16 K-step holds all 9 * cin (27) products, zero padded to 32.
// packed[ntile_total][lane][8]: cout = ntile*16 + (lane & 15), k = 8 (lane >> 4) + j = tap * cin + c
__device__ __forceinline__ void pack_first16_elem(const PackJob& jb, int64_t e) {
    const int cin = jb.cin;
    const int j = e & 7, lane = (e >> 3) & 63, ntl = (int)(e >> 9);
    const int k = 8 * (lane >> 4) + j;
    float v = 0.f;
    if (k < 9 * cin) {
        const int tap = k / cin, c = k - tap * cin, co = ntl * 16 + (lane & 15);
        v = jb.src[((size_t)co * cin + c) * 9 + tap];
    }
    const __bf16 h = (__bf16)v;
    reinterpret_cast<unsigned short*>(jb.dst)[((size_t)(jb.cout_off / 16 + ntl) * 64 + lane) * 8 + j] = __builtin_bit_cast(unsigned short, h);
}

// split-bf16 / split-fp16 weight planes in 16x16x32 fragment order:
// packed[chunk][tap][ntile_total][plane][lane][j]: cout = ntile*16 + (lane & 15), k = chunk*32 + 8*(lane>>4) + j.
// Input channels >= cin_split (second concat source) start a fresh 32-channel chunk, as the kernel stages them.
constexpr float PACK_F16_W_SCALE = 256.f;                     // == F16_W_SCALE of kernels_conv_bf16.hip
// one thread = one OCTET: the 8 consecutive k of a lane's fragment register pair, i.e. 16 bytes per plane
__device__ __forceinline__ void pack_bf16_elem(const PackJob& jb, int64_t o) {
    const int cout = jb.cout, cin = jb.cin, taps = jb.taps, cin_split = jb.cin_split, chunks_total = jb.chunks_total;
    const int lane = o & 63;
    int64_t r = o >> 6;
    const int ntl = r % (cout / 16);
    r /= (cout / 16);
    const int tap = r % taps;
    const int chunk = r / taps;
    const int chunks_a = cin_split > 0 ? (cin_split + 31) / 32 : chunks_total;
    const int k0 = (chunk < chunks_a ? chunk * 32 : (chunk - chunks_a) * 32) + 8 * (lane >> 4);
    const int co = ntl * 16 + (lane & 15);
    const int ts = jb.flip ? taps - 1 - tap : tap;            // data-gradient packing of a stride-1 conv: taps reversed
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const int ci = chunk < chunks_a ? k : cin_split + k;
        const bool valid = chunk < chunks_a ? k < (cin_split > 0 ? cin_split : cin) : ci < cin;
        v[j] = 0.f;
        if (valid) v[j] = jb.transposed ? jb.src[((size_t)ci * cout + co) * taps + ts] : jb.src[((size_t)co * cin + ci) * taps + ts];
    }
    typedef unsigned pk_u32x4 __attribute__((ext_vector_type(4)));
    pk_u32x4* packed = reinterpret_cast<pk_u32x4*>(jb.dst);
    const size_t base = (((size_t)(chunk + jb.chunk_off) * taps + tap) * jb.ntile_total + jb.cout_off / 16 + ntl) * 3;   // chunk_off: this tensor
                                                            // is the second input-channel source of a fused launch (data-gradient packings)
    unsigned short h[3][8];
    if (jb.f16) {                                             // fp16x3: planes of w * 2^8: h0, remainder, unused
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = __builtin_amdgcn_fmed3f(v[j] * PACK_F16_W_SCALE, -65504.f, 65504.f);   // saturate (|w| >= 255.9 is outside the mode's domain)
            const _Float16 h0 = (_Float16)w;
            const _Float16 h1 = (_Float16)(w - (float)h0);
            h[0][j] = __builtin_bit_cast(unsigned short, h0);
            h[1][j] = __builtin_bit_cast(unsigned short, h1);
            h[2][j] = 0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float w = v[j];
#pragma unroll
            for (int pn = 0; pn < 3; ++pn) {
                const __bf16 hb = (__bf16)w;
                w -= (float)hb;
                h[pn][j] = __builtin_bit_cast(unsigned short, hb);
            }
        }
    }
#pragma unroll
    for (int pn = 0; pn < 3; ++pn) {
        pk_u32x4 q;
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = (unsigned)h[pn][2 * i] | ((unsigned)h[pn][2 * i + 1] << 16);
        packed[(base + pn) * 64 + lane] = q;
    }
}

// U = G g G^T of one 3x3 kernel (fp64, rounded once); g[a * 3 + b]
__device__ __forceinline__ void wino_weight_transform(const float (&g)[9], float (&u)[16]) {
    double col[4][3];                                         // (G g)[xi][b]
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const double g0 = g[b], g1 = g[3 + b], g2 = g[6 + b];
        col[0][b] = g0;
        col[1][b] = 0.5 * (g0 + g1 + g2);
        col[2][b] = 0.5 * (g0 - g1 + g2);
        col[3][b] = g2;
    }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        u[xi * 4 + 0] = (float)col[xi][0];
        u[xi * 4 + 1] = (float)(0.5 * (col[xi][0] + col[xi][1] + col[xi][2]));
        u[xi * 4 + 2] = (float)(0.5 * (col[xi][0] - col[xi][1] + col[xi][2]));
        u[xi * 4 + 3] = (float)col[xi][2];
    }
}

// Winograd F(2x2, 3x3) weights U = G g G^T (fp64, rounded once) in fp32 MFMA fragment order (kernels_conv_wino.hip):
// packed[chunk][position xi*4+nu][ntile_total][lane][j]: cout = ntile*16 + (lane & 15), cin = chunk*16 + 4*(lane>>4) + j.
// transposed + flip = the data-gradient convolution of the same tensor (channel roles swapped, taps reversed).
__device__ __forceinline__ void pack_wino_elem(const PackJob& jb, int64_t e) {
    const int cout = jb.cout, cin = jb.cin;
    const int j = e & 3, lane = (e >> 2) & 63;
    int64_t r = e >> 8;
    const int ntl = r % (cout / 16);
    r /= (cout / 16);
    const int pos = r & 15;
    const int chunk = (int)(r >> 4);
    const int co = ntl * 16 + (lane & 15), ci = chunk * 16 + 4 * (lane >> 4) + j;
    const float* src = jb.transposed ? jb.src + ((size_t)ci * cout + co) * 9 : jb.src + ((size_t)co * cin + ci) * 9;
    double gk[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) gk[a][b] = (double)src[jb.flip ? 8 - (a * 3 + b) : a * 3 + b];
    const int xi = pos >> 2, nu = pos & 3;
    double col[3];                                            // (G g)[xi][b]
#pragma unroll
    for (int b = 0; b < 3; ++b)
        col[b] = xi == 0 ? gk[0][b] : xi == 3 ? gk[2][b] : 0.5 * (xi == 1 ? gk[0][b] + gk[1][b] + gk[2][b] : gk[0][b] - gk[1][b] + gk[2][b]);
    const double u = nu == 0 ? col[0] : nu == 3 ? col[2] : 0.5 * (nu == 1 ? col[0] + col[1] + col[2] : col[0] - col[1] + col[2]);
    reinterpret_cast<float*>(jb.dst)[(((size_t)chunk * 16 + pos) * jb.ntile_total + ntl) * 256 + lane * 4 + j] = (float)u;
}

// Winograd F(4x4, 3x3) weights U = G g G^T (6 x 6 positions, fp64, rounded once) in fp32 MFMA fragment order:
//   jb.f16 == 0 (16x16x4 form of kernels_conv_wino4x.hip): packed[chunk][position xi*6+nu][ntile_total][lane][j],
//                cout = ntile*16 + (lane & 15), cin = chunk*16 + 4*(lane>>4) + j;
//   jb.f16 == 1 (kernels_conv_wino4x.hip, 32x32x2): packed[chunk][position][cout/32][u 2][lane][j],
//                cout = nt32*32 + (lane & 31), cin = chunk*16 + 2*(4u + j) + (lane>>5)  (k-step 4u + j of the chunk).
__device__ __forceinline__ void pack_wino4_elem(const PackJob& jb, int64_t e) {
    const int cout = jb.cout, cin = jb.cin;
    const int j = e & 3, lane = (e >> 2) & 63;
    int64_t r = e >> 8;
    int co, ci, pos, chunk;
    if (jb.f16) {
        const int u = (int)(r & 1);
        r >>= 1;
        const int nt32 = (int)(r % (cout / 32));
        r /= (cout / 32);
        pos = (int)(r % 36);
        chunk = (int)(r / 36);
        co = nt32 * 32 + (lane & 31);
        ci = chunk * 16 + 2 * (4 * u + j) + (lane >> 5);
    } else {
        const int ntl = (int)(r % (cout / 16));
        r /= (cout / 16);
        pos = (int)(r % 36);
        chunk = (int)(r / 36);
        co = ntl * 16 + (lane & 15);
        ci = chunk * 16 + 4 * (lane >> 4) + j;
    }
    // transposed + flip = the data-gradient convolution of the same tensor (channel roles swapped, taps reversed)
    const float* src = jb.transposed ? jb.src + ((size_t)ci * cout + co) * 9 : jb.src + ((size_t)co * cin + ci) * 9;
    const int xi = pos / 6, nu = pos - xi * 6;
    // row xi of G: {1/4, 0, 0}, {-1/6, -1/6, -1/6}, {-1/6, 1/6, -1/6}, {1/24, 1/12, 1/6}, {1/24, -1/12, 1/6}, {0, 0, 1}
    auto grow = [](int k, double (&o)[3]) {
        switch (k) {
            case 0: o[0] = 0.25; o[1] = 0.0; o[2] = 0.0; break;
            case 1: o[0] = -1.0 / 6; o[1] = -1.0 / 6; o[2] = -1.0 / 6; break;
            case 2: o[0] = -1.0 / 6; o[1] = 1.0 / 6; o[2] = -1.0 / 6; break;
            case 3: o[0] = 1.0 / 24; o[1] = 1.0 / 12; o[2] = 1.0 / 6; break;
            case 4: o[0] = 1.0 / 24; o[1] = -1.0 / 12; o[2] = 1.0 / 6; break;
            default: o[0] = 0.0; o[1] = 0.0; o[2] = 1.0; break;
        }
    };
    double ga[3], gb[3];
    grow(xi, ga);
    grow(nu, gb);
    double u = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) u += ga[a] * (double)src[jb.flip ? 8 - (a * 3 + b) : a * 3 + b] * gb[b];
    // both orders are a permutation of the same element range: the flattened element index IS the destination
    reinterpret_cast<float*>(jb.dst)[e] = (float)u;
}


// The F(2,2) forms of the stride-2 layers: the 25 products (row form a) x (column form b), U[a][b] = sum over (ky, kx) of
// rho_a[ky] rho_b[kx] W[ky][kx], summed in fp64, rounded once.
//   transposed conv k3 s2 p1 op1 (kernels_convt_wino.hip; jb.transposed = 1: W[cin][cout][3][3], jb.flip = 0):
//       rho_0 = rho_1 = e_1 (one-tap phase), rho_2 = e_2, rho_3 = e_2 + e_0, rho_4 = e_0     (two-tap phase: g0 = w[2], g1 = w[0])
//   (jb.transposed = 0, jb.flip = 1 gives the forms of the ADJOINT, a stride-2 conv k3 p1 with W[cout][cin][3][3]: g0 = w[0], g1 = w[2].
//    A kernel for it was built and measured in round 4 - 7.55 vs 7.80 ms per slide for the fused stride-2 layers, no gain on the slide,
//    3.6e-5 on the logits - and not kept: DESIGN.md 3.1c.)
// packed[chunk][product a * 5 + b][ntile_total][lane][j]: cout = cout_off + ntile*16 + (lane & 15), cin = chunk*16 + 4*(lane>>4) + j.
__device__ __forceinline__ void pack_convt_wino_elem(const PackJob& jb, int64_t e) {
    const int cout = jb.cout, cin = jb.cin;
    const int j = e & 3, lane = (e >> 2) & 63;
    int64_t r = e >> 8;
    const int ntl = (int)(r % (cout / 16));
    r /= (cout / 16);
    const int prod = (int)(r % 25);
    const int chunk = (int)(r / 25);
    const int co = ntl * 16 + (lane & 15), ci = chunk * 16 + 4 * (lane >> 4) + j;
    const float* src = jb.transposed ? jb.src + ((size_t)ci * cout + co) * 9 : jb.src + ((size_t)co * cin + ci) * 9;
    const int a = prod / 5, b = prod - a * 5;
    const int k_first = jb.flip ? 0 : 2, k_second = jb.flip ? 2 : 0;        // tap of g0 / of g1 on the two-tap phase
    auto rho = [&](int f, double (&o)[3]) {
        o[0] = o[1] = o[2] = 0.0;
        if (f == 0 || f == 1) o[1] = 1.0;
        if (f == 2 || f == 3) o[k_first] = 1.0;
        if (f == 3 || f == 4) o[k_second] = 1.0;
    };
    double ra[3], rb[3];
    rho(a, ra);
    rho(b, rb);
    double u = 0.0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) u += ra[ky] * rb[kx] * (double)src[ky * 3 + kx];
    reinterpret_cast<float*>(jb.dst)[(((size_t)chunk * 25 + prod) * jb.ntile_total + jb.cout_off / 16 + ntl) * 256 + lane * 4 + j] = (float)u;
}

}  // namespace cs
