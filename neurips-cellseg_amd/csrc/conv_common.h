// Device-side helpers shared by the fp32 convolution kernels (kernels_conv.hip, kernels_conv_wino.hip).  Not part of the C-ABI.
#pragma once
#include "kernels.h"

namespace cs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SrcDev {
    const float* ptr;
    const float2* tab;
    const float* slope;
    int c, pitch, tab_c, tab_off;
    const float* ptr2;             // second tensor of a SUM source (ConvSrc::ptr2), nullptr otherwise
    const float2* tab2;
    const float* slope2;
};
__device__ __forceinline__ SrcDev second_of(const SrcDev& s) { return SrcDev{s.ptr2, s.tab2, s.slope2, s.c, s.pitch, s.tab_c, s.tab_off, nullptr, nullptr, nullptr}; }

struct ConvParams {
    SrcDev a, b;
    const float* wpack;
    const float* bias;
    float* out;
    double* stats;
    int pitch_out, cout, ntile_total;
    int n, hin, win, hout, wout, tiles_x, tiles_y;
    // backward-pass extras (data gradients): out = conv + bias + addend; channels >= split_nt*16 go to out2
    const float* addend;
    const float* addend2;
    float* out2;
    int pitch_add, pitch_add2, pitch_out2, split_nt;
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

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// PReLU without a compare / select pair (v_max, v_min, v_fma: no VCC round trip between dependent VALU instructions, which
// matters on the fp32 path where every VALU cycle is taken from the MFMA stream); bit-identical to v >= 0 ? v : v * slope
__device__ __forceinline__ float prelu(float v, float slope) { return __builtin_fmaf(slope, __builtin_fminf(v, 0.f), __builtin_fmaxf(v, 0.f)); }

// lazy norm + activation on 4 consecutive channels.  Branch-free on purpose: a conditional load makes hipcc wait
// vmcnt(0) at the join, which would serialise the table fetch in front of the chunk's input loads.
struct Lazy4 {
    f32x4 t0, t1;      // (scale0, shift0, scale1, shift1), (scale2, shift2, scale3, shift3)
    float slope;
    bool on;
    __device__ __forceinline__ void load(const SrcDev& s, int img, int ch, const float* dummy) {
        on = s.tab != nullptr;
        const f32x4* t = on ? reinterpret_cast<const f32x4*>(s.tab + (size_t)img * s.tab_c + s.tab_off + ch)
                            : reinterpret_cast<const f32x4*>(dummy);
        t0 = t[0];
        t1 = t[1];
        const float* sp = (on && s.slope) ? s.slope : dummy;
        slope = *sp;
        if (!(on && s.slope)) slope = 0.f;
    }
    __device__ __forceinline__ f32x4 apply(f32x4 v) const {
        f32x4 r;
        r.x = prelu(v.x * t0.x + t0.y, slope);
        r.y = prelu(v.y * t0.z + t0.w, slope);
        r.z = prelu(v.z * t1.x + t1.y, slope);
        r.w = prelu(v.w * t1.z + t1.w, slope);
        return on ? r : v;
    }
    // the same for a caller that KNOWS the table is present (no select)
    __device__ __forceinline__ f32x4 apply_on(f32x4 v) const {
        f32x4 r;
        r.x = prelu(v.x * t0.x + t0.y, slope);
        r.y = prelu(v.y * t0.z + t0.w, slope);
        r.z = prelu(v.z * t1.x + t1.y, slope);
        r.w = prelu(v.w * t1.z + t1.w, slope);
        return r;
    }
};

// fp64 sum / sum-of-squares of one output channel held by the 4 lane groups of a wave -> atomics by group 0
__device__ __forceinline__ void stats_commit(double s, double q, double* dst, int lane) {
    s += __shfl_xor(s, 16);
    q += __shfl_xor(q, 16);
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    if (lane < 16) {
        atomicAdd(dst, s);
        atomicAdd(dst + 1, q);
    }
}

// destination of a block's channel tiles: (out, addend) below the split, (out2, addend2) at or above it
struct OutSel {
    float* out;
    const float* add;
    int pitch, pitch_add, ch0;
};
__device__ __forceinline__ OutSel select_out(const ConvParams& p, int nt0) {
    if (p.out2 && nt0 >= p.split_nt) return OutSel{p.out2, p.addend2, p.pitch_out2, p.pitch_add2, (nt0 - p.split_nt) * 16};
    return OutSel{p.out, p.addend, p.pitch_out, p.pitch_add, nt0 * 16};
}

}  // namespace cs
