// Internal helpers shared by the gfx950 kernels and the engine.  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/cellseg_hip.h"

namespace cs {

void set_error(const char* fmt, ...);

#define CS_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            cs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return CS_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define CS_REQUIRE(cond, ...)                    \
    do {                                         \
        if (!(cond)) {                           \
            cs::set_error(__VA_ARGS__);          \
            return CS_ERR_INVALID;               \
        }                                        \
    } while (0)

// ---- profiling (HIP events around launches, enabled by cs_prof_enable) ------------------------
enum KernelId {
    K_CONV3_S1 = 0, K_CONV3_S2, K_CONV1, K_CONV_FIRST, K_CONVT, K_CONVT_TOP, K_CONV_LAST,
    K_NORM_FINALIZE, K_NORM_ACT_ADD, K_SWI_FINALIZE, K_PACK, K_CCL_INIT, K_CCL_MERGE, K_CCL_COMPRESS,
    K_CCL_SIZES, K_CCL_FILTER, K_CCL_COUNT, K_CCL_SCAN, K_CCL_RANK, K_CCL_RELABEL, K_MEMSET, K_NORM_BWD, K_COLSUM,
    K_WGRAD, K_TOP_BWD, K_PRE, K_METRIC, K_UNPACK, K_CONV3_WINO, K_LOSS, K_CONV3_WINO4, K_LABEL_DEFLATE,
    K_CONV3_S1_FP8, K_CONV3_S2_FP8, K_CONV1_FP8, K_CONVT_FP8,      // the e4m3 launches of the fp8 mode (their own matrix peak)
    K_CCL_SEAM,                                                    // edge records + seam union-find of the strip-local labelling
    K_NUM_IDS
};
const char* kernel_name(int id);
struct ProfScope {   // records start event on construction and stop event on destruction when profiling is on
    ProfScope(int id, hipStream_t s, double flops, double bytes);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};

// A dense NHWC activation view: n images of h x w pixels, `pitch` floats between pixels,
// channels [0, c) valid starting at `ptr` (sub-ranges of a wider tensor are just pointer offsets).
struct View {
    float* ptr = nullptr;
    int c = 0, pitch = 0;
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// bf16 <-> fp32 on 4 packed values (activations of the bf16 datapath live in HBM as bf16, round-to-nearest-even on the way out)
typedef float cs_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cs_f32x4 bf16x4_to_f32(uint2 v) {
    cs_f32x4 r;
    r.x = __builtin_bit_cast(float, v.x << 16);
    r.y = __builtin_bit_cast(float, v.x & 0xffff0000u);
    r.z = __builtin_bit_cast(float, v.y << 16);
    r.w = __builtin_bit_cast(float, v.y & 0xffff0000u);
    return r;
}
__device__ __forceinline__ unsigned f32_to_bf16_bits(float f) {
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;            // RNE (NaN payloads are not expected on this path)
}
// 8 floats -> 8 e4m3 bytes (v_cvt_pk_fp8_f32; saturating when the kernel has set MODE.FP16_OVFL)
__device__ __forceinline__ uint2 f32x8_to_fp8(const float (&v)[8]) {
    unsigned lo = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    lo = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], (int)lo, true);
    unsigned hi = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
    hi = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], (int)hi, true);
    return make_uint2(lo, hi);
}
__device__ __forceinline__ uint2 f32x4_to_bf16(cs_f32x4 v) {
    return make_uint2(f32_to_bf16_bits(v.x) | (f32_to_bf16_bits(v.y) << 16), f32_to_bf16_bits(v.z) | (f32_to_bf16_bits(v.w) << 16));
}

}  // namespace cs
