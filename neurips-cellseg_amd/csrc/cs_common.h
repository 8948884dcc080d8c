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
    K_WGRAD, K_TOP_BWD, K_PRE, K_METRIC, K_NUM_IDS
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

}  // namespace cs
