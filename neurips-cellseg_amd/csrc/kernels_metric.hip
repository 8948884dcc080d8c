// gfx950 kernels for the F1 @ IoU harness: the O(H*W) parts of
// /root/reference/baseline/compute_metric.py run on the device, the small matching problem stays on the host.
//   _label_overlap          :39-70   -> label_overlap_kernel   (pixel-pair histogram, integer atomics)
//   remove_boundary_cells   :124-133 -> border flags + order-preserving relabel (prefix scan over label ids)
// Integer work: results are bit-identical to the numpy/numba reference.
#include "kernels.h"

namespace cs {

namespace {

__global__ __launch_bounds__(256) void label_max_kernel(const int32_t* a, const int32_t* b, int64_t n, int32_t* out2) {
    int32_t ma = 0, mb = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        ma = max(ma, a[i]);
        if (b) mb = max(mb, b[i]);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        ma = max(ma, __shfl_xor(ma, off));
        mb = max(mb, __shfl_xor(mb, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(out2, ma);
        atomicMax(out2 + 1, mb);
    }
}

__global__ __launch_bounds__(256) void label_overlap_kernel(const int32_t* x, const int32_t* y, int64_t n, int32_t ny1,
                                                            unsigned int* overlap) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&overlap[(size_t)x[i] * ny1 + y[i]], 1u);
}

// flag[l] bit0: label l present, bit1: label l touches the `border`-pixel frame
__global__ __launch_bounds__(256) void border_flags_kernel(const int32_t* lab, int h, int w, int border, unsigned char* flag) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int32_t l = lab[(size_t)y * w + x];
    if (l <= 0) return;
    const bool on_border = y < border || y >= h - border || x < border || x >= w - border;
    if (on_border) flag[l] = 3;                 // plain byte stores of monotone values: 3 wins over 1 in a later pass
    else if (flag[l] == 0) flag[l] = 1;
}
__global__ __launch_bounds__(256) void border_fix_kernel(const int32_t* lab, int h, int w, int border, unsigned char* flag) {
    // second pass: re-assert the border bit (a racing interior store of 1 may have overwritten a 3)
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const bool on_border = y < border || y >= h - border || x < border || x >= w - border;
    if (!on_border) return;
    const int32_t l = lab[(size_t)y * w + x];
    if (l > 0) flag[l] = 3;
}

// exclusive scan over label ids of keep = (flag == 1), one block (label counts are small next to pixel counts)
__global__ __launch_bounds__(1024) void relabel_scan_kernel(const unsigned char* flag, int32_t n_labels, int32_t* new_id, int32_t* n_out) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int base = 0; base <= n_labels; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i >= 1 && i <= n_labels && flag[i] == 1) ? 1 : 0;
        int inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        int wave_off = 0;
        for (int k = 0; k < wv; ++k) wave_off += wave_tot[k];
        const int carry = carry_s;
        if (i <= n_labels) new_id[i] = v ? carry + wave_off + inc : 0;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0 && n_out) *n_out = carry_s;
}

__global__ __launch_bounds__(256) void relabel_apply_kernel(const int32_t* lab, int64_t n, const int32_t* new_id, int32_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t l = lab[i];
    out[i] = l > 0 ? new_id[l] : 0;
}

}  // namespace

int label_max(hipStream_t s, const int32_t* a, const int32_t* b, int64_t n, int32_t* out2) {
    CS_HIP(hipMemsetAsync(out2, 0, 8, s));
    if (n == 0) return CS_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    ProfScope prof(K_METRIC, s, 0, 8.0 * n);
    hipLaunchKernelGGL(label_max_kernel, dim3(blocks), dim3(256), 0, s, a, b, n, out2);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int label_overlap(hipStream_t s, const int32_t* x, const int32_t* y, int64_t n, int32_t nx1, int32_t ny1, unsigned int* overlap) {
    CS_HIP(hipMemsetAsync(overlap, 0, (size_t)nx1 * ny1 * sizeof(unsigned int), s));
    if (n == 0) return CS_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    ProfScope prof(K_METRIC, s, 0, 12.0 * n);
    hipLaunchKernelGGL(label_overlap_kernel, dim3(blocks), dim3(256), 0, s, x, y, n, ny1, overlap);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

size_t remove_boundary_workspace_bytes(int32_t n_labels) { return ((size_t)n_labels + 1 + 255) / 256 * 256 + ((size_t)n_labels + 1) * 4 + 512; }

int remove_boundary_cells(hipStream_t s, const int32_t* lab, int h, int w, int border, int32_t n_labels, int32_t* out, int32_t* n_out,
                          void* ws, size_t ws_bytes) {
    if (ws_bytes < remove_boundary_workspace_bytes(n_labels)) {
        set_error("remove_boundary_cells: workspace too small");
        return CS_ERR_WORKSPACE;
    }
    unsigned char* flag = (unsigned char*)ws;
    int32_t* new_id = (int32_t*)((char*)ws + ((size_t)n_labels + 1 + 255) / 256 * 256);
    CS_HIP(hipMemsetAsync(flag, 0, (size_t)n_labels + 1, s));
    const int64_t n = (int64_t)h * w;
    ProfScope prof(K_METRIC, s, 0, 16.0 * n);
    if (n) {
        dim3 grid((unsigned)ceil_div(w, 256), (unsigned)h);
        hipLaunchKernelGGL(border_flags_kernel, grid, dim3(256), 0, s, lab, h, w, border, flag);
        hipLaunchKernelGGL(border_fix_kernel, grid, dim3(256), 0, s, lab, h, w, border, flag);
    }
    hipLaunchKernelGGL(relabel_scan_kernel, dim3(1), dim3(1024), 0, s, flag, n_labels, new_id, n_out);
    if (n) hipLaunchKernelGGL(relabel_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, lab, n, new_id, out);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
