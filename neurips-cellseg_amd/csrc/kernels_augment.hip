// gfx950 training augmentations on the device (the "next" row that was CPU-side until round 3).
//
// Replaces the per-sample work of the reference's MONAI transform chain (baseline/model_training_3class.py:119-162):
//   SpatialPadd -> RandSpatialCropd(roi 256) -> RandAxisFlipd -> RandRotate90d -> RandGaussianNoised(std <= 0.1)
//   -> RandAdjustContrastd(gamma in (1, 2)) -> RandGaussianSmoothd(sigma_x in (1, 2)) -> RandHistogramShiftd(3 control points)
//   -> RandZoomd(0.8 ... 1.5, area / nearest, keep_size, edge padding)
// for a whole BATCH at once, from a dataset that lives in HBM (cellseg_amd.gpu_data.DeviceCellDataset): the host draws one small
// parameter record per sample (cs_aug_params: which transforms fire, crop origin, flip axis, rotation count, noise seed, gamma, the
// Gaussian taps, the control points, the zoom factor) and the kernels below do everything that touches pixels.  The host pipeline
// (cellseg_amd/data.py, DataLoader workers) manages ~130 samples/s per worker; the training step consumes 5 800 images/s.
//
// HBM-bound byte / float work, a handful of tiny launches per batch (grid.y = sample; a stage that does not fire for a sample returns
// at once).  Deterministic stages follow the host implementations operation by operation (cellseg_amd/data.py, themselves pinned to
// the torch / numpy / scipy primitives MONAI calls: tests/golden/augment_golden.npz): fp64 accumulation where scipy / numpy use it,
// torch's index rules for the area / nearest resampling.  The Gaussian noise comes from a counter-based generator (splitmix64 +
// Box-Muller) seeded per sample: same distribution as numpy's, not the same stream.
#include "kernels.h"

namespace cs {

namespace {

typedef cs_aug_params AugParams;

// crop pixel (y, x) of the flipped / rotated output <- pixel of the crop window (np.flip(axis) then np.rot90(k, axes=(1, 2)))
__device__ __forceinline__ void unrotate(const AugParams& p, int s, int y, int x, int* cy, int* cx) {
    int by, bx;
    switch (p.rot_k & 3) {
        case 1: by = x; bx = s - 1 - y; break;              // rot90: out[y][x] = in[x][s - 1 - y]
        case 2: by = s - 1 - y; bx = s - 1 - x; break;
        case 3: by = s - 1 - x; bx = y; break;
        default: by = y; bx = x; break;
    }
    if (p.flip_axis == 1) by = s - 1 - by;
    else if (p.flip_axis == 2) bx = s - 1 - bx;
    *cy = by;
    *cx = bx;
}

__device__ __forceinline__ float gauss_noise(unsigned seed, unsigned long long idx) {
    unsigned long long z = ((unsigned long long)seed << 32) ^ (idx * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u1 = ((float)(unsigned)(z >> 40) + 0.5f) * (1.f / 16777216.f);          // (0, 1)
    const float u2 = (float)(unsigned)((z >> 16) & 0xFFFFFFu) * (1.f / 16777216.f);
    return sqrtf(-2.f * __logf(u1)) * __cosf(6.2831853f * u2);
}

// stage 1: pad / crop / flip / rotate (+ noise): img [n][c][s][s], lab [n][s][s]
__global__ __launch_bounds__(256) void aug_gather_kernel(const float* img_arena, const unsigned char* lab_arena, const AugParams* params, int c,
                                                         int s, float* img, unsigned char* lab) {
    const AugParams& p = params[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= s * s) return;
    const int y = i / s, x = i - y * s;
    int cy, cx;
    unrotate(p, s, y, x, &cy, &cx);
    const int sy = p.crop_y + cy - p.pad_top, sx = p.crop_x + cx - p.pad_left;           // source pixel (zero padding outside)
    const bool in = sy >= 0 && sy < p.src_h && sx >= 0 && sx < p.src_w;
    const size_t so = (size_t)sy * p.src_w + sx, plane = (size_t)p.src_h * p.src_w;
    float* o = img + (size_t)blockIdx.y * c * s * s + i;
    for (int ch = 0; ch < c; ++ch) {
        float v = in ? img_arena[p.img_off + ch * plane + so] : 0.f;
        if (p.noise_std >= 0.f) v += p.noise_std * gauss_noise(p.noise_seed, (unsigned long long)ch * s * s + i);
        o[(size_t)ch * s * s] = v;
    }
    lab[(size_t)blockIdx.y * s * s + i] = in ? lab_arena[p.lab_off + so] : (unsigned char)0;
}

// min / max of a sample's image: one block per sample, result in stats[sample] = (min, max)
__global__ __launch_bounds__(1024) void aug_minmax_kernel(const float* img, const AugParams* params, int which, int count, float2* stats) {
    const AugParams& p = params[blockIdx.x];
    if (which == 0 ? !(p.gamma > 0.f) : p.hist_n < 2) return;
    __shared__ float lo_s[16], hi_s[16];
    const float* a = img + (size_t)blockIdx.x * count;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < count; i += 1024) {
        const float v = a[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        lo_s[threadIdx.x >> 6] = lo;
        hi_s[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            lo = fminf(lo, lo_s[w]);
            hi = fmaxf(hi, hi_s[w]);
        }
        stats[blockIdx.x] = make_float2(fminf(lo, lo_s[0]), fmaxf(hi, hi_s[0]));
    }
}

// stage 2: gamma contrast, in place:  ((x - lo) / (hi - lo + 1e-7)) ** gamma * (hi - lo) + lo   (float32, as numpy evaluates it)
__global__ __launch_bounds__(256) void aug_gamma_kernel(float* img, const AugParams* params, int count, const float2* stats) {
    const AugParams& p = params[blockIdx.y];
    if (!(p.gamma > 0.f)) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float lo = stats[blockIdx.y].x, hi = stats[blockIdx.y].y;
    const float range = hi - lo;
    float* a = img + (size_t)blockIdx.y * count;
    a[i] = powf((a[i] - lo) / (range + 1e-7f), p.gamma) * range + lo;
}

// stage 3: one axis of the separable Gaussian (scipy.ndimage.correlate1d, mode="constant": zero padding, fp64 accumulation)
template <int AXIS>   // 0: along y (sigma_y taps), 1: along x
__global__ __launch_bounds__(256) void aug_smooth_kernel(const float* src, float* dst, const AugParams* params, int c, int s) {
    const AugParams& p = params[blockIdx.y];
    const int nt_raw = AXIS == 0 ? p.taps_y_n : p.taps_x_n;
    const int nt = nt_raw > CS_AUG_MAX_TAPS ? CS_AUG_MAX_TAPS : (nt_raw < 0 ? 0 : nt_raw);   // (records come from the host: never index past the arrays)
    if (p.taps_y_n <= 0 && p.taps_x_n <= 0) return;                  // smoothing does not fire: the data stays where it is
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= c * s * s) return;
    const size_t base = (size_t)blockIdx.y * c * s * s;
    if (nt == 0) {                                                   // only the other axis is smoothed: pass through
        dst[base + i] = src[base + i];
        return;
    }
    const float* taps = AXIS == 0 ? p.taps_y : p.taps_x;
    const int tail = nt / 2;
    const int x = i % s, y = (i / s) % s;
    const int pos = AXIS == 0 ? y : x, stride = AXIS == 0 ? s : 1;
    const float* line = src + base + i - (size_t)pos * stride;
    double acc = 0.0;
    for (int j = 0; j < nt; ++j) {
        const int q = pos + j - tail;
        if (q >= 0 && q < s) acc += (double)taps[j] * (double)line[(size_t)q * stride];
    }
    dst[base + i] = (float)acc;
}

// stage 4: histogram shift, in place: np.interp(x, ref * (hi - lo) + lo, floating * (hi - lo) + lo) in fp64, ref = linspace(0, 1, n)
__global__ __launch_bounds__(256) void aug_hist_kernel(float* img, const AugParams* params, int count, const float2* stats) {
    const AugParams& p = params[blockIdx.y];
    if (p.hist_n < 2) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float lo = stats[blockIdx.y].x, hi = stats[blockIdx.y].y;
    const double range = (double)(hi - lo), lod = (double)lo;
    const int n = p.hist_n > 4 ? 4 : p.hist_n;
    float* a = img + (size_t)blockIdx.y * count;
    const double x = (double)a[i];
    double xp0 = lod, fp0 = (double)p.hist_y[0] * range + lod, r = fp0;
    bool done = x <= xp0;                                            // np.interp: left value at and below the first point
    for (int j = 1; j < n && !done; ++j) {
        const double xp1 = ((double)j / (double)(n - 1)) * range + lod, fp1 = (double)p.hist_y[j] * range + lod;
        if (x < xp1 || (j == n - 1 && x == xp1)) {
            r = x == xp1 ? fp1 : (fp1 - fp0) / (xp1 - xp0) * (x - xp0) + fp0;
            done = true;
        } else {
            xp0 = xp1;
            fp0 = fp1;
            r = fp1;                                                 // right value beyond the last point
        }
    }
    a[i] = (float)r;
}

// stage 5: zoom with the size kept (or a plain copy): torch's "area" (adaptive average) for the image, "nearest" for the label, then a
// centre crop (z > 1) or edge padding (z < 1) back to s x s
__global__ __launch_bounds__(256) void aug_zoom_kernel(const float* src, const unsigned char* lab_src, const AugParams* params, int c, int s,
                                                       float* out, unsigned char* lab_out) {
    const AugParams& p = params[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= s * s) return;
    const size_t ibase = (size_t)blockIdx.y * c * s * s, lbase = (size_t)blockIdx.y * s * s;
    if (!(p.zoom > 0.f) || p.zoom_h <= 0 || p.zoom_w <= 0) {
        for (int ch = 0; ch < c; ++ch) out[ibase + (size_t)ch * s * s + i] = src[ibase + (size_t)ch * s * s + i];
        lab_out[lbase + i] = lab_src[lbase + i];
        return;
    }
    const int y = i / s, x = i - y * s;
    const int oh = p.zoom_h, ow = p.zoom_w;                          // size of the resampled image: floor(s * z)
    // position inside the resampled image: centre crop (oh > s) or edge padding (oh < s: d // 2 before, the rest after)
    int zy = oh >= s ? y + (oh - s) / 2 : y - (s - oh) / 2;
    int zx = ow >= s ? x + (ow - s) / 2 : x - (s - ow) / 2;
    zy = zy < 0 ? 0 : (zy > oh - 1 ? oh - 1 : zy);
    zx = zx < 0 ? 0 : (zx > ow - 1 ? ow - 1 : zx);
    // adaptive average: rows [floor(zy * s / oh), ceil((zy + 1) * s / oh)), torch's float formulas
    int y0 = (int)floorf((float)(zy * s) / (float)oh), y1 = (int)ceilf((float)((zy + 1) * s) / (float)oh);
    int x0 = (int)floorf((float)(zx * s) / (float)ow), x1 = (int)ceilf((float)((zx + 1) * s) / (float)ow);
    y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0; y1 = y1 > s ? s : y1; x1 = x1 > s ? s : x1;
    // (floor / ceil of exact small-integer quotients: always inside [0, s]; the clamps only guard against a malformed record)
    const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
    for (int ch = 0; ch < c; ++ch) {
        const float* a = src + ibase + (size_t)ch * s * s;
        float sum = 0.f;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) sum += a[yy * s + xx];
        out[ibase + (size_t)ch * s * s + i] = sum * inv;
    }
    // nearest (torch's legacy rule): src = min(floor(dst * (float)in / out), in - 1)
    const float sy = (float)s / (float)oh, sx = (float)s / (float)ow;
    int ny = (int)floorf((float)zy * sy), nx = (int)floorf((float)zx * sx);
    ny = ny > s - 1 ? s - 1 : ny;
    nx = nx > s - 1 ? s - 1 : nx;
    lab_out[lbase + i] = lab_src[lbase + ny * s + nx];
}

size_t align256a(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

size_t augment_workspace_bytes(int n, int c, int s) {
    if (n <= 0 || c <= 0 || s <= 0) return 0;
    return 2 * align256a((size_t)n * c * s * s * 4) + align256a((size_t)n * s * s) + align256a((size_t)n * 8);
}

int augment_batch(hipStream_t st, const float* img_arena, const unsigned char* lab_arena, const cs_aug_params* params_dev, int n, int c, int s,
                  float* out_img, unsigned char* out_lab, void* workspace, size_t ws_bytes) {
    CS_REQUIRE(n > 0 && c > 0 && c <= 4 && s > 0 && s <= 4096, "augment: bad batch shape (%d samples, %d channels, size %d)", n, c, s);
    if (ws_bytes < augment_workspace_bytes(n, c, s)) {
        set_error("augment: workspace %zu < %zu bytes", ws_bytes, augment_workspace_bytes(n, c, s));
        return CS_ERR_WORKSPACE;
    }
    char* w = static_cast<char*>(workspace);
    float* a = reinterpret_cast<float*>(w);
    w += align256a((size_t)n * c * s * s * 4);
    float* b = reinterpret_cast<float*>(w);
    w += align256a((size_t)n * c * s * s * 4);
    unsigned char* lab = reinterpret_cast<unsigned char*>(w);
    w += align256a((size_t)n * s * s);
    float2* stats = reinterpret_cast<float2*>(w);
    const int count = c * s * s;
    ProfScope prof(K_PRE, st, 0, 4.0 * n * count * 6);
    const dim3 gpx((unsigned)ceil_div(s * s, 256), (unsigned)n), gel((unsigned)ceil_div(count, 256), (unsigned)n);
    hipLaunchKernelGGL(aug_gather_kernel, gpx, dim3(256), 0, st, img_arena, lab_arena, params_dev, c, s, a, lab);
    hipLaunchKernelGGL(aug_minmax_kernel, dim3((unsigned)n), dim3(1024), 0, st, a, params_dev, 0, count, stats);
    hipLaunchKernelGGL(aug_gamma_kernel, gel, dim3(256), 0, st, a, params_dev, count, stats);
    hipLaunchKernelGGL((aug_smooth_kernel<0>), gel, dim3(256), 0, st, a, b, params_dev, c, s);
    hipLaunchKernelGGL((aug_smooth_kernel<1>), gel, dim3(256), 0, st, b, a, params_dev, c, s);
    hipLaunchKernelGGL(aug_minmax_kernel, dim3((unsigned)n), dim3(1024), 0, st, a, params_dev, 1, count, stats);
    hipLaunchKernelGGL(aug_hist_kernel, gel, dim3(256), 0, st, a, params_dev, count, stats);
    hipLaunchKernelGGL(aug_zoom_kernel, gpx, dim3(256), 0, st, a, lab, params_dev, c, s, out_img, out_lab);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
