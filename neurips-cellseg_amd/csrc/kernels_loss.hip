// gfx950 (MI355X): the training loss of the reference, DiceCE on softmax probabilities (row L of SURVEY.md 8(a)).
//
// Replaces monai.losses.DiceCELoss(softmax=True)(outputs, one_hot(labels)) and its autograd backward
// (baseline/model_training_3class.py:234 and :256-260).  MONAI semantics restated:
//   p = softmax(logits, dim 1);   dice_nc = 1 - (2 sum(p t) + nr) / (sum(t) + sum(p) + dr)   per image and class, mean over n * c;
//   ce = mean over pixels of -log p[argmax t];   loss = lambda_dice * mean(dice) + lambda_ce * ce.
// Torch needed ~15 element-wise / reduction launches over the [N,3,H,W] tensors for forward + backward (0.40 ms of a 6.95 ms
// training step at 32x3x256x256); here it is one pass that reads logits and target once (per-(image, class) sums in fp64:
// thread -> wave shuffles -> block LDS -> one atomic per block), a one-block finalize, and one pass that rebuilds p and writes
// d loss / d logits in the logits' own memory layout (the engine's NHWC), ~30 us together.
#include "kernels.h"

namespace cs {

namespace {
constexpr int LOSS_MAXC = 4;
constexpr int LOSS_BLOCKS_PER_IMAGE = 64;

struct LossArgs {
    const float* logits;
    int64_t sn, sc, sh, sw;
    const float* target;
    int64_t tn, tc, th, tw;
    int n, c, h, w;
    float smooth_nr, smooth_dr, lambda_dice, lambda_ce;
    double* sums;          // [n][c][3] = (sum p t, sum p, sum t), then [n] = per-image sums of -log p[argmax t] (ONE word for the whole
                           // batch took 2048 serialised fp64 atomics per launch: 70 us of a 160 us loss)
};

// softmax over c <= 4 classes of one pixel; returns the index of the largest target (first maximum, as torch.argmax) and
// log(sum exp(z - max)) + max through `lse`
template <int C>
__device__ __forceinline__ int pixel_softmax(const LossArgs& a, int img, int y, int x, float (&p)[C], float (&t)[C], float (&z)[C], float* lse) {
    const float* zp = a.logits + (int64_t)img * a.sn + (int64_t)y * a.sh + (int64_t)x * a.sw;
    const float* tp = a.target + (int64_t)img * a.tn + (int64_t)y * a.th + (int64_t)x * a.tw;
    float m = -3.4e38f;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        z[k] = zp[k * a.sc];
        t[k] = tp[k * a.tc];
        m = fmaxf(m, z[k]);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        p[k] = __expf(z[k] - m);
        s += p[k];
    }
    const float inv = 1.f / s;
    int best = 0;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        p[k] *= inv;
        if (t[k] > t[best]) best = k;
    }
    *lse = m + __logf(s);
    return best;
}

template <int C>
__global__ __launch_bounds__(256) void dice_ce_sums_kernel(LossArgs a) {
    __shared__ double red[4][3 * C + 1];
    const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int npix = a.h * a.w;
    double acc[3 * C + 1];
#pragma unroll
    for (int k = 0; k < 3 * C + 1; ++k) acc[k] = 0.0;
    for (int i = blockIdx.x * 256 + tid; i < npix; i += gridDim.x * 256) {
        const int y = i / a.w, x = i - y * a.w;
        float p[C], t[C], z[C], lse;
        const int best = pixel_softmax<C>(a, img, y, x, p, t, z, &lse);
#pragma unroll
        for (int k = 0; k < C; ++k) {
            acc[3 * k] += (double)(p[k] * t[k]);
            acc[3 * k + 1] += (double)p[k];
            acc[3 * k + 2] += (double)t[k];
        }
        float zb = z[0];
#pragma unroll
        for (int k = 1; k < C; ++k) zb = best == k ? z[k] : zb;
        acc[3 * C] += (double)(lse - zb);
    }
#pragma unroll
    for (int k = 0; k < 3 * C + 1; ++k) {
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) acc[k] += __shfl_xor(acc[k], sh);
        if (lane == 0) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (tid < 3 * C + 1) {
        const double v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        if (tid < 3 * C) atomicAdd(a.sums + ((size_t)img * C) * 3 + tid, v);
        else atomicAdd(a.sums + (size_t)a.n * C * 3 + img, v);
    }
}

__global__ void dice_ce_finalize_kernel(LossArgs a, float* loss_out) {
    __shared__ double part[256];
    const int tid = threadIdx.x, items = a.n * a.c;
    double d = 0.0;
    for (int i = tid; i < items; i += 256) {
        const double inter = a.sums[(size_t)i * 3], ps = a.sums[(size_t)i * 3 + 1], ts = a.sums[(size_t)i * 3 + 2];
        d += 1.0 - (2.0 * inter + (double)a.smooth_nr) / (ts + ps + (double)a.smooth_dr);
    }
    part[tid] = d;
    __syncthreads();
    if (tid == 0) {
        double dice = 0.0;
        for (int i = 0; i < 256; ++i) dice += part[i];
        dice /= (double)items;
        double ce = 0.0;
        for (int i = 0; i < a.n; ++i) ce += a.sums[(size_t)items * 3 + i];
        ce /= (double)a.n * a.h * a.w;
        *loss_out = (float)((double)a.lambda_dice * dice + (double)a.lambda_ce * ce);
    }
}

// d loss / d logits:  dz_k = p_k (g_k - sum_c g_c p_c) + lambda_ce (p_k - y_k) / (n h w),
// g_c = lambda_dice / (n c) * d dice_nc / d p_c = -lambda_dice / (n c) * (2 t_c (D + dr) - (2 I + nr)) / (D + dr)^2,  D = sum t + sum p
template <int C>
__global__ __launch_bounds__(256) void dice_ce_backward_kernel(LossArgs a, const float* grad_loss, float* grad) {
    const int img = blockIdx.y, tid = threadIdx.x;
    const float gl = grad_loss ? *grad_loss : 1.f;
    float ga[C], gb[C];                                       // g_c = ga * t_c + gb
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const double inter = a.sums[((size_t)img * C + k) * 3], dd = a.sums[((size_t)img * C + k) * 3 + 1] + a.sums[((size_t)img * C + k) * 3 + 2] + (double)a.smooth_dr;
        const double coef = (double)gl * (double)a.lambda_dice / ((double)a.n * C);
        ga[k] = (float)(-coef * 2.0 / dd);
        gb[k] = (float)(coef * (2.0 * inter + (double)a.smooth_nr) / (dd * dd));
    }
    const float cce = gl * a.lambda_ce / ((float)a.n * (float)a.h * (float)a.w);
    const int npix = a.h * a.w;
    for (int i = blockIdx.x * 256 + tid; i < npix; i += gridDim.x * 256) {
        const int y = i / a.w, x = i - y * a.w;
        float p[C], t[C], z[C], lse;
        const int best = pixel_softmax<C>(a, img, y, x, p, t, z, &lse);
        float g[C], dot = 0.f;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            g[k] = ga[k] * t[k] + gb[k];
            dot += g[k] * p[k];
        }
        float* gp = grad + (int64_t)img * a.sn + (int64_t)y * a.sh + (int64_t)x * a.sw;
#pragma unroll
        for (int k = 0; k < C; ++k) gp[k * a.sc] = p[k] * (g[k] - dot) + cce * (p[k] - (best == k ? 1.f : 0.f));
    }
}
}  // namespace

size_t dice_ce_workspace_bytes(int n, int c) { return ((size_t)n * c * 3 + n) * sizeof(double); }

static int fill_args(LossArgs& a, const float* logits, const int64_t* ls, const float* target, const int64_t* ts, int n, int c, int h, int w,
                     float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce, void* ws, size_t ws_bytes) {
    CS_REQUIRE(logits && target && ws, "dice_ce: null argument");
    CS_REQUIRE(n > 0 && h > 0 && w > 0 && c >= 2 && c <= LOSS_MAXC, "dice_ce: n %d, c %d (2..4), h %d, w %d", n, c, h, w);
    CS_REQUIRE(ws_bytes >= dice_ce_workspace_bytes(n, c), "dice_ce: workspace %zu < %zu bytes", ws_bytes, dice_ce_workspace_bytes(n, c));
    a.logits = logits; a.sn = ls[0]; a.sc = ls[1]; a.sh = ls[2]; a.sw = ls[3];
    a.target = target; a.tn = ts[0]; a.tc = ts[1]; a.th = ts[2]; a.tw = ts[3];
    a.n = n; a.c = c; a.h = h; a.w = w;
    a.smooth_nr = smooth_nr; a.smooth_dr = smooth_dr; a.lambda_dice = lambda_dice; a.lambda_ce = lambda_ce;
    a.sums = reinterpret_cast<double*>(ws);
    return CS_OK;
}

int dice_ce_forward(hipStream_t s, const float* logits, const int64_t* ls, const float* target, const int64_t* ts, int n, int c, int h, int w,
                    float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce, float* loss_out, void* ws, size_t ws_bytes) {
    LossArgs a{};
    if (int rc = fill_args(a, logits, ls, target, ts, n, c, h, w, smooth_nr, smooth_dr, lambda_dice, lambda_ce, ws, ws_bytes)) return rc;
    CS_REQUIRE(loss_out, "dice_ce: null loss pointer");
    ProfScope prof(K_LOSS, s, 0, 8.0 * n * c * (double)h * w);
    if (int zrc = launch_zero(s, ws, dice_ce_workspace_bytes(n, c))) return zrc;
    const int bx = ceil_div(h * w, 256) < LOSS_BLOCKS_PER_IMAGE ? ceil_div(h * w, 256) : LOSS_BLOCKS_PER_IMAGE;
    const dim3 grid((unsigned)bx, (unsigned)n);
    switch (c) {
        case 2: hipLaunchKernelGGL(dice_ce_sums_kernel<2>, grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(dice_ce_sums_kernel<3>, grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(dice_ce_sums_kernel<4>, grid, dim3(256), 0, s, a); break;
    }
    hipLaunchKernelGGL(dice_ce_finalize_kernel, dim3(1), dim3(256), 0, s, a, loss_out);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int dice_ce_backward(hipStream_t s, const float* logits, const int64_t* ls, const float* target, const int64_t* ts, int n, int c, int h, int w,
                     float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce, const float* grad_loss, const void* ws,
                     size_t ws_bytes, float* grad_logits) {
    LossArgs a{};
    if (int rc = fill_args(a, logits, ls, target, ts, n, c, h, w, smooth_nr, smooth_dr, lambda_dice, lambda_ce, const_cast<void*>(ws), ws_bytes))
        return rc;
    CS_REQUIRE(grad_logits, "dice_ce: null gradient pointer");
    ProfScope prof(K_LOSS, s, 0, 12.0 * n * c * (double)h * w);
    const int bx = ceil_div(h * w, 256) < LOSS_BLOCKS_PER_IMAGE ? ceil_div(h * w, 256) : LOSS_BLOCKS_PER_IMAGE;
    const dim3 grid((unsigned)bx, (unsigned)n);
    switch (c) {
        case 2: hipLaunchKernelGGL(dice_ce_backward_kernel<2>, grid, dim3(256), 0, s, a, grad_loss, grad_logits); break;
        case 3: hipLaunchKernelGGL(dice_ce_backward_kernel<3>, grid, dim3(256), 0, s, a, grad_loss, grad_logits); break;
        default: hipLaunchKernelGGL(dice_ce_backward_kernel<4>, grid, dim3(256), 0, s, a, grad_loss, grad_logits); break;
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs

// ------------------------------------------------------------------------------------------------
// AdamW (row O): `optimizer = torch.optim.AdamW(model.parameters(), lr)` ... `optimizer.step()`
// (baseline/model_training_3class.py:237 and :262), torch's update rule (amsgrad off, maximize off):
//   p *= 1 - lr wd;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// for ALL parameter tensors in one launch: the gradients already sit in one flat bucket (cs_unet_backward), the two moment
// buffers use the same offsets, and a small device table maps flat offsets to the parameter tensors torch owns.  torch's
// foreach implementation needs a dozen multi-tensor launches for the 63 tensors (0.18 ms of a 6.5 ms training step).
// ------------------------------------------------------------------------------------------------
namespace cs {
namespace {
// One thread = one aligned quad of the flat bucket (every tensor starts on a multiple of 4 floats, so a quad never straddles two
// tensors; the last quad of a tensor may be partial).  The offset table sits in LDS for the binary search.  (The first form searched
// the table in global memory once per ELEMENT and moved 4 bytes per lane: 81 us for the 6.5 M parameters of the 32-base net, 1.3 TB/s.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct AdamwCoef { float decay, b1, b2, eps, step_size, inv_bc2_sqrt; };
__device__ __forceinline__ void adamw_quad(const int64_t* table, int n_tensors, int64_t total, const float* grad, float* exp_avg,
                                           float* exp_avg_sq, const AdamwCoef& k) {
    __shared__ int64_t off_s[256];                 // flat offsets of the tensors (n_tensors <= 256: checked by the launchers)
    for (int i = threadIdx.x; i < n_tensors; i += 256) off_s[i] = table[i * 3 + 1];
    __syncthreads();
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= total) return;
    int lo = 0, hi = n_tensors - 1;                            // last tensor whose flat offset is <= e
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off_s[mid] <= e) lo = mid;
        else hi = mid - 1;
    }
    const int64_t local = e - off_s[lo], numel = table[lo * 3 + 2];
    if (local < 0 || local >= numel) return;                   // alignment gap between two tensors
    float* p = reinterpret_cast<float*>(table[lo * 3]) + local;
    const int cnt = numel - local >= 4 ? 4 : (int)(numel - local);
    const bool vec = cnt == 4 && e + 4 <= total && ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
    float g[4], m[4], v[4], w[4];
    if (vec) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(grad + e), m4 = *reinterpret_cast<const f32x4*>(exp_avg + e),
                    v4 = *reinterpret_cast<const f32x4*>(exp_avg_sq + e), w4 = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) { g[j] = g4[j]; m[j] = m4[j]; v[j] = v4[j]; w[j] = w4[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < cnt) { g[j] = grad[e + j]; m[j] = exp_avg[e + j]; v[j] = exp_avg_sq[e + j]; w[j] = p[j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = m[j] + (g[j] - m[j]) * (1.f - k.b1);
        v[j] = v[j] * k.b2 + (1.f - k.b2) * g[j] * g[j];
        const float denom = sqrtf(v[j]) * k.inv_bc2_sqrt + k.eps;
        w[j] = w[j] * k.decay - k.step_size * (m[j] / denom);
    }
    if (vec) {
        *reinterpret_cast<f32x4*>(exp_avg + e) = f32x4{m[0], m[1], m[2], m[3]};
        *reinterpret_cast<f32x4*>(exp_avg_sq + e) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p) = f32x4{w[0], w[1], w[2], w[3]};
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < cnt) { exp_avg[e + j] = m[j]; exp_avg_sq[e + j] = v[j]; p[j] = w[j]; }
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(const int64_t* table, int n_tensors, int64_t total, const float* grad, float* exp_avg,
                                                    float* exp_avg_sq, float decay, float b1, float b2, float eps, float step_size, float inv_bc2_sqrt) {
    adamw_quad(table, n_tensors, total, grad, exp_avg, exp_avg_sq, AdamwCoef{decay, b1, b2, eps, step_size, inv_bc2_sqrt});
}
// device-resident hyperparameters (captured training step: nothing of the update may live in kernel arguments that change per step)
//   hyper[0..4] = lr, beta1, beta2, eps, weight_decay (written by the host when they change); hyper[8..10] = decay, step size, 1 / sqrt(bc2)
//   of the step being applied (written by the tick kernel); *step = number of steps applied so far
__global__ void adamw_tick_kernel(float* hyper, int* step) {
    const int t = *step + 1;
    *step = t;
    const double lr = hyper[0], b1 = hyper[1], b2 = hyper[2], wd = hyper[4];
    const double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
    hyper[8] = (float)(1.0 - lr * wd);
    hyper[9] = (float)(lr / bc1);
    hyper[10] = (float)(1.0 / sqrt(bc2));
}
__global__ __launch_bounds__(256) void adamw_dev_kernel(const int64_t* table, int n_tensors, int64_t total, const float* grad, float* exp_avg,
                                                        float* exp_avg_sq, const float* hyper) {
    adamw_quad(table, n_tensors, total, grad, exp_avg, exp_avg_sq, AdamwCoef{hyper[8], hyper[1], hyper[2], hyper[3], hyper[9], hyper[10]});
}
}  // namespace

int adamw_step_dev(hipStream_t s, const int64_t* table, int n_tensors, int64_t total, const float* grad, float* exp_avg, float* exp_avg_sq,
                   float* hyper, int* step) {
    CS_REQUIRE(table && grad && exp_avg && exp_avg_sq && hyper && step && n_tensors > 0 && n_tensors <= 256 && total > 0, "adamw(dev): bad argument");
    ProfScope prof(K_LOSS, s, 0, 28.0 * (double)total);
    hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, s, hyper, step);
    hipLaunchKernelGGL(adamw_dev_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, s, table, n_tensors, total, grad, exp_avg, exp_avg_sq,
                       hyper);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int adamw_step(hipStream_t s, const int64_t* table, int n_tensors, int64_t total, const float* grad, float* exp_avg, float* exp_avg_sq,
               float lr, float beta1, float beta2, float eps, float weight_decay, int step) {
    CS_REQUIRE(table && grad && exp_avg && exp_avg_sq && n_tensors > 0 && n_tensors <= 256 && total > 0 && step >= 1, "adamw: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    ProfScope prof(K_LOSS, s, 0, 28.0 * (double)total);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, s, table, n_tensors, total, grad, exp_avg, exp_avg_sq,
                       (float)(1.0 - (double)lr * weight_decay), beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)));
    CS_HIP(hipGetLastError());
    return CS_OK;
}
}  // namespace cs
