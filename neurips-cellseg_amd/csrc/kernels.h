// Internal launcher API of the gfx950 kernels (host side).  Not part of the C-ABI.
#pragma once
#include "cs_common.h"

namespace cs {

// One input source of a convolution: raw NHWC tensor + optional lazy norm/act (see Lazy).
struct ConvSrc {
    const float* ptr = nullptr;
    int c = 0;          // channels taken from this source (multiple of 16 for the MFMA kernels)
    int pitch = 0;      // floats between pixels
    const float2* tab = nullptr;   // lazy norm table [n][tab_c] (scale, shift) or nullptr (raw)
    const float* slope = nullptr;  // PReLU slope (device scalar); nullptr with tab set = ReLU
    int tab_c = 0;                 // channels per image in `tab` (>= c when this source is a channel sub-range)
    int tab_off = 0;               // channel offset of this source inside `tab`
    bool h16 = false;              // `ptr` really points at bf16 elements (the bf16 datapath keeps activations as bf16 in HBM);
                                   // pitch / c still count elements
    bool q8 = false;               // `ptr` points at e4m3 bytes: an ACTIVATED tensor materialised for the fp8 kernels (no table)
    // SUM source: the value is act(norm(ptr)) + act(norm(ptr2)), two raw tensors of the same shape and pitch with their own lazy
    // tables (tab_c / tab_off shared; tab2 == nullptr: the second tensor is added as it is) - a residual sum, consumed without
    // ever being written to HBM.  Only the `b` operand of launch_convt_mfma understands it (fp32 inference); every other
    // launcher refuses such a source.
    const float* ptr2 = nullptr;
    const float2* tab2 = nullptr;
    const float* slope2 = nullptr;
};

// ---- weight packing (PyTorch layout -> MFMA fragment order) -----------------------------------
// conv  W[cout][cin][k][k]  (transposed == false)   /   conv-transpose W[cin][cout][k][k] (transposed == true)
// packed[chunk][tap][ntile_total][64 lanes][4] with cout = ntile*16 + (lane & 15), cin = chunk*16 + 4*(lane>>4) + j.
// `cout_off`/`cout_total` let two convs (unit0 || residual) share one packed tensor; `cin_total` is the
// concatenated input width.
// `dgrad` (optional): the same tensor packed a second time for the DATA-GRADIENT launch, i.e. with the roles of the two channel
// axes swapped (a conv's data gradient contracts over its output channels): chunk = output-channel tile (+ dgrad_chunk_off
// tiles: the second half of a fused [unit0 || residual] pair), tile = input-channel tile (of cin/16), taps reversed when
// dgrad_flip (data gradient of a stride-1 conv).  One block reads a 16 x 16 x taps weight tile once (coalesced rows) and
// writes both layouts as whole 1 KB fragments.
struct PackCtx;
PackCtx* pack_ctx_create(int max_jobs);                    // allocates the job tables on the current device
void pack_ctx_destroy(PackCtx* pc);
int pack_conv_weights(PackCtx* pc, hipStream_t s, const float* w, bool transposed, int cout, int cin, int ksize,
                      float* packed, int cout_off, int cout_total, float* dgrad = nullptr, bool dgrad_flip = false,
                      int dgrad_chunk_off = 0, float* wino = nullptr, float* wino_dgrad = nullptr);   // wino*: see packed_wino_floats
// first layer (cin < 16): packed[kstep][ntile_total][64], k = 4*kstep + (lane>>4) = tap*cin + c, zero padded
int pack_first_weights(PackCtx* pc, hipStream_t s, const float* w, int cout, int cin, float* packed, int cout_off, int cout_total);
// the same layer as ONE bf16 K-step (9 * cin <= 32): packed16[ntile_total][64][8] bf16 (bf16 / fp8 datapaths)
int pack_first_weights16(PackCtx* pc, hipStream_t s, const float* w, int cout, int cin, unsigned short* packed16, int cout_off);
// batched packing: between begin and flush the pack_* entry points only record jobs; flush runs them in (at most) two kernels
void pack_batch_begin(PackCtx* pc);
int pack_batch_flush(PackCtx* pc, hipStream_t s);
int pack_batch_freeze(PackCtx* pc, hipStream_t s);         // flush + keep the device job table: pack_replay re-runs it (kernel launches only)
int pack_replay(PackCtx* pc, hipStream_t s);
int pack_copy(PackCtx* pc, hipStream_t s, const float* src, float* dst, size_t n);
int pack_bf16_submit(PackCtx* pc, hipStream_t s, const float* w, bool transposed, int cout, int cin, int taps, unsigned short* packed,
                     int cout_off, int ntile_total, int cin_split, int chunks_total, bool f16, bool flip = false, int chunk_off = 0);
size_t packed_conv_floats(int cout_total, int cin_total, int ksize);
// Winograd F(2x2, 3x3) weights of a 3x3 stride-1 conv W[cout][cin][3][3] (kernels_conv_wino.hip): U = G g G^T in fragment order
// packed[cin/16][16 positions][cout/16][64 lanes][4].  dgrad: the weights of the data-gradient convolution of the same tensor
// (cout = the forward conv's cin and vice versa, taps reversed).
size_t packed_wino_floats(int cout_total, int cin_total);
int pack_wino_weights(PackCtx* pc, hipStream_t s, const float* w, bool dgrad, int cout, int cin, float* packed);
// Winograd F(4x4, 3x3) (kernels_conv_wino4x.hip): the wide (>= 128 input channels) 3x3 stride-1 layers of the fp32 path, forward only;
// packed[cin/16][36 positions][cout/16][64 lanes][4]
struct ConvOutExtra;
constexpr int SMALL_GRID_BLOCKS = 256;   // launches with fewer blocks than CUs take narrower channel blocks (same kernels, more blocks)
constexpr int WINO4_MIN_CIN = 32;   // layers with at least this many input channels (cin, cout multiples of 32) take the F(4x4,3x3) form
size_t packed_wino4_floats(int cout_total, int cin_total);
int pack_wino4_weights(PackCtx* pc, hipStream_t s, const float* w, int cout, int cin, float* packed, bool dgrad = false);   // dgrad: (cout, cin) of the
                                                                                          // DATA-GRADIENT convolution (= the layer's cin, cout)
bool wino4_fits(int cin, int cout, int h, int w);
// layers with cin % 32 == 0 and cout % 32 == 0 take the 32x32x2-MFMA form (kernels_conv_wino4x.hip): its weight
// fragments are packed[cin/16][36 positions][cout/32][2][64 lanes][4] (same size, other order) - pack and launch ask this predicate
bool wino4_layout32(int cin, int cout);
int wino4_form(int cin, int cout);     // 0: none; 1 / 2 / 3: the 64-channel, 32-channel and register-epilogue kernels of kernels_conv_wino4x.hip
int launch_conv_wino4x(hipStream_t s, const ConvSrc& a, const float* wpack_wino4x, const float* bias, float* out, int pitch_out, int cout,
                       double* stats, int n, int h, int w, const ConvOutExtra* ex = nullptr);
int launch_conv_wino4(hipStream_t s, const ConvSrc& a, const float* wpack_wino4, const float* bias, float* out, int pitch_out, int cout,
                      double* stats, int n, int h, int w, const ConvOutExtra* ex = nullptr);
size_t packed_first_floats(int cout_total, int cin);
// Transposed conv k3 s2 p1 op1 in Winograd form F(2,2) (kernels_convt_wino.hip; inference): 25 instead of 36 multiplies per 2 x 2 input
// pixels; weights W[cin][cout][3][3] -> packed[cin/16][25 products][cout/16][64 lanes][4].  CS_CONVT_WINO=0 keeps the direct kernels.
struct ConvSrc;
size_t packed_convt_wino_floats(int cout_total, int cin_total);
int pack_convt_wino_weights(PackCtx* pc, hipStream_t s, const float* w, int cout, int cin, float* packed);
bool convt_wino_enabled();
bool convt_wino_fits(int cin_a, int cin_b, int cout, int h, int w);
int launch_convt_wino(hipStream_t s, const ConvSrc& a, const ConvSrc& b, const float* wpack_tw, const float* bias, float* out, int pitch_out,
                      int cout, double* stats, int n, int hin, int win);

// Extras used by the backward pass when a forward kernel computes a data gradient:
//   out = conv(...) + bias + addend ; output channels >= `split` are written to out2 (+ addend2) instead.
struct ConvOutExtra {
    const float* addend = nullptr;
    int pitch_add = 0;
    float* out2 = nullptr;
    int pitch_out2 = 0;
    int split = 0;
    const float* addend2 = nullptr;
    int pitch_add2 = 0;
};

// ---- convolutions ------------------------------------------------------------------------------
// 3x3 (pad 1) or 1x1 (pad 0) convolution, stride 1 or 2, fp32 MFMA 16x16x4.  Input = concat(a, b) along C.
// out: raw conv + bias (NHWC, pitch floats/pixel).  stats: [n][cout][2] doubles (sum, sum of squares) or null.
int launch_conv_mfma(hipStream_t s, int ksize, int stride, const ConvSrc& a, const ConvSrc& b, const float* wpack,
                     const float* bias, float* out, int pitch_out, int cout, double* stats, int n, int hin, int win,
                     const ConvOutExtra* ex = nullptr);

// 3x3 stride-1 pad-1 convolution as Winograd F(2x2, 3x3) on the fp32 matrix pipe (kernels_conv_wino.hip): 16 instead of 36
// multiplies per 2x2 output tile; fp32 throughout, same outputs / statistics contract as launch_conv_mfma.  cout % 32 == 0.
bool wino_enabled();           // CS_WINO=0 disables it (A/B against the direct kernels)
int wino_level();              // CS_WINO: 0 direct kernels, 1 F(2x2,3x3) everywhere, 2 (default) F(4x4,3x3) where the layer has a kernel for it
// the kernel addresses one image's input tensor with 32-bit byte offsets; larger tensors (a plain forward over a > 4 GB image
// level) take the direct kernels
static inline bool wino_fits(int h, int w, int pitch) { return (size_t)h * w * pitch * 4 < ((size_t)1 << 32); }
int launch_conv_wino(hipStream_t s, const ConvSrc& a, const float* wpack_wino, const float* bias, float* out, int pitch_out, int cout,
                     double* stats, int n, int h, int w, const ConvOutExtra* ex = nullptr);   // ex: addend only

// split-bf16 variant (kernels_conv_bf16.hip): `planes` = 1 (plain bf16), 2 (3 MFMA terms) or 3 (6 terms, fp32-grade).
// Weights come from pack_conv_weights_bf16 (always 3 planes; the kernel reads the first `planes`).
// act16: inputs (a.h16 / b.h16) and the output are bf16 tensors in HBM (plain-bf16 mode only: planes == 1); statistics still
// come from the fp32 accumulators.
int launch_conv_bf16s(hipStream_t s, int ksize, int stride, int planes, const ConvSrc& a, const ConvSrc& b,
                      const unsigned short* wpack, const float* bias, float* out, int pitch_out, int cout, double* stats, int n,
                      int hin, int win, bool act16 = false, const float* in_amax = nullptr, const ConvOutExtra* ex = nullptr);
int launch_convt_bf16s(hipStream_t s, int planes, const ConvSrc& a, const ConvSrc& b, const unsigned short* wpack, const float* bias,
                       float* out, int pitch_out, int cout, double* stats, int n, int hin, int win, bool act16 = false,
                       const ConvOutExtra* ex = nullptr);
size_t packed_bf16_elems(int cout_total, int cin_a, int cin_b, int ksize);
int pack_conv_weights_bf16(PackCtx* pc, hipStream_t s, const float* w, bool transposed, int cout, int cin, int ksize,
                           unsigned short* packed, int cout_off, int cout_total, int cin_split, bool f16 = false, bool flip = false, int chunk_off = 0);

// fp8 variant (kernels_conv_fp8.hip): e4m3 weights with one fp32 scale per output channel, inputs = ACTIVATED e4m3 tensors
// (ConvSrc::q8; launch_act_quant / launch_norm_act_add produce them) staged as plain 16-byte copies,
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales, fp32 accumulate, bf16 out (raw conv output for the next norm).
int launch_conv_fp8(hipStream_t s, int ksize, int stride, const ConvSrc& a, const ConvSrc& b, const unsigned char* wpack,
                    const float* wscale, const float* bias, float* out, int pitch_out, int cout, double* stats, int n, int hin, int win);
int launch_convt_fp8(hipStream_t s, const ConvSrc& a, const ConvSrc& b, const unsigned char* wpack, const float* wscale, const float* bias,
                     float* out, int pitch_out, int cout, double* stats, int n, int hin, int win);
size_t packed_fp8_bytes(int cout_total, int cin_a, int cin_b, int ksize);
int pack_conv_weights_fp8(hipStream_t s, const float* w, bool transposed, int cout, int cin, int ksize, unsigned char* packed, float* wscale,
                          int cout_off, int cout_total, int cin_split);

// first layer: cin < 16 (3), 3x3 pad 1, reads windows straight out of the image (tiler gather fused).
// image element strides (sc, sh, sw); per-window origins (y0, x0) in image coordinates (may be negative when the
// image is smaller than the roi); pixels outside the window or the image read as 0.
struct WindowSrc {
    const float* image = nullptr;
    int64_t sn = 0, sc = 0, sh = 0, sw = 0;   // sn used when origins == nullptr (plain batched forward)
    const int2* origins = nullptr;           // [n] (y0, x0) or nullptr
    int img_h = 0, img_w = 0;                // bounds of the image (or of each batch item)
    int roi_h = 0, roi_w = 0;                // window size = conv input size
    // inference: the same image as (r, g, b, 0) pixel records (launch_image_records) - float4, or 4 x bf16 when rec16 - [img_h][img_w]
    // with origins, [n][img_h][img_w] without; the record kernels of kernels_first.hip read these instead of `image`
    const void* rec = nullptr;
    bool rec16 = false;
};
int launch_conv_first(hipStream_t s, int stride, const WindowSrc& src, int cin, const float* wpack, const float* bias,
                      float* out, int pitch_out, int cout, double* stats, int n, bool out16 = false, int stats_c = 0,   // stats_c: statistics for channels [0, stats_c) only (0 = all)
                      int cout_pack = 0, int cout_off = 0,     // channels [cout_off, cout_off + cout) of a pack of cout_pack channels (0 = the whole pack)
                      const ConvSrc* add = nullptr);           // out = conv + bias + act(norm(*add)): the ResidualUnit's sum, fp32
// first layer from pixel records (kernels_first.hip; inference): wave-private tiles walked along the row, no block barrier; fp32 or,
// with src.rec16, the bf16 datapath (bf16 out); `add`: out = conv + bias + act(norm(*add)) in the datapath's own format
// rows [row_begin, row_end) only (row_end < 0: all): a rank that runs a share of a slide's windows converts the rows they read
int launch_image_records(hipStream_t s, const float* image, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int n, int h, int w, void* rec,
                         bool as16, int row_begin = 0, int row_end = -1);
bool first_rec_enabled();
int launch_conv_first_rec(hipStream_t s, const WindowSrc& src, const float* wpack, const float* bias, void* out, int pitch_out, int cout,
                          double* stats, int n, int stats_c = 0, int cout_pack = 0, int cout_off = 0, const ConvSrc* add = nullptr);
// bf16 / fp8 datapaths: the image is converted to bf16 while it is staged, one v_mfma_f32_16x16x32_bf16 per 16 x 16 output tile
// (the fp32 form needs seven 16x16x4 steps = 14x the matrix-pipe cycles), bf16 out
int launch_conv_first16(hipStream_t s, const WindowSrc& src, int cin, const unsigned short* wpack16, const float* bias, float* out,
                        int pitch_out, int cout, double* stats, int n, int stats_c = 0);

// transposed 3x3 stride 2 pad 1 output_padding 1 (MFMA), input = concat(a, b); out is [n, 2h, 2w, cout]
int launch_convt_mfma(hipStream_t s, const ConvSrc& a, const ConvSrc& b, const float* wpack, const float* bias,
                      float* out, int pitch_out, int cout, double* stats, int n, int hin, int win,
                      const ConvOutExtra* ex = nullptr);

// small-cout (<= 4) VALU kernels for the top of the decoder
// transposed conv cin -> cout(<=4); weights in PyTorch layout W[cin][cout][3][3] (no packing)
int launch_convt_small(hipStream_t s, const ConvSrc& a, const ConvSrc& b, const float* w_torch, const float* bias,
                       float* out, int cout, double* stats, int n, int hin, int win);
// final conv c -> c (c <= 4) on lazy input + identity residual of the same lazy input: out = conv(T(u)) + b + T(u)
int launch_conv_last(hipStream_t s, const ConvSrc& u, const float* w_torch, const float* bias, float* out, int c,
                     int n, int h, int w);

// ---- normalisation ------------------------------------------------------------------------------
// stats [n][c][2] (double) -> tab [n][c] (rstd, -mean*rstd); count = h*w; eps = 1e-5
int launch_norm_finalize(hipStream_t s, const double* stats, float2* tab, int n, int c, int count, float eps, bool snap_constant);
// eval-mode BatchNorm: tab[n][c] from running stats and affine parameters
int launch_bn_table(hipStream_t s, const float* weight, const float* bias, const float* mean, const float* var,
                    float2* tab, int n, int c, int c_total, float eps);
// training-mode BatchNorm: per-image sums [n][c_total][2] -> batch statistics; tab / xtab [n][c_total] (xtab may be null);
// running statistics (may be null) updated in place with `momentum` and the unbiased variance
int launch_bn_train_finalize(hipStream_t s, const double* stats, const float* weight, const float* bias, float* run_mean,
                             float* run_var, float2* tab, float2* xtab, int n, int c, int c_total, int count, float eps,
                             float momentum);
// out = T_y(y) + T_r(r)   (T = lazy norm/act or identity), all NHWC with channel count c
// out8 (fp8 mode): the sum is ALSO (or, with out == nullptr, only) written as e4m3 bytes with pixel pitch pitch8 - the form the
// fp8 convolutions stage without any arithmetic
int launch_norm_act_add(hipStream_t s, const ConvSrc& y, const ConvSrc& r, float* out, int pitch_out, int64_t npix_per_image, int n,
                        bool act16 = false, unsigned char* out8 = nullptr, int pitch8 = 0);
// fp8 mode: dst8[pix][c] = e4m3(T(src)) - InstanceNorm / BatchNorm table + PReLU applied ONCE per element instead of by every
// output-channel block of every consumer; src is a bf16 tensor with a lazy table
int launch_act_quant(hipStream_t s, const ConvSrc& src, unsigned char* dst8, int pitch8, int64_t npix_per_image, int n);

// ---- backward pass (kernels_bwd.hip) -----------------------------------------------------------
// InstanceNorm + PReLU backward of `y` (raw conv output + forward table/slope): da -> dy; red [n][c][2] and dslope
// are fp64 accumulators that must be zero on entry.
// BatchNorm (batch statistics): pass the forward's `xtab` (x_hat table) and batch = true - the two sums are then reduced over
// the whole batch, and d(gamma) = sum dz*x_hat / d(beta) = sum dz are written to dgamma / dbeta (fp32, may be null).
// amax (may be null): AMAX_SLOTS device floats that receive partial maxima of |dy| over the whole tensor (zero before the launch): the dynamic fp16 scale of
// the weight-gradient / data-gradient kernels on the 16-bit matrix pipe (kernels_bwd16.hip)
// dasum (may be null): [n][c] doubles, zero on entry, that receive the per-image column sums of `da` itself - `da` is then also the
// gradient of a conv bias (the raw residual branch), and its bias gradient needs no pass of its own
int launch_norm_bwd(hipStream_t s, const ConvSrc& y, const float* da, int da_pitch, float* dy, int dy_pitch, double* red,
                    double* dslope, int64_t npix, int n, const float2* xtab = nullptr, bool batch = false, float* dgamma = nullptr,
                    float* dbeta = nullptr, float* amax = nullptr, double* dasum = nullptr);
// dst[c] = sum over pixels (fp32 out, fp64 final accumulation); partial: colsum_blocks(npix_total) * c floats of scratch
int colsum_blocks(int64_t npix_total);
int launch_colsum(hipStream_t s, const float* src, int pitch, int c, int64_t npix_total, float* partial, float* dst);
struct WgradNarrow {          // "x-like" operand with <= 4 channels addressed by element strides
    const float* ptr = nullptr;
    int c = 0;
    int64_t sn = 0, sc = 0, sh = 0, sw = 0;
};
// part[k][tap][co_off + i][ci_off + j] = sum over split-K slice k of a[pix][i] * b[pix*stride + tap - pad][j]  (plain stores;
// every element of the launch's [co_off, co_off+a_c) x [ci_off, ..) block is written, nothing needs zeroing).
// split_k from wgrad_split_k(); launches that share one partial buffer (fused / concatenated operands) must use the same value.
int wgrad_split_k(int a_c, int b_c, int n, int ho, int wo, int ksize = 3, int stride = 2);   // (ksize, stride) of the launch that will use it
// a_h16 / b.h16: the operand is an activation tensor of the bf16 tape (bf16 elements; the bf16 training mode)
int launch_wgrad(hipStream_t s, int ksize, int stride, const float* a, int a_pitch, int a_c, const ConvSrc& b, const WgradNarrow* nb,
                 float* part, int split_k, int co_total, int ci_total, int co_off, int ci_off, int n, int ho, int wo, int hi, int wi,
                 bool a_h16 = false);
// sums the split_k partials and converts to the PyTorch layout dst[row][col][tap]
// unscale (may be null): the partials come from launch_wgrad16 in fp16x3 mode and carry the operands' power-of-two scales
struct WgradUnscale {
    const float* a_amax = nullptr;   // AMAX_SLOTS device floats behind each dynamic scale (null: the static activation scale)
    const float* b_amax = nullptr;
};
// dst2 (may be null): the NEXT `rows` rows of the same partial buffer go to a second tensor in the same launch (a fused pair)
int launch_unpack_wgrad(hipStream_t s, const float* part, int split_k, int rows_total, int cols_total, int row_off, int col_off,
                        int rows, int cols, int taps, float* dst, const WgradUnscale* unscale = nullptr, float* dst2 = nullptr,
                        const WgradUnscale* unscale2 = nullptr);
// ---- weight gradients on the 16-bit matrix pipe (kernels_bwd16.hip) ----
constexpr float F16_ACT_SCALE = 16.f;          // static pre-scale of activation operands in the split-fp16 modes (== F16_X_SCALE of the forward)
constexpr int AMAX_SLOTS = 32;                 // a gradient tensor's max |dy| is kept as AMAX_SLOTS partial maxima (see amax_commit)
#ifdef __HIPCC__
// power-of-two scale of a gradient operand: its largest element lands in [2^13, 2^14); null pointer = an activation (static scale)
__device__ __forceinline__ float f16_dyn_scale(const float* amax_slots) {
    if (!amax_slots) return F16_ACT_SCALE;
    unsigned bits = 0;
#pragma unroll
    for (int i = 0; i < AMAX_SLOTS; ++i) {
        const unsigned b = reinterpret_cast<const unsigned*>(amax_slots)[i];
        bits = b > bits ? b : bits;
    }
    const int ex = (int)((bits >> 23) & 0xffu) - 127;               // floor(log2 max|v|)
    if (ex < -100 || ex > 100) return 1.f;                          // an all-zero (or broken) tensor: any scale will do
    return __builtin_bit_cast(float, (unsigned)(127 + 13 - ex) << 23);
}
// Running maximum of non-negative floats (their bits order like unsigned integers), called by EVERY thread of a 256-thread block with
// its own maximum: wave shuffle -> LDS -> ONE fire-and-forget atomicMax per block, spread over AMAX_SLOTS words by block index
// (readers take the maximum of all).  Measured on the training step's norm backward (16 launches of ~4096 blocks): an atomic per
// wave on one word +0.9 ms, a look-before-you-leap load in front of it (the wave then waits for an L2 round trip before it can
// retire) +0.35...0.55 ms, this form: see DESIGN.md.
__device__ __forceinline__ void amax_commit(unsigned* slots, float thread_max) {
    __shared__ float amax_sh[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) thread_max = fmaxf(thread_max, __shfl_xor(thread_max, off));
    if ((threadIdx.x & 63) == 0) amax_sh[threadIdx.x >> 6] = thread_max;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(amax_sh[0], amax_sh[1]), fmaxf(amax_sh[2], amax_sh[3]));
        if (m > 0.f) atomicMax(slots + ((blockIdx.x + 7u * blockIdx.y) & (AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, m));
    }
}
#endif
int wgrad16_split_k(int a_c, int b_c, int n, int ho, int wo);
// mode 0: split fp16, 3 terms (fp32-grade); mode 1: plain bf16.  3x3 only, stride 1 / 2, both operands >= 16 channels; a_amax / b_amax:
// device scalars max|operand| for gradient operands (dynamic scale, mode 0), null for activations.  Partial layout as launch_wgrad.
int launch_wgrad16(hipStream_t s, int mode, int stride, const float* a, int a_pitch, int a_c, const float* a_amax, const ConvSrc& b,
                   const float* b_amax, float* part, int split_k, int co_total, int ci_total, int co_off, int ci_off, int n, int ho, int wo,
                   int hi, int wi, bool a_h16 = false);
int launch_amax(hipStream_t s, const float* src, int64_t numel, float* out);   // out[AMAX_SLOTS]: partial maxima of |src|; zero before the first use
constexpr int NORM_BWD_FUSED_MAX_PIX = 1024;   // InstanceNorm backward of images with at most this many pixels: one launch (norm_bwd_fused_kernel)
constexpr int NORM_DSLOPE_SLOTS = 64;          // partial accumulators of one PReLU slope gradient (launch_norm_bwd's `dslope`)
struct F64Job { const double* src; float* dst; int count, slots; };
struct F64Jobs { F64Job j[96]; int n = 0; };   // passed by value as a kernel argument (2.3 KB)
int launch_f64_jobs(hipStream_t s, const F64Jobs& jobs);   // dst[i] = (float) sum over slots of src[slot * count + i], one block per job
// dw_db: 84 doubles = [81 weight gradients | 3 bias gradients]; partial: conv_last_bwd_partial_floats(n, h, w) floats of scratch
size_t conv_last_bwd_partial_floats(int n, int h, int w);
int launch_conv_last_bwd(hipStream_t s, const float* dlogits, const ConvSrc& u, const float* w_torch, float* d_au, double* dw_db,
                         float* partial, int n, int h, int w);
int launch_convt_small_bwd_data(hipStream_t s, const float* du, const float* w_torch, int cin_total, int c_begin, int c_count,
                                float* dst, int dst_pitch, bool accumulate, int n, int hin, int win);

// zero-fill of one or two 16-byte aligned regions as ONE kernel launch (kernels_misc.hip; other alignments fall back to hipMemsetAsync)
int launch_zero(hipStream_t s, void* a, size_t a_bytes, void* b = nullptr, size_t b_bytes = 0);

// ---- tiler finalize -------------------------------------------------------------------------------
struct SwiFinalizeArgs {
    const float* win_logits;    // [n_h*n_w][roi_h][roi_w][c] NHWC
    cs_swi_plan plan;
    int c;                      // classes (<= 4)
    int ih_base;                // window-row index of win_logits[0] (a rank holds only the window rows it ran)
    int row_begin, row_end;     // image rows finalised by this call
    int col_begin, col_end;     // ... and columns (col_end == 0: the whole width)
    float* logits_nchw;         // [c][img_h][img_w] or null
    float* prob1;               // [img_h][img_w] or null
    uint8_t* mask;              // [img_h][img_w] or null
};
int launch_swi_finalize(hipStream_t s, const SwiFinalizeArgs& a);

// ---- input normalisation (kernels_pre.hip) ------------------------------------------------------
size_t normalize_workspace_bytes(int h, int w);
int normalize_image(hipStream_t s, const void* img, int is_u16, int h, int w, int c_in, float* out_hwc, unsigned char* u8_out,
                    void* ws, size_t ws_bytes);

// ---- F1 harness pieces (kernels_metric.hip) -----------------------------------------------------
int label_max(hipStream_t s, const int32_t* a, const int32_t* b, int64_t n, int32_t* out2);
// kernels_augment.hip: the training augmentations of a batch, from an HBM-resident dataset
size_t augment_workspace_bytes(int n, int c, int s);
int augment_batch(hipStream_t st, const float* img_arena, const unsigned char* lab_arena, const ::cs_aug_params* params_dev, int n, int c, int s,
                  float* out_img, unsigned char* out_lab, void* workspace, size_t ws_bytes);
// kernels_deflate.hip: zlib streams of the label TIFF's strips, on the device
size_t label_deflate_bound(int h, int w, int rps);
size_t label_deflate_workspace_bytes(int h, int w, int rps);
int label_deflate(hipStream_t s, const int* labels, int h, int w, int rps, unsigned char* out, size_t out_capacity, unsigned* meta,
                  void* workspace, size_t ws_bytes);
int label_overlap(hipStream_t s, const int32_t* x, const int32_t* y, int64_t n, int32_t nx1, int32_t ny1, unsigned int* overlap);
size_t remove_boundary_workspace_bytes(int32_t n_labels);
int remove_boundary_cells(hipStream_t s, const int32_t* lab, int h, int w, int border, int32_t n_labels, int32_t* out, int32_t* n_out,
                          void* ws, size_t ws_bytes);

// ---- DiceCE loss (kernels_loss.hip) ---------------------------------------------------------------
// logits / target: [n, c, h, w] fp32 addressed by element strides ls[4] / ts[4] (n, c, h, w); c in 2..4.  ws: dice_ce_workspace_bytes
// (fp64 sums: zeroed and filled by the forward, read by the backward).  grad_logits is written with the logits' strides.
size_t dice_ce_workspace_bytes(int n, int c);
int dice_ce_forward(hipStream_t s, const float* logits, const int64_t* ls, const float* target, const int64_t* ts, int n, int c, int h, int w,
                    float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce, float* loss_out, void* ws, size_t ws_bytes);
int dice_ce_backward(hipStream_t s, const float* logits, const int64_t* ls, const float* target, const int64_t* ts, int n, int c, int h, int w,
                     float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce, const float* grad_loss, const void* ws,
                     size_t ws_bytes, float* grad_logits);

// AdamW over many tensors in one launch: table [n_tensors][3] = (parameter pointer, flat offset, element count), offsets ascending
int adamw_step_dev(hipStream_t s, const int64_t* table, int n_tensors, int64_t total, const float* grad, float* exp_avg, float* exp_avg_sq,
                   float* hyper, int* step);
int adamw_step(hipStream_t s, const int64_t* table, int n_tensors, int64_t total, const float* grad, float* exp_avg, float* exp_avg_sq,
               float lr, float beta1, float beta2, float eps, float weight_decay, int step);

// ---- connected components ----------------------------------------------------------------------
size_t ccl_workspace_bytes(int h, int w);
int ccl_postprocess(hipStream_t s, const uint8_t* mask, int h, int w, int hole_area, int min_size, int32_t* labels,
                    int32_t* n_labels, void* ws, size_t ws_bytes);
int interior_map(hipStream_t s, const int32_t* inst, int h, int w, int min_size, uint8_t* out, void* ws, size_t ws_bytes);
int ccl_label(hipStream_t s, const uint8_t* mask, int h, int w, int conn, int32_t* labels, int32_t* n_labels, void* ws,
              size_t ws_bytes);
// strip-local labelling with a seam merge (one rank per strip of rows; kernels_ccl.hip)
int ccl_strip_halo(int hole_area, int min_size);
int ccl_strip_edge_ints(int w);
size_t ccl_seam_workspace_bytes(int world, int w);
int ccl_strip_local(hipStream_t s, const uint8_t* mask_ext, int ext_h, int w, int strip_row0, int strip_h, int hole_area, int min_size,
                    int32_t* edge_rec, void* ws, size_t ws_bytes);
int ccl_strip_resolve(hipStream_t s, const int32_t* all_recs, int world, int rank, int ext_h, int w, int strip_h, int32_t* edge_ids,
                      int32_t* n_labels, void* ws, size_t ws_bytes, void* seam_ws, size_t seam_bytes);
int ccl_strip_finish(hipStream_t s, const int32_t* all_edge_ids, int world, int rank, int ext_h, int w, int strip_h, int32_t* labels,
                     void* ws, size_t ws_bytes);

}  // namespace cs
