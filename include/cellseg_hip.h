/* cellseg_hip.h - C-ABI of the MI355X (gfx950) engine for the NeurIPS-CellSeg 3-class U-Net hot path.
 *
 * The reference (JunMa11/NeurIPS-CellSeg) has no FFI layer: its hot path is three Python call sites that
 * dispatch into MONAI / PyTorch / scikit-image.  Each entry point below names the reference interface it
 * replaces (file:line under the reference tree).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless the name ends in `_host`
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); calls return without synchronising
 *     unless documented otherwise
 *   - return 0 on success, a negative cs_status otherwise; cs_last_error() gives a message (thread-local)
 *   - activations are NHWC fp32 inside the engine; images cross the boundary with explicit element strides,
 *     so the reference's `[1,3,H,W]` tensor (physically NHWC after its .permute, predict.py:108) and a
 *     contiguous NCHW tensor are both accepted without a copy
 *   - thread-compatible, not thread-safe: one host thread per handle (one process per GPU under data parallel)
 */
#ifndef CELLSEG_HIP_H
#define CELLSEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cs_status {
    CS_OK = 0,
    CS_ERR_INVALID = -1,      /* bad argument / unsupported shape            */
    CS_ERR_HIP = -2,          /* a HIP runtime call failed                   */
    CS_ERR_WORKSPACE = -3,    /* caller workspace too small                  */
    CS_ERR_NOT_LOADED = -4,   /* a parameter was never loaded                */
    CS_ERR_UNKNOWN_KEY = -5   /* state-dict key does not exist in this net   */
} cs_status;

enum { CS_NORM_INSTANCE = 0, CS_NORM_BATCH = 1 };
enum { CS_ACT_PRELU = 0, CS_ACT_RELU = 1 };

const char* cs_last_error(void);
int cs_version(void);
/* Number of visible HIP devices (>=1) or a negative cs_status: lets a host fail loudly without torch. */
int cs_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * U-Net engine  -  replaces monai.networks.nets.UNet(spatial_dims=2, ...) as instantiated at
 *   baseline/model_training_3class.py:201-208 and baseline/predict.py:45-52
 * Topology: MONAI residual U-Net, kernel 3, every stride 2, `num_res_units` >= 1, ADN order "NDA".
 * ---------------------------------------------------------------------------------------------- */
typedef struct cs_unet cs_unet;

/* channels[n_channels], strides are all 2 (n_channels-1 of them).  Allocates the packed-weight arena on the
 * current device (the only allocation the engine ever makes; it happens here, never on the launch path). */
int cs_unet_create(cs_unet** out, int in_channels, int out_channels, const int* channels_host, int n_channels,
                   int num_res_units, int norm_kind, int act_kind);
int cs_unet_destroy(cs_unet* net);
/* Arithmetic of the 3x3 / 1x1 convolutions on the matrix cores:
 *   0  v_mfma_f32_16x16x4_f32, exact fp32 (default; the 1e-4 logit-parity path)
 *   3  split-bf16, 3 planes / 6 MFMA terms on v_mfma_f32_16x16x32_bf16: fp32-grade results (~1e-5 logit error)
 *   2  split-bf16, 2 planes / 3 terms (~1e-4)
 *   1  the bf16 DATAPATH (throughput mode; judged by F1 / argmax, not by logits): bf16 weights, fp32 accumulate, and - outside
 *      training mode - every activation tensor between the first conv and the 3-channel top layers is bf16 IN HBM
 *      (statistics still come from the fp32 accumulators); BASELINE.json configs[1]
 *   4  split-fp16: planes h0 = fp16(x'), h1 = fp16(x' - h0) (22 mantissa bits) of the power-of-two pre-scaled operands
 *      (weights * 2^8, staged activations * 2^4, sum * 2^-12), 3 terms on v_mfma_f32_16x16x32_f16 in one accumulator:
 *      fp32-grade results at half the MFMAs of mode 3; operands must lie inside fp16's range after scaling
 *      (|x| < 4094, |w| < 255: normalised activations and trained weights do).
 *   5  fp8 (BASELINE.json configs[4]): weights stored as OCP e4m3 with one fp32 scale per output channel (amax / 448, computed
 *      when the parameter is loaded), activations bf16 in HBM and converted to e4m3 while they are staged (after the lazy
 *      InstanceNorm + PReLU), v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (2x the bf16 rate), fp32 accumulate,
 *      weight scale applied in the epilogue, bf16 out.  Inference only; judged by argmax / F1 like mode 1.
 * Modes 0, 2, 3, 4 keep activations fp32 in HBM.
 * Parameters must be (re)loaded after switching between the fp32 / bf16 (1..3) / fp16 (4) / fp8 (5) families. */
int cs_unet_set_precision(cs_unet* net, int planes);
/* on != 0: every cs_unet_load_param also packs the data-gradient layout of the tensor, and cs_unet_forward records the
 * activation tape cs_unet_backward replays (activations then stay fp32 in every precision mode).
 * Parameters loaded under the wrong mode / precision are refused at forward / backward time (CS_ERR_NOT_LOADED). */
int cs_unet_set_training(cs_unet* net, int on);
/* BatchNorm networks (norm_kind CS_NORM_BATCH), replaces `model.train()` / `model.eval()` for nn.BatchNorm2d inside
 * monai's ADN (the norm= argument of the UNet constructor, baseline/model_training_3class.py:201-208):
 *   on != 0  every cs_unet_forward normalises with the statistics of ITS batch (mean / biased variance over N*H*W per channel,
 *            reduced from the conv epilogues' per-image fp64 sums) and updates the engine's running_mean / running_var
 *            (momentum 0.1, unbiased variance); cs_unet_backward then also returns d(weight) / d(bias) of every BatchNorm.
 *            Under data parallel each rank uses its LOCAL batch statistics (torch DDP's default; no SyncBatchNorm).
 *   on == 0  running statistics (default; sliding-window inference always uses them). */
int cs_unet_set_batch_stats(cs_unet* net, int on);
/* Copy a tensor the engine keeps verbatim back to device memory `dst` (BatchNorm weight / bias / running_mean / running_var,
 * PReLU slope): how the host mirrors the updated running statistics into its state dict.  Batches with cs_unet_load_begin/end. */
int cs_unet_read_param(cs_unet* net, const char* key_host, float* dst, int64_t numel, void* stream);
/* n (1..4) internal streams for cs_swi_infer: window chunks are dealt round-robin to them (forked from / joined back to the
 * caller's stream with events), so the HBM-bound kernels of one chunk overlap the MFMA-bound ones of another.  The side
 * streams and events are created here, never on the launch path.  cs_swi_workspace_bytes scales with n. */
int cs_unet_set_streams(cs_unet* net, int n);
/* Introspection for the per-layer parity tests (tests/test_gpu_layers.py; no counterpart in the reference, whose layers are
 * torch.nn modules one can hook - baseline/predict.py:45-52).  With tracing on, cs_unet_forward records where every activation
 * tensor and lazy-norm table of the forward lives inside the CALLER's workspace; afterwards
 *   cs_unet_trace_count  number of recorded tensors
 *   cs_unet_trace_entry  name (64 chars, e.g. "down1.E", "bottom.tab0", "up2.U") and info[8] = {byte offset in the workspace, n, h, w,
 *                        channels, pixel pitch in elements, kind (0 fp32, 1 bf16, 2 e4m3 activations NHWC; 3 float2 (scale, shift)
 *                        table [n][channels] of act(norm(x)) = prelu(x * scale + shift)), 0}.
 * The tests read the tensors out of the workspace, recompute one layer at a time on the CPU from the layer's own inputs and compare. */
int cs_unet_trace(cs_unet* net, int on);
int cs_unet_trace_count(const cs_unet* net);
int cs_unet_trace_entry(const cs_unet* net, int index, char* name_out_host, int64_t* info_out_host);

/* State-dict surface (replaces model.load_state_dict / model.state_dict, predict.py:80-81,
 * model_training_3class.py:269-274).  Keys are MONAI's: "model.0.conv.unit0.conv.weight", ... */
int cs_unet_num_params(const cs_unet* net);
/* key_out must hold >= 160 bytes; shape_out[4]; returns rank (1..4) or negative status. */
int cs_unet_param_info(const cs_unet* net, int index, char* key_out_host, int64_t* shape_out_host);
/* Copy + repack one tensor given in PyTorch layout (conv [Cout,Cin,k,k], transposed conv [Cin,Cout,k,k],
 * bias [C], PReLU slope [1], BatchNorm weight/bias/running_mean/running_var [C]) from device memory. */
int cs_unet_load_param(cs_unet* net, const char* key_host, const float* data, int64_t numel, void* stream);
/* Optional batching of parameter uploads: between cs_unet_load_begin and cs_unet_load_end, cs_unet_load_param only records
 * the packing work; cs_unet_load_end runs all of it in ONE kernel on `stream` (a training step re-packs all 63 tensors after
 * every optimiser update).  The source buffers must stay alive and unchanged until cs_unet_load_end has been called. */
int cs_unet_load_begin(cs_unet* net);
int cs_unet_load_end(cs_unet* net, void* stream);

/* Training loops whose parameter tensors stay where they are (model_training_3class.py:247-267: the optimiser updates them in
 * place, step after step): cs_unet_bind_params records the (re)pack of ALL parameters from `param_ptrs_host[i]` (device pointer of
 * parameter i in cs_unet_param_info order, PyTorch layout) ONCE - under the precision / training mode set at that moment - runs
 * it, and keeps the job table on the device; cs_unet_repack then re-runs it with two kernel launches and no host bookkeeping (what
 * follows every optimiser step), and cs_unet_readback copies the BatchNorm running statistics the last train-mode forward updated
 * back into the bound running_mean / running_var tensors.  Both are plain launches: they can be stream-captured. */
int cs_unet_bind_params(cs_unet* net, const float* const* param_ptrs_host, int n_params, void* stream);
int cs_unet_repack(cs_unet* net, void* stream);
int cs_unet_readback(cs_unet* net, void* stream);

/* Bytes of caller workspace needed for a forward over `n` windows of h x w (multiples of 2^(levels)). */
size_t cs_unet_workspace_bytes(const cs_unet* net, int n, int h, int w);

/* forward: replaces `model(inputs)` (model_training_3class.py:255) in eval/no-grad use.
 * x: n images of in_channels x h x w addressed as x[i*sn + c*sc + y*sh + xx*sw] (element strides).
 * logits_nhwc: [n,h,w,out_channels] fp32 contiguous. */
int cs_unet_forward(cs_unet* net, const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                    int n, int h, int w, float* logits_nhwc, void* workspace, size_t workspace_bytes, void* stream);

/* backward: replaces `loss.backward()` through the network (model_training_3class.py:260).  Uses the activations the
 * LAST cs_unet_forward left in its workspace (which must still be intact) and the input it read.  Writes every
 * parameter gradient, in PyTorch layout, into one flat fp32 buffer of cs_unet_grad_numel() floats at
 * cs_unet_grad_offset(param_index) - the flat buffer is the single bucket a data-parallel host all-reduces.
 * InstanceNorm networks, and BatchNorm networks whose forward ran under cs_unet_set_batch_stats(1). */
int64_t cs_unet_grad_numel(const cs_unet* net);
int64_t cs_unet_grad_offset(const cs_unet* net, int param_index);
size_t cs_unet_backward_workspace_bytes(const cs_unet* net, int n, int h, int w);
int cs_unet_backward(cs_unet* net, const float* grad_logits_nhwc, void* workspace, size_t workspace_bytes,
                     float* flat_grad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sliding-window inference  -  replaces monai.inferers.sliding_window_inference(x, roi, sw_batch, model)
 *   (baseline/predict.py:109, baseline/model_training_3class.py:291-293; constant blending, zero padding)
 * fused with softmax(dim=1)[0,1] (predict.py:110-111) and `> 0.5` (predict.py:113).
 * ---------------------------------------------------------------------------------------------- */
typedef struct cs_swi_plan {
    int img_h, img_w;          /* unpadded image                                 */
    int roi_h, roi_w;
    int pad_top, pad_left;     /* symmetric padding when the image < roi        */
    int pad_h, pad_w;          /* padded size = max(img, roi)                    */
    int interval_h, interval_w;
    int n_h, n_w;              /* windows per dimension; total = n_h * n_w       */
} cs_swi_plan;

int cs_swi_make_plan(int img_h, int img_w, int roi_h, int roi_w, double overlap, cs_swi_plan* plan_host);
/* start offset (in padded coordinates) of window i along a dimension */
int cs_swi_window_start(int i, int interval, int padded_size, int roi);

size_t cs_swi_workspace_bytes(const cs_unet* net, const cs_swi_plan* plan_host, int engine_batch);

/* image: in_channels x img_h x img_w with element strides (sc, sh, sw).  Outputs (any may be NULL):
 *   logits_nchw [out_channels, img_h, img_w] fp32  - the tensor the reference's call returns
 *   prob1       [img_h, img_w] fp32                - softmax(dim=1)[0,1]
 *   mask        [img_h, img_w] uint8               - prob1 > 0.5
 * Windows owned by this rank: indices k with first_window <= k < first_window + n_windows in the row-major
 * window list, and output rows [row_begin, row_end): a rank that owns a strip of rows runs every window that
 * intersects it (multi-GPU tiled inference); single GPU: 0, n_h*n_w, 0, img_h. */
int cs_swi_infer(cs_unet* net, const cs_swi_plan* plan_host, const float* image, int64_t sc, int64_t sh, int64_t sw,
                 int engine_batch, int row_begin, int row_end,
                 float* logits_nchw, float* prob1, uint8_t* mask,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Window pass alone (the first half of cs_swi_infer): runs every window of window rows [win_row_begin, win_row_end) through the
 * network and writes their logits to win_logits[(ih - win_row_begin) * n_w + iw][roi_h][roi_w][out_channels] fp32 - the layout
 * cs_swi_finalize reads with ih_base = win_row_begin.  workspace: cs_swi_windows_workspace_bytes (activations + window origins,
 * no window-logit buffer).  Together with cs_swi_partition and cs_swi_finalize this is the multi-GPU split of ONE image by WINDOW
 * rows ("split the window list evenly and exchange seam partial sums with one neighbour send/recv"): a rank runs its
 * own window rows, receives from the rank above the lower part of that rank's last window row(s) into the slots in front of
 * its own, and finalises its output rows - no window is evaluated twice.  The reference is single-device
 * (baseline/predict.py:109 calls sliding_window_inference on the whole image). */
size_t cs_swi_windows_workspace_bytes(const cs_unet* net, const cs_swi_plan* plan_host, int engine_batch);
int cs_swi_windows(cs_unet* net, const cs_swi_plan* plan_host, const float* image, int64_t sc, int64_t sh, int64_t sw,
                   int engine_batch, int win_row_begin, int win_row_end, float* win_logits,
                   void* workspace, size_t workspace_bytes, void* stream);
/* Window-row partition of one image over `world` ranks (shares differ by at most one window row, the larger shares last: rank 0,
 * which also labels the gathered mask, takes a smaller one - 31 rows over 8 ranks = 3,4,4,4,4,4,4,4):
 *   [win_row_begin, win_row_end)  window rows rank `rank` evaluates
 *   [row_begin, row_end)          output rows it finalises: from the first image row of its first window row to the first image row
 *                                 of the next rank's first window row (rank 0 starts at 0, the last rank ends at img_h)
 *   [halo_begin, win_row_begin)   window rows of EARLIER ranks that still cover row_begin (one row for overlap <= 0.5, two next to a
 *                                 clamped last window): their window-local rows >= row_begin + pad_top - start(ih) are needed */
typedef struct cs_swi_strip {
    int win_row_begin, win_row_end;
    int halo_begin;
    int row_begin, row_end;
} cs_swi_strip;
int cs_swi_partition(const cs_swi_plan* plan_host, int rank, int world, cs_swi_strip* out_host);

/* Window-LIST partition (round 5): the row-major window list cut into `world` contiguous shares that differ by at most ONE window
 * (961 windows over 8 ranks = 120 or 121 each; whole window rows give 3,4,...,4 rows = 93...124).  A pixel is finalised by the rank that
 * evaluates the LAST window (highest index) covering it, so every window a rank still needs has a lower index and comes from an earlier
 * rank - the same one-directional seam as the row partition, and the sums keep the reference's window-index order (bit-identical result).
 *   [win_begin, win_end)            window indices rank `rank` evaluates (cs_swi_windows_range)
 *   [slot_row_begin, slot_row_end)  window rows its window buffer holds: its own rows and the earlier rows that still cover its first
 *                                   image row; window k lives in slot k - slot_row_begin * n_w
 *   rect[n_rects][4]                the image rectangles (row_begin, row_end, col_begin, col_end) it finalises (cs_swi_finalize_rect):
 *                                   the rest of its first window row's band, whole bands, the beginning of its last band */
typedef struct cs_swi_share {
    int win_begin, win_end;
    int slot_row_begin, slot_row_end;
    int n_rects;
    int rect[3][4];
} cs_swi_share;
int cs_swi_partition_windows(const cs_swi_plan* plan_host, int rank, int world, cs_swi_share* out_host);
/* cs_swi_windows for a range of window INDICES: logits of window k go to win_logits[k - win_begin]. */
int cs_swi_windows_range(cs_unet* net, const cs_swi_plan* plan_host, const float* image, int64_t sc, int64_t sh, int64_t sw,
                         int engine_batch, int win_begin, int win_end, float* win_logits,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Finalize step alone (average overlapping windows in window-index order, softmax, threshold) for window
 * predictions produced by any predictor: win_logits [(#window rows from ih_base) * n_w][roi_h][roi_w][classes]. */
int cs_swi_finalize(const cs_swi_plan* plan_host, const float* win_logits, int classes, int ih_base,
                    int row_begin, int row_end, float* logits_nchw, float* prob1, uint8_t* mask, void* stream);
/* ... restricted to the image rectangle [row_begin, row_end) x [col_begin, col_end). */
int cs_swi_finalize_rect(const cs_swi_plan* plan_host, const float* win_logits, int classes, int ih_base,
                         int row_begin, int row_end, int col_begin, int col_end,
                         float* logits_nchw, float* prob1, uint8_t* mask, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input normalisation  -  replaces normalize_channel + the channel fix-up + `/ np.max`
 *   (baseline/predict.py:14-21, :94-108): per-channel 1-99 percentile stretch of the non-zero pixels
 *   (np.percentile linear interpolation, float64), truncation to uint8, division by the image maximum.
 * image: [h, w, channels] uint8 (bytes_per_sample 1) or uint16 (2), contiguous; channels 1 (gray -> replicated)
 * or >= 3 (first three used).  out_hwc: float32 [h, w, 3] in [0, 1] - pass it to cs_swi_infer with strides
 * (sc, sh, sw) = (1, 3 w, 3).  u8_hwc (optional): the intermediate uint8 image.  Bit-exact with numpy.
 * ---------------------------------------------------------------------------------------------- */
size_t cs_normalize_workspace_bytes(int h, int w);
int cs_normalize_image(const void* image, int bytes_per_sample, int h, int w, int channels, float* out_hwc,
                       uint8_t* u8_hwc, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Label-map writer  -  replaces the compression half of
 *   tif.imwrite(join(output_path, name + '_label.tiff'), test_pred_mask, compression='zlib')
 *   (baseline/predict.py:114, inside the per-image timer :106-116).
 * labels: int32 [h][w] in HBM.  The map is cut into TIFF strips of rows_per_strip rows; every strip becomes one zlib
 * stream (RFC 1950, TIFF compression 8: fixed-Huffman run-length coding of the labels, chunks joined by sync-flush
 * markers).  out: the streams back to back, each on an even offset; meta (device, uint32 [2 n_strips + 2]):
 * (offset in out, byte count) per strip, then the total byte count and an overflow flag (1: out_capacity too small,
 * nothing written).  cs_label_deflate_bound() is the worst-case total; typical label maps need 1-2 % of h*w*4.
 * The host writes header + out[0:total] + IFD (cellseg_amd.preprocess.write_label_tiff does).
 * ---------------------------------------------------------------------------------------------- */
size_t cs_label_deflate_workspace_bytes(int h, int w, int rows_per_strip);
size_t cs_label_deflate_bound(int h, int w, int rows_per_strip);
int cs_label_deflate(const int32_t* labels, int h, int w, int rows_per_strip, uint8_t* out, size_t out_capacity,
                     uint32_t* meta, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training augmentations on the device  -  replaces the per-sample work of the MONAI transform chain
 *   SpatialPadd, RandSpatialCropd, RandAxisFlipd, RandRotate90d, RandGaussianNoised, RandAdjustContrastd,
 *   RandGaussianSmoothd, RandHistogramShiftd, RandZoomd   (baseline/model_training_3class.py:119-162)
 * for a whole batch, from a dataset resident in HBM.  The host draws one parameter record per sample (which transforms
 * fire and with what values); the library does everything that touches pixels.
 * img_arena: float32, every image [channels][src_h][src_w] (already intensity-scaled, :126) at element offset img_off;
 * lab_arena: uint8 labels [src_h][src_w] at lab_off.  out_img: [n][channels][size][size] float32, out_lab: [n][size][size].
 * ---------------------------------------------------------------------------------------------- */
#define CS_AUG_MAX_TAPS 17
typedef struct cs_aug_params {
    int64_t img_off, lab_off;      /* element offsets of the sample inside the arenas                                         */
    int32_t src_h, src_w;
    int32_t pad_top, pad_left;     /* SpatialPadd: zero rows / columns in front of the image (symmetric padding's first half)   */
    int32_t crop_y, crop_x;        /* RandSpatialCropd: window origin in the padded image                                      */
    int32_t flip_axis;             /* RandAxisFlipd: 0 none, 1 rows, 2 columns                                                 */
    int32_t rot_k;                 /* RandRotate90d: number of counter-clockwise quarter turns                                 */
    float noise_std;               /* RandGaussianNoised: < 0 = off                                                            */
    uint32_t noise_seed;
    float gamma;                   /* RandAdjustContrastd: <= 0 = off                                                          */
    int32_t taps_y_n, taps_x_n;    /* RandGaussianSmoothd: erf taps along rows / columns (odd counts, 0 = axis untouched)      */
    float taps_y[CS_AUG_MAX_TAPS], taps_x[CS_AUG_MAX_TAPS];
    int32_t hist_n;                /* RandHistogramShiftd: control points (0 = off, at most 4)                                 */
    double hist_y[4];              /*   their images as fractions of [min, max]; the reference points are evenly spaced        */
    float zoom;                    /* RandZoomd: <= 0 = off                                                                    */
    int32_t zoom_h, zoom_w;        /*   size of the resampled image: floor(size * zoom)                                        */
} cs_aug_params;
size_t cs_augment_workspace_bytes(int n, int channels, int size);
int cs_augment_batch(const float* img_arena, const uint8_t* lab_arena, const cs_aug_params* params_device, int n, int channels,
                     int size, float* out_img, uint8_t* out_lab, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Connected-component post-process  -  replaces
 *   measure.label(morphology.remove_small_objects(morphology.remove_small_holes(mask), 16))
 *   (baseline/predict.py:113): 4-connected holes < 64 filled, 4-connected objects < 16 removed,
 *   8-connected labels numbered 1..K in raster order of each component's first pixel (int32).
 * ---------------------------------------------------------------------------------------------- */
size_t cs_ccl_workspace_bytes(int h, int w);
/* mask: uint8 [h,w] (non-zero = foreground); labels: int32 [h,w]; n_labels: device int (may be NULL). */
int cs_ccl_postprocess(const uint8_t* mask, int h, int w, int hole_area, int min_size,
                       int32_t* labels, int32_t* n_labels, void* workspace, size_t workspace_bytes, void* stream);
/* Plain labelling of a mask: connectivity 4 or 8, same numbering rule (scipy.ndimage.label semantics). */
int cs_ccl_label(const uint8_t* mask, int h, int w, int connectivity,
                 int32_t* labels, int32_t* n_labels, void* workspace, size_t workspace_bytes, void* stream);

/* The same post-process with the slide cut into strips of rows, one rank (GPU) per strip - SURVEY.md 8(e) third row, "strip-local
 * union-find + one seam-merge pass"; replaces the single-device call of baseline/predict.py:113 when one slide is spread over GPUs.
 * Labels are bit-identical to cs_ccl_postprocess on the whole mask.  Per rank, with two all-gathers by the caller in between:
 *   cs_ccl_strip_local    mask_rows: uint8 [ext_h][w] = the strip's rows plus cs_ccl_strip_halo(hole_area, min_size) rows above and
 *                         below it (fewer only at the image's own first / last row); the strip is rows [strip_row0, strip_row0 + strip_h)
 *                         of those.  Filters the rows, labels the strip locally, writes the edge record (cs_ccl_strip_edge_ints(w) int32:
 *                         per pixel of the strip's first and last row, which local component it belongs to).
 *   -- all-gather the records: all_records int32 [world][cs_ccl_strip_edge_ints(w)] --
 *   cs_ccl_strip_resolve  the seam union-find (identical on every rank), final numbering of this strip; edge_ids: int32 [2][w] out (ids
 *                         of the strip's first / last row), n_labels: device int, components of the whole slide (may be NULL).
 *   -- all-gather the edge ids: all_edge_ids int32 [world][2][w] --
 *   cs_ccl_strip_finish   labels: int32 [strip_h][w] out.
 * workspace: cs_ccl_workspace_bytes(ext_h, w), the SAME buffer for the three calls (it carries the strip's forest between them);
 * seam_workspace: cs_ccl_seam_workspace_bytes(world, w).  strip_h = 0 (more ranks than rows) is allowed. */
int cs_ccl_strip_halo(int hole_area, int min_size);
int cs_ccl_strip_edge_ints(int w);
size_t cs_ccl_seam_workspace_bytes(int world, int w);
int cs_ccl_strip_local(const uint8_t* mask_rows, int ext_h, int w, int strip_row0, int strip_h, int hole_area, int min_size,
                       int32_t* edge_record, void* workspace, size_t workspace_bytes, void* stream);
int cs_ccl_strip_resolve(const int32_t* all_records, int world, int rank, int ext_h, int w, int strip_h, int32_t* edge_ids,
                         int32_t* n_labels, void* workspace, size_t workspace_bytes, void* seam_workspace, size_t seam_workspace_bytes,
                         void* stream);
int cs_ccl_strip_finish(const int32_t* all_edge_ids, int world, int rank, int ext_h, int w, int strip_h, int32_t* labels,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Instance map -> the 3-class training label (0 background, 1 interior, 2 boundary).
 * Replaces create_interior_map (/root/reference/data/pre_process_3class.py:31-56):
 *   boundary = binary_dilation(find_boundaries(inst, mode='inner'), disk(1));
 *   interior = remove_small_objects(~boundary & inst > 0, min_size = 16);  out[interior] = 1; out[boundary] = 2.
 * inst: int32 [h][w]; out: uint8 [h][w]; workspace: cs_ccl_workspace_bytes(h, w). */
int cs_interior_map(const int32_t* inst, int h, int w, int min_size, uint8_t* out,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training loss  -  replaces `loss_function = monai.losses.DiceCELoss(softmax=True)` and
 * `loss = loss_function(outputs, labels_onehot); loss.backward()` (baseline/model_training_3class.py:234, :256-260):
 *   p = softmax(logits, dim 1); dice_nc = 1 - (2 sum(p t) + nr) / (sum t + sum p + dr), mean over n * classes (include_background);
 *   ce = mean_pixels -log p[argmax t];  loss = lambda_dice * dice + lambda_ce * ce     (MONAI defaults: nr = dr = 1e-5, lambdas 1).
 * logits / target: fp32 [n][classes][h][w] addressed by 4 element strides (n, class, h, w) - the engine's NHWC logits are read
 * in place; classes in 2..4.  cs_dice_ce_forward zeroes and fills `workspace` (cs_dice_ce_workspace_bytes: fp64 sums) and writes
 * the scalar loss to the device float `loss_out`; cs_dice_ce_backward reads the same workspace and writes
 * grad_loss * d loss / d logits (grad_loss: device float or NULL = 1) with the logits' strides.
 * ---------------------------------------------------------------------------------------------- */
size_t cs_dice_ce_workspace_bytes(int n, int classes);
int cs_dice_ce_forward(const float* logits, const int64_t* logit_strides, const float* target, const int64_t* target_strides, int n, int classes,
                       int h, int w, float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce, float* loss_out, void* workspace,
                       size_t workspace_bytes, void* stream);
int cs_dice_ce_backward(const float* logits, const int64_t* logit_strides, const float* target, const int64_t* target_strides, int n, int classes,
                        int h, int w, float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce, const float* grad_loss,
                        const void* workspace, size_t workspace_bytes, float* grad_logits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser  -  replaces `optimizer = torch.optim.AdamW(model.parameters(), args.initial_lr)` ... `optimizer.step()`
 * (baseline/model_training_3class.py:237, :262) with torch's update rule (amsgrad / maximize off) for ALL tensors in one launch:
 *   p *= 1 - lr wd;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * table: device int64 [n_tensors][3] = (device pointer of the parameter tensor, float offset of the tensor in the flat buffers,
 * element count), offsets ascending (cs_unet_grad_offset); grad: the flat gradient bucket cs_unet_backward wrote (`total` =
 * cs_unet_grad_numel floats); exp_avg / exp_avg_sq: flat moment buffers of the same size (zero before step 1); step = 1, 2, ...
 * Limits: n_tensors <= 256 (the offset table is staged in LDS; the reference net has 63 tensors) - more is refused with an error.
 * ONE step count for all tensors: torch keeps a step per parameter, which differs only for a tensor whose first gradient arrives
 * after step 1 (its bias correction then uses the global count); cellseg_amd.AdamW refuses that case (the active set must not grow).
 * ---------------------------------------------------------------------------------------------- */
int cs_adamw_step(const int64_t* table, int n_tensors, int64_t total, const float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, void* stream);

/* The same update with every per-step quantity in device memory, so that the launch can be stream-captured and replayed:
 * hyper: device float[16] - [0..4] = lr, beta1, beta2, eps, weight_decay (the host rewrites them when a scheduler changes them),
 * [8..10] scratch (decay, step size, 1 / sqrt(bias correction 2) of the step being applied); step: device int32, the number of steps
 * applied so far (zero before the first): the call increments it and applies that step. */
int cs_adamw_step_dev(const int64_t* table, int n_tensors, int64_t total, const float* grad, float* exp_avg, float* exp_avg_sq,
                      float* hyper, int32_t* step, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Captured launch sequences (hipGraph).  The loop body of baseline/model_training_3class.py:250-262 - forward, DiceCE, backward,
 * AdamW, weight re-pack - is ~230 launches of a few microseconds each at the reference's width and batch 8: issued one by one the
 * host, not the GPU, sets the pace.  Every entry point of this library that only enqueues work (cs_unet_forward, cs_unet_backward
 * with its side-stream weight gradients, cs_dice_ce_*, cs_adamw_step_dev, cs_unet_repack, cs_unet_readback) may be called between
 * cs_graph_begin and cs_graph_end on buffers that stay allocated; cs_graph_launch then replays the whole sequence as ONE submission.
 * Not capturable: anything that uploads host tables (cs_unet_load_*), per-kernel profiling (cs_prof_enable).
 * ---------------------------------------------------------------------------------------------- */
typedef struct cs_graph cs_graph;
int cs_graph_begin(void* stream);                 /* hipStreamBeginCapture on `stream` */
int cs_graph_end(void* stream, cs_graph** out);   /* end the capture and instantiate the executable graph */
int cs_graph_abort(void* stream);                 /* end a capture whose recording failed, discarding it */
int cs_graph_launch(cs_graph* g, void* stream);
int cs_graph_num_nodes(const cs_graph* g);
int cs_graph_destroy(cs_graph* g);

/* ------------------------------------------------------------------------------------------------
 * F1 harness pieces  -  the O(H*W) parts of baseline/compute_metric.py (the metric of record):
 *   cs_label_max               max label of one or two int32 maps (out2[0], out2[1]) - `x.max()`, `y.max()` (:60)
 *   cs_label_overlap           _label_overlap (:39-70): overlap[(n_gt+1) x (n_seg+1)] pixel-pair counts (uint32)
 *   cs_remove_boundary_cells   remove_boundary_cells (:124-133): drop labels touching the `border`-pixel frame, then
 *                              relabel_sequential (order preserving); n_labels >= max label of the input
 * IoU, Hungarian matching and the F1 formula operate on the small overlap matrix and stay on the host.
 * ---------------------------------------------------------------------------------------------- */
int cs_label_max(const int32_t* a, const int32_t* b /* may be NULL */, int64_t n, int32_t* out2, void* stream);
int cs_label_overlap(const int32_t* gt, const int32_t* seg, int64_t n, int32_t n_gt, int32_t n_seg, uint32_t* overlap, void* stream);
size_t cs_remove_boundary_workspace_bytes(int32_t n_labels);
int cs_remove_boundary_cells(const int32_t* labels, int h, int w, int border, int32_t n_labels, int32_t* out, int32_t* n_out,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-kernel timing (HIP events on the launch stream) for bench.py's roofline line.
 * cs_prof_enable(1) brackets every kernel launch with events; cs_prof_collect synchronises and
 * returns per-kernel-id totals.  Disabled (default) costs nothing.  The state is kept per device: the calls act on the
 * device that is current when they are made.
 * ---------------------------------------------------------------------------------------------- */
enum { CS_PROF_MAX_KERNELS = 48 };
int cs_prof_enable(int on);
int cs_prof_is_enabled(void);
int cs_prof_reset(void);
/* names_out: CS_PROF_MAX_KERNELS x 48 chars; ms_out / launches_out / flops_out / bytes_out: CS_PROF_MAX_KERNELS.
 * flops/bytes are ALGORITHMIC figures accumulated per launch (DESIGN.md "Roofline accounting"). Returns #ids. */
int cs_prof_collect(char* names_out_host, double* ms_out_host, int64_t* launches_out_host,
                    double* flops_out_host, double* bytes_out_host);

#ifdef __cplusplus
}
#endif
#endif /* CELLSEG_HIP_H */
