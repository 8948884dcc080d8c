// Host-side engine + C-ABI (include/cellseg_hip.h) for the MI355X 3-class U-Net hot path.
//
// cs_unet mirrors monai.networks.nets.UNet as the reference instantiates it
// (baseline/predict.py:45-52, baseline/model_training_3class.py:201-208): same topology, same state-dict
// key names, weights re-packed once into MFMA fragment order.  The forward schedule fuses each
// ResidualUnit's strided unit0 conv with its residual conv (one pass over the input), keeps every
// InstanceNorm + PReLU lazy (applied by the consumer) and materialises only ResidualUnit outputs.
#include <stdarg.h>
#include <string.h>

#include <functional>
#include <string>
#include <vector>

#include "kernels.h"
#include "pack_jobs.h"

namespace cs {

// ---------------------------------------------------------------------------------------------------
// error + profiling plumbing
// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static const char* kNames[K_NUM_IDS] = {
    "conv3x3_s1_mfma", "conv3x3_s2_mfma", "conv1x1_mfma", "conv_first_mfma", "convT3x3s2_mfma", "convT_top_mfma",
    "conv_last_valu", "norm_finalize", "norm_act_add", "swi_finalize", "pack_weights", "ccl_init", "ccl_merge",
    "ccl_compress", "ccl_sizes", "ccl_filter", "ccl_count", "ccl_scan", "ccl_rank", "ccl_relabel", "memset", "norm_bwd",
    "colsum", "wgrad_mfma", "top_bwd_valu", "normalize_input", "f1_metric", "wgrad_unpack", "conv3x3_s1_wino", "dice_ce_loss", "conv3x3_s1_wino4",
    "label_deflate", "conv3x3_s1_fp8", "conv3x3_s2_fp8", "conv1x1_fp8", "convT3x3s2_fp8", "ccl_seam"};
static_assert(K_NUM_IDS <= CS_PROF_MAX_KERNELS, "cs_prof_collect: the caller's arrays hold CS_PROF_MAX_KERNELS entries");
const char* kernel_name(int id) { return (id >= 0 && id < K_NUM_IDS) ? kNames[id] : "?"; }

struct ProfRecord {
    int id;
    hipEvent_t start, stop;
};
// Profiling state is kept PER DEVICE (two handles on two devices in one process do not share event pools or totals); the
// entry points below act on the device that is current when they are called.
struct ProfState {
    bool on = false;
    std::vector<ProfRecord> records;
    std::vector<hipEvent_t> pool;
    double flops[K_NUM_IDS] = {}, bytes[K_NUM_IDS] = {}, ms[K_NUM_IDS] = {};
    int64_t launches[K_NUM_IDS] = {};
};
static ProfState g_prof[64];       // one slot per HIP device index (a node has 8; 64 leaves room for partitioned modes)
static bool g_prof_any = false;                 // fast path: nobody has ever enabled profiling in this process
static ProfState& prof_state() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return g_prof[dev >= 0 && dev < 64 ? dev : 0];
}
bool prof_enabled() { return g_prof_any && prof_state().on; }

static hipEvent_t get_event(ProfState& ps) {
    if (!ps.pool.empty()) {
        hipEvent_t e = ps.pool.back();
        ps.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(int id, hipStream_t s, double flops, double bytes) : slot(-1), stream(s) {
    if (!g_prof_any) return;
    ProfState& ps = prof_state();
    if (!ps.on) return;
    ProfRecord r{id, get_event(ps), get_event(ps)};
    (void)hipEventRecord(r.start, s);
    ps.records.push_back(r);
    slot = (int)ps.records.size() - 1;
    ps.flops[id] += flops;
    ps.bytes[id] += bytes;
    ps.launches[id] += 1;
}
ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(prof_state().records[slot].stop, stream);
}

static void prof_drain(ProfState& ps) {
    for (auto& r : ps.records) {
        (void)hipEventSynchronize(r.stop);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) ps.ms[r.id] += ms;
        ps.pool.push_back(r.start);
        ps.pool.push_back(r.stop);
    }
    ps.records.clear();
}

// ---------------------------------------------------------------------------------------------------
// bump allocator over the caller's workspace; base == nullptr measures only
// ---------------------------------------------------------------------------------------------------
struct Bump {
    char* base;
    size_t off = 0, cap;
    Bump(void* b, size_t c) : base((char*)b), cap(c) {}
    template <typename T>
    T* take(size_t count) {
        const size_t bytes = (count * sizeof(T) + 255) / 256 * 256;
        T* p = base ? (T*)(base + off) : nullptr;
        off += bytes;
        return p;
    }
    bool dry() const { return base == nullptr; }
    bool ok() const { return dry() || off <= cap; }
};

// ---------------------------------------------------------------------------------------------------
// network description
// ---------------------------------------------------------------------------------------------------
struct NormSite {          // one ADN: InstanceNorm (stats at run time) or eval BatchNorm (running stats) + PReLU/ReLU
    int c = 0;
    int p_slope = -1;                       // parameter index of the PReLU weight
    float* slope = nullptr;                 // PReLU weight (device) or nullptr for ReLU
    float *bn_w = nullptr, *bn_b = nullptr, *bn_m = nullptr, *bn_v = nullptr;
    int p_bn_w = -1, p_bn_b = -1;           // parameter indices of the BatchNorm affine weight / bias
};

struct ConvW {             // packed weights + bias of one (possibly fused) conv launch
    float* w = nullptr;    // packed (MFMA kernels) or PyTorch layout (small VALU kernels)
    float* wd = nullptr;   // the same weights packed for the DATA-GRADIENT launch (see kernels_bwd.hip header)
    float* ww = nullptr;   // 3x3 stride-1 layers: Winograd F(2x2,3x3) weights G g G^T in fragment order (kernels_conv_wino.hip)
    float* wwd = nullptr;  // ... and those of the layer's DATA-GRADIENT convolution (channel roles swapped, taps reversed)
    float* ww4 = nullptr;  // 3x3 stride-1 layers with >= WINO4_MIN_CIN input channels: Winograd F(4x4,3x3) weights (kernels_conv_wino4x.hip).
                           // INFERENCE only: a training step re-packs every weight, and packing 36 positions in fp64 costs more there
                           // (0.25 ms per step) than the kernel saves on training-sized launches - loads in training mode leave them stale
    bool ww4_fresh = false; // the F(4x4) weights correspond to the parameter's current value (else the F(2x2) kernel runs)
    float* wtw = nullptr;   // transposed convs: the 25 F(2,2) products' weights (kernels_convt_wino.hip) - an inference-only packing too
    bool wtw_fresh = false;
    unsigned short* wb = nullptr;   // 3 bf16 planes in 16x16x32 fragment order (split-bf16 forward kernels)
    unsigned short* wbd = nullptr;  // the same planes for the layer's DATA-GRADIENT launch (bf16 training): stride-1 conv (roles swapped, taps reversed),
                                    // fused stride-2 pair (as a transposed conv over [dy0 || dr]), transposed conv (as a stride-2 conv)
    unsigned char* w8 = nullptr;    // e4m3 weights in 16x16x128 fragment order (fp8 forward kernels)
    float* wscale = nullptr;        // their per-output-channel scales [cout]
    bool f8 = false;                // fp8 mode: THIS layer runs on the fp8 kernels (else on the bf16 kernels from `wb`): see fp8_layer()
    float* bias = nullptr;
    int cin = 0, cout = 0; // cout = total (fused) output channels
    int p_w[2] = {-1, -1}, p_b[2] = {-1, -1};   // parameter indices (slot 1 = fused residual conv)
};

enum DgradKind { DG_NONE = 0, DG_S1, DG_S2_AS_CONVT, DG_CONVT_AS_S2, DG_1X1 };

// fp8 mode ("mixed fp8 weights / bf16", BASELINE.json configs[4]): a layer takes the e4m3 kernels (K = 128 channels per matrix
// instruction) when both its channel counts reach CS_FP8_MIN_C (default 128); narrower layers - where most of a K = 128 step is
// padding and whose e4m3 rounding sits closest to the logits - run on the bf16 kernels of the bf16 datapath, reading bf16 tensors.
static int fp8_min_channels() {
    static const int v = [] { const char* e = getenv("CS_FP8_MIN_C"); return e ? atoi(e) : 128; }();
    return v;
}
static bool fp8_layer(int cin, int cout) { return cin >= fp8_min_channels() && cout >= fp8_min_channels(); }
// A TRANSPOSED conv may take the e4m3 kernel already when its output has CS_FP8_MIN_COUT_T channels (default: CS_FP8_MIN_C, i.e. off).
// Round-6 A/B with CS_FP8_MIN_COUT_T=64 (configs[4]'s 128 + 256 -> 64 level, 11.4 ms per slide on the bf16 kernel at 3.4 x its HBM floor;
// both inputs exist as e4m3 tensors anyway): slide 104.9 -> 101.1 ms, 32-base fp8 slide 11.0 -> 10.7 ms - but this is the e4m3 layer
// closest to the logits: instance F1 against the fp32 instances 0.996 -> 0.993, delta F1 vs truth pooled over 3 training seeds x 48 images
// -2.1e-4 +- 1.3e-4, ONE of the three nets at -1.2e-3 +- 2e-4 (profiles/configs4_accuracy_r06_t64.json) - outside north_star's 1e-3 for that
// net, so the default keeps the layer on the bf16 kernel.
static int fp8_min_cout_t() {
    static const int v = [] { const char* e = getenv("CS_FP8_MIN_COUT_T"); return e ? atoi(e) : fp8_min_channels(); }();
    return v;
}
static bool fp8_layer_t(int cin_min, int cout) { return cin_min >= fp8_min_channels() && cout >= (fp8_min_cout_t() < fp8_min_channels() ? fp8_min_cout_t() : fp8_min_channels()); }

struct DownUnit {          // ResidualUnit in the encoder / bottom
    int cin = 0, c = 0, stride = 2, subunits = 2;
    bool first = false;            // cin == in_channels (3): first-layer kernel
    bool res_conv = true;          // false: identity residual
    bool fused_res = true;         // residual conv fused into unit0 (strided 3x3 residual)
    ConvW unit0;                   // fused [unit0 || residual] when fused_res
    ConvW res1x1;                  // bottom layer: separate 1x1 residual
    std::vector<ConvW> units;      // subunits 1..
    std::vector<NormSite> norms;   // one per subunit
};

struct UpUnit {
    int cin_skip = 0, cin_sub = 0, cout = 0;
    bool top = false;
    ConvW convt;
    NormSite norm_t;
    ConvW unit0;
    NormSite norm_u;   // unused at top (conv only)
};

struct ParamEntry {
    std::string key;
    int rank = 0;
    int64_t shape[4] = {0, 0, 0, 0};
    std::function<int(const float*, hipStream_t)> load;
    bool loaded = false;
    float* raw = nullptr;                         // tensors the engine keeps verbatim (BatchNorm affine / running statistics, PReLU slope)
    bool has_dgrad = false, has_bf16 = false;     // tensor takes part in the data-gradient / split-bf16 packings
    bool dgrad_packed = false;
    int wb_format = 0;                            // low-precision packing present: 0 none, 1 split-bf16, 2 split-fp16, 3 fp8
    int64_t grad_off = 0;      // float offset of this parameter's gradient in the flat gradient buffer
    int64_t numel() const {
        int64_t n = 1;
        for (int i = 0; i < rank; ++i) n *= shape[i];
        return n;
    }
};

struct NormRt {   // run-time state of one norm site for the current forward
    double* stats = nullptr;
    float2* tab = nullptr;
    float2* xtab = nullptr;   // BatchNorm with batch statistics: (rstd, -mean*rstd) for the backward pass
};

// cs_unet_trace: one tensor of the forward inside the caller's workspace (kind: 0 fp32, 1 bf16, 2 e4m3 activations [n, h, w, pitch];
// 3 float2 norm table [n, c] = (scale, shift) of act(norm(x)) = prelu(x * scale + shift))
struct TraceEntry {
    std::string name;
    int64_t off = 0;
    int n = 0, h = 0, w = 0, c = 0, pitch = 0, kind = 0;
};

// Pointers into the forward workspace recorded by the last forward; the backward pass replays them.
struct Tape {
    bool valid = false;
    int n = 0, h = 0, w = 0;
    WindowSrc src;
    std::vector<float*> E, X;                 // per encoder level: fused conv output, ResidualUnit output
    std::vector<std::vector<float*>> Y;       // per encoder level: subunit outputs (index 0 unused: unit0 lives in E)
    std::vector<std::vector<NormRt>> rt_down;
    std::vector<float*> Yb;                   // bottom subunit outputs
    float* Rb = nullptr;
    float* Sb = nullptr;                      // bottom ResidualUnit output
    std::vector<NormRt> rt_bottom;
    std::vector<float*> U, V, D;              // per decoder level: transposed-conv output, unit0 output, ResidualUnit output
    std::vector<NormRt> rt_upt, rt_upu;
    const float* logits = nullptr;
    bool h16 = false;                         // the activation tensors above (all but the 3-channel top level's U) hold bf16 elements
};

}  // namespace cs

using namespace cs;

struct cs_unet {
    int in_ch = 3, out_ch = 3, nl = 0, nru = 2, norm = CS_NORM_INSTANCE, act = CS_ACT_PRELU;
    std::vector<int> ch;
    std::vector<DownUnit> down;   // nl encoder units
    DownUnit bottom;
    std::vector<UpUnit> up;       // up[i] pairs with down[i]
    std::vector<ParamEntry> params;
    float* arena = nullptr;
    size_t arena_floats = 0;
    int64_t grad_floats = 0;
    int planes = 0;   // 0: fp32 MFMA kernels; 1..3: split-bf16 kernels with that many planes (3 = fp32-grade); 4: split-fp16 (fp32-grade);
                      // 5: fp8 weights + fp8-staged activations on the block-scaled matrix instruction (bf16 datapath)
    bool training = false;   // pack the data-gradient weight layouts on every parameter load
    bool batch_stats = false;   // BatchNorm networks: normalise with the statistics of the current batch and update the running ones
                                // (nn.BatchNorm2d in train mode); false = running statistics (eval mode)
    // sliding-window inference can spread its window chunks over side streams: the HBM-bound kernels of one chunk
    // (residual adds, 3-channel layers, statistics) then overlap the MFMA-bound convolutions of another
    int n_streams = 1;
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    Tape tape;
    // backward: the weight-gradient launches (off the critical chain dy -> data gradient -> norm backward -> dy) run on a side
    // stream, forked from the caller's stream by an event once their dy is complete
    hipStream_t bwd_side[2] = {nullptr, nullptr};
    hipEvent_t bwd_ev[48] = {};
    int device = -1;                 // the device the weight arena and the packing tables live on
    cs::PackCtx* pack = nullptr;     // per-handle weight-packing state (job tables allocated once, at creation)
    // cs_unet_bind_params: the re-pack of ALL parameters from fixed source tensors as a frozen job table (cs_unet_repack re-runs it),
    // and the copy of the BatchNorm running statistics back to the host's buffers (cs_unet_readback)
    cs::PackCtx* pack_bound = nullptr;
    cs::PackCtx* pack_readback = nullptr;
    int bound_planes = -1;
    bool bound_training = false;
    // cs_unet_trace: where every tensor of the last cs_unet_forward lives inside the caller's workspace (per-layer parity tests)
    bool trace = false;
    mutable std::vector<cs::TraceEntry> trace_log;
};

struct cs_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

namespace cs {

static int copy_param(cs_unet* net, float* dst, const float* src, size_t n, hipStream_t s) { return cs::pack_copy(net->pack, s, src, dst, n); }

struct ArenaPlan {   // two passes: count floats, then hand out pointers
    float* base = nullptr;
    size_t off = 0;
    float* take(size_t n) {
        n = (n + 63) / 64 * 64;
        float* p = base ? base + off : nullptr;
        off += n;
        return p;
    }
};

static int add_param(cs_unet* net, const std::string& key, std::initializer_list<int64_t> shape,
                     std::function<int(const float*, hipStream_t)> fn) {
    ParamEntry e;
    e.key = key;
    e.rank = (int)shape.size();
    int i = 0;
    for (auto v : shape) e.shape[i++] = v;
    e.load = std::move(fn);
    net->params.push_back(std::move(e));
    return (int)net->params.size() - 1;
}

static void add_norm_params(cs_unet* net, ArenaPlan& ar, const std::string& adn, NormSite& ns, int c) {
    ns.c = c;
    if (net->norm == CS_NORM_BATCH) {
        ns.bn_w = ar.take(c); ns.bn_b = ar.take(c); ns.bn_m = ar.take(c); ns.bn_v = ar.take(c);
        float *w = ns.bn_w, *b = ns.bn_b, *m = ns.bn_m, *v = ns.bn_v;
        ns.p_bn_w = add_param(net, adn + ".N.weight", {c}, [=](const float* d, hipStream_t s) { return copy_param(net, w, d, c, s); });
        ns.p_bn_b = add_param(net, adn + ".N.bias", {c}, [=](const float* d, hipStream_t s) { return copy_param(net, b, d, c, s); });
        const int pm = add_param(net, adn + ".N.running_mean", {c}, [=](const float* d, hipStream_t s) { return copy_param(net, m, d, c, s); });
        const int pv = add_param(net, adn + ".N.running_var", {c}, [=](const float* d, hipStream_t s) { return copy_param(net, v, d, c, s); });
        net->params[ns.p_bn_w].raw = w; net->params[ns.p_bn_b].raw = b; net->params[pm].raw = m; net->params[pv].raw = v;
    }
    if (net->act == CS_ACT_PRELU) {
        ns.slope = ar.take(1);
        float* sl = ns.slope;
        ns.p_slope = add_param(net, adn + ".A.weight", {1}, [=](const float* d, hipStream_t s) { return copy_param(net, sl, d, 1, s); });
        net->params[ns.p_slope].raw = sl;
    }
}

// conv weight + bias of a conv that writes channels [cout_off, cout_off + cout) of a fused launch.
// `dg` says how the same tensor is packed a second time (into cw.wd) for the data-gradient launch.
static void add_conv_params(cs_unet* net, const std::string& conv, ConvW& cw, int cout, int cin, int ksize, int cout_off,
                            bool transposed, int mode /*0 mfma, 1 first, 2 raw*/, int dg = DG_NONE, int slot = 0, int cin_split = 0) {
    float* wdst = cw.w;
    float* wd = cw.wd;
    float* ww = cw.ww;
    float* wwd = cw.wwd;
    float* ww4 = cw.ww4;
    bool* const ww4_fresh = &cw.ww4_fresh;
    float* wtw = cw.wtw;
    bool* const wtw_fresh = &cw.wtw_fresh;
    unsigned short* wb = cw.wb;
    unsigned short* wbd = cw.wbd;
    unsigned char* w8 = cw.w8;
    float* wsc = cw.wscale;
    const bool* const f8p = &cw.f8;
    float* bdst = cw.bias + cout_off;
    const int cout_total = cw.cout;
    // the data-gradient layout of the same tensor (channel axes swapped; see pack_conv_weights) is written by the same tile
    // kernel when the handle is in training mode: taps reversed for a stride-1 conv, second half of the fused pair offset by
    // its output-channel tiles
    auto dgrad_dst = [=]() -> float* { return (net->training && dg != DG_NONE) ? wd : nullptr; };
    const bool dg_flip = dg == DG_S1;
    const int dg_chunk_off = dg == DG_S2_AS_CONVT ? cout_off / 16 : 0;
    if (transposed) {
        cw.p_w[slot] = add_param(net, conv + ".weight", {cin, cout, ksize, ksize}, [=](const float* d, hipStream_t s) {
            if (mode == 2) return copy_param(net, wdst, d, (size_t)cin * cout * ksize * ksize, s);
            if (wb && ((net->planes > 0 && net->planes <= 4) || (net->planes == 5 && !*f8p)))
                if (int rc = pack_conv_weights_bf16(net->pack, s, d, true, cout, cin, ksize, wb, cout_off, cout_total, cin_split, net->planes == 4)) return rc;
            // bf16 training: the data gradient of a transposed conv is a stride-2 conv of dy with the SAME tensor read as conv weights
            // W[cin_t][cout_t] (no flip)
            if (wbd && net->training && dg == DG_CONVT_AS_S2 && net->planes == 1)
                if (int rc = pack_conv_weights_bf16(net->pack, s, d, false, cin, cout, ksize, wbd, 0, cin, 0, false)) return rc;
            if (w8 && net->planes == 5 && *f8p)
                if (int rc = pack_conv_weights_fp8(s, d, true, cout, cin, ksize, w8, wsc, cout_off, cout_total, cin_split)) return rc;
            if (wtw) {                                       // fp32 inference: the F(2,2) form of the transposed conv
                *wtw_fresh = convt_wino_enabled() && net->planes == 0 && !net->training && cout_off == 0 && cout == cout_total;
                if (*wtw_fresh)
                    if (int rc = pack_convt_wino_weights(net->pack, s, d, cout, cin, wtw)) return rc;
            }
            return pack_conv_weights(net->pack, s, d, true, cout, cin, ksize, wdst, cout_off, cout_total, dgrad_dst(), dg_flip, dg_chunk_off);
        });
    } else {
        cw.p_w[slot] = add_param(net, conv + ".weight", {cout, cin, ksize, ksize}, [=](const float* d, hipStream_t s) {
            if (mode == 2) return copy_param(net, wdst, d, (size_t)cin * cout * ksize * ksize, s);
            if (mode == 1) {
                if (wb && (net->planes == 5 || (net->planes >= 1 && net->planes <= 3)))   // every mode of the bf16 / fp8 upload families
                    if (int rc = pack_first_weights16(net->pack, s, d, cout, cin, wb, cout_off)) return rc;
                return pack_first_weights(net->pack, s, d, cout, cin, wdst, cout_off, cout_total);
            }
            if (wb && ((net->planes > 0 && net->planes <= 4) || (net->planes == 5 && !*f8p)))
                if (int rc = pack_conv_weights_bf16(net->pack, s, d, false, cout, cin, ksize, wb, cout_off, cout_total, 0, net->planes == 4)) return rc;
            // training on the 16-bit pipe (fp16x3 / bf16): the data-gradient convolution of a 3x3 stride-1 layer is a 3x3 stride-1
            // convolution with the channel roles swapped and the taps reversed - the same kernel family, its own weight planes
            if (wbd && net->training && dg == DG_S1 && net->planes == 1)
                if (int rc = pack_conv_weights_bf16(net->pack, s, d, true, cin, cout, ksize, wbd, 0, cin, 0, net->planes == 4, true)) return rc;
            // ... and of a (fused) stride-2 conv a transposed conv of [dy0 || dr] with the tensors read as transposed-conv weights
            // W[cin_t = cout][cout_t = cin] (no flip); the residual conv's tensor is the second input-channel source
            if (wbd && net->training && dg == DG_S2_AS_CONVT && net->planes == 1)
                if (int rc = pack_conv_weights_bf16(net->pack, s, d, true, cin, cout, ksize, wbd, 0, cin, 0, false, false, cout_off ? (cout + 31) / 32 : 0))
                    return rc;
            if (w8 && net->planes == 5 && *f8p)
                if (int rc = pack_conv_weights_fp8(s, d, false, cout, cin, ksize, w8, wsc, cout_off, cout_total, 0)) return rc;
            // 3x3 stride-1 layers: the tile kernel also writes the Winograd weights (forward; data gradient in training mode)
            if (ww4) {
                *ww4_fresh = wino_level() >= 2 && net->planes == 0 && !net->training;
                if (*ww4_fresh)
                    if (int rc = pack_wino4_weights(net->pack, s, d, cout, cin, ww4)) return rc;
            }
            float* const wino_f = (ww && wino_enabled()) ? ww : nullptr;
            float* const wino_d = (wwd && wino_enabled() && net->training && dg == DG_S1) ? wwd : nullptr;
            return pack_conv_weights(net->pack, s, d, false, cout, cin, ksize, wdst, cout_off, cout_total, dgrad_dst(), dg_flip, dg_chunk_off,
                                     wino_f, wino_d);
        });
    }
    net->params[cw.p_w[slot]].has_dgrad = dg != DG_NONE;
    net->params[cw.p_w[slot]].has_bf16 = wb != nullptr && (mode == 0 || mode == 1);
    cw.p_b[slot] = add_param(net, conv + ".bias", {cout}, [=](const float* d, hipStream_t s) { return copy_param(net, bdst, d, cout, s); });
}

static void build_down(cs_unet* net, ArenaPlan& ar, DownUnit& u, const std::string& pfx, int cin, int c, int stride) {
    u.cin = cin; u.c = c; u.stride = stride; u.subunits = net->nru;
    u.first = (cin == net->in_ch) && stride == 2 && cin < 16;
    u.res_conv = (stride != 1) || (cin != c);
    u.fused_res = u.res_conv && stride != 1;
    u.norms.resize(u.subunits);
    u.units.resize(u.subunits > 1 ? u.subunits - 1 : 0);
    // unit0 (+ fused residual)
    u.unit0.cin = cin;
    u.unit0.cout = u.fused_res ? 2 * c : c;
    u.unit0.w = ar.take(u.first ? packed_first_floats(u.unit0.cout, cin) : packed_conv_floats(u.unit0.cout, cin, 3));
    u.unit0.bias = ar.take(u.unit0.cout);
    // data gradient of unit0 (+ fused residual): a transposed conv over [dy0 || dr] for the strided units, a flipped
    // stride-1 conv for the bottom unit; the first layer needs none (the image has no gradient)
    const int dg0 = u.first ? DG_NONE : (stride == 2 ? DG_S2_AS_CONVT : DG_S1);
    if (dg0 != DG_NONE) u.unit0.wd = ar.take(packed_conv_floats(cin, u.unit0.cout, 3));
    if (stride == 1 && !u.first && cin % 16 == 0 && c % 32 == 0) u.unit0.ww = ar.take(packed_wino_floats(c, cin));
    if (stride == 1 && !u.first && cin >= WINO4_MIN_CIN && cin % 32 == 0 && c % 32 == 0) u.unit0.ww4 = ar.take(packed_wino4_floats(c, cin));
    if (stride == 1 && !u.first && c % 16 == 0 && cin % 32 == 0) u.unit0.wwd = ar.take(packed_wino_floats(cin, c));
    if (!u.first) u.unit0.wb = (unsigned short*)ar.take((packed_bf16_elems(u.unit0.cout, cin, 0, 3) + 1) / 2);
    if (stride == 1 && !u.first) u.unit0.wbd = (unsigned short*)ar.take((packed_bf16_elems(cin, c, 0, 3) + 1) / 2);
    if (stride == 2 && !u.first && u.fused_res) u.unit0.wbd = (unsigned short*)ar.take((packed_bf16_elems(cin, c, c, 3) + 1) / 2);
    if (u.first) u.unit0.wb = (unsigned short*)ar.take((size_t)(u.unit0.cout / 16) * 512 / 2);      // one bf16 K-step (bf16 / fp8 datapaths)
    if (!u.first) { u.unit0.w8 = (unsigned char*)ar.take((packed_fp8_bytes(u.unit0.cout, cin, 0, 3) + 3) / 4); u.unit0.wscale = ar.take(u.unit0.cout); }
    u.unit0.f8 = !u.first && fp8_layer(cin, c);
    add_conv_params(net, pfx + ".conv.unit0.conv", u.unit0, c, cin, 3, 0, false, u.first ? 1 : 0, dg0, 0);
    add_norm_params(net, ar, pfx + ".conv.unit0.adn", u.norms[0], c);
    for (int su = 1; su < u.subunits; ++su) {
        ConvW& cw = u.units[su - 1];
        cw.cin = c; cw.cout = c;
        cw.w = ar.take(packed_conv_floats(c, c, 3));
        cw.bias = ar.take(c);
        cw.wd = ar.take(packed_conv_floats(c, c, 3));
        if (c % 32 == 0) cw.ww = ar.take(packed_wino_floats(c, c));
        if (c >= WINO4_MIN_CIN && c % 32 == 0) cw.ww4 = ar.take(packed_wino4_floats(c, c));
        if (c % 32 == 0) cw.wwd = ar.take(packed_wino_floats(c, c));
        cw.wb = (unsigned short*)ar.take((packed_bf16_elems(c, c, 0, 3) + 1) / 2);
        cw.wbd = (unsigned short*)ar.take((packed_bf16_elems(c, c, 0, 3) + 1) / 2);
        cw.w8 = (unsigned char*)ar.take((packed_fp8_bytes(c, c, 0, 3) + 3) / 4);
        cw.wscale = ar.take(c);
        cw.f8 = fp8_layer(c, c);
        const std::string un = pfx + ".conv.unit" + std::to_string(su);
        add_conv_params(net, un + ".conv", cw, c, c, 3, 0, false, 0, DG_S1, 0);
        add_norm_params(net, ar, un + ".adn", u.norms[su], c);
    }
    if (u.fused_res) {
        add_conv_params(net, pfx + ".residual", u.unit0, c, cin, 3, c, false, u.first ? 1 : 0, dg0, 1);
    } else if (u.res_conv) {
        u.res1x1.cin = cin; u.res1x1.cout = c;
        u.res1x1.w = ar.take(packed_conv_floats(c, cin, 1));
        u.res1x1.wd = ar.take(packed_conv_floats(cin, c, 1));
        u.res1x1.wb = (unsigned short*)ar.take((packed_bf16_elems(c, cin, 0, 1) + 1) / 2);
        u.res1x1.w8 = (unsigned char*)ar.take((packed_fp8_bytes(c, cin, 0, 1) + 3) / 4);
        u.res1x1.wscale = ar.take(c);
        u.res1x1.f8 = fp8_layer(cin, c);
        u.res1x1.bias = ar.take(c);
        add_conv_params(net, pfx + ".residual", u.res1x1, c, cin, 1, 0, false, 0, DG_1X1, 0);
    }
}

static void build_up(cs_unet* net, ArenaPlan& ar, UpUnit& u, const std::string& pfx, int cin_skip, int cin_sub, int cout, bool top) {
    u.cin_skip = cin_skip; u.cin_sub = cin_sub; u.cout = cout; u.top = top;
    const int cin = cin_skip + cin_sub;
    const bool small = (cout % 16) != 0;
    u.convt.cin = cin; u.convt.cout = cout;
    u.convt.w = ar.take(small ? (size_t)cin * cout * 9 : packed_conv_floats(cout, cin, 3));
    u.convt.bias = ar.take(cout);
    if (!small) u.convt.wd = ar.take(packed_conv_floats(cin, cout, 3));
    if (!small) u.convt.wb = (unsigned short*)ar.take((packed_bf16_elems(cout, cin_skip, cin_sub, 3) + 1) / 2);
    if (!small) u.convt.wbd = (unsigned short*)ar.take((packed_bf16_elems(cin, cout, 0, 3) + 1) / 2);
    if (!small) { u.convt.w8 = (unsigned char*)ar.take((packed_fp8_bytes(cout, cin_skip, cin_sub, 3) + 3) / 4); u.convt.wscale = ar.take(cout); }
    u.convt.f8 = !small && fp8_layer_t(cin_skip < cin_sub ? cin_skip : cin_sub, cout);
    if (!small && cout % 32 == 0 && cin_skip % 16 == 0 && cin_sub % 16 == 0) u.convt.wtw = ar.take(packed_convt_wino_floats(cout, cin));
    add_conv_params(net, pfx + ".0.conv", u.convt, cout, cin, 3, 0, true, small ? 2 : 0, small ? DG_NONE : DG_CONVT_AS_S2, 0, cin_skip);
    add_norm_params(net, ar, pfx + ".0.adn", u.norm_t, cout);
    u.unit0.cin = cout; u.unit0.cout = cout;
    u.unit0.w = ar.take(small ? (size_t)cout * cout * 9 : packed_conv_floats(cout, cout, 3));
    u.unit0.bias = ar.take(cout);
    if (!small) u.unit0.wd = ar.take(packed_conv_floats(cout, cout, 3));
    if (!small && cout % 32 == 0) u.unit0.ww = ar.take(packed_wino_floats(cout, cout));
    if (!small && cout >= WINO4_MIN_CIN && cout % 32 == 0) u.unit0.ww4 = ar.take(packed_wino4_floats(cout, cout));
    if (!small && cout % 32 == 0) u.unit0.wwd = ar.take(packed_wino_floats(cout, cout));
    if (!small) u.unit0.wb = (unsigned short*)ar.take((packed_bf16_elems(cout, cout, 0, 3) + 1) / 2);
    if (!small) u.unit0.wbd = (unsigned short*)ar.take((packed_bf16_elems(cout, cout, 0, 3) + 1) / 2);
    if (!small) { u.unit0.w8 = (unsigned char*)ar.take((packed_fp8_bytes(cout, cout, 0, 3) + 3) / 4); u.unit0.wscale = ar.take(cout); }
    u.unit0.f8 = !small && fp8_layer(cout, cout);
    add_conv_params(net, pfx + ".1.conv.unit0.conv", u.unit0, cout, cout, 3, 0, false, small ? 2 : 0, small ? DG_NONE : DG_S1, 0);
    if (!top) add_norm_params(net, ar, pfx + ".1.conv.unit0.adn", u.norm_u, cout);
}

static void build_net(cs_unet* net, ArenaPlan& ar) {
    net->params.clear();
    net->down.assign(net->nl, DownUnit());
    net->up.assign(net->nl, UpUnit());
    std::string pfx = "model";
    // registration order follows PyTorch: down (".0"), inner block (".1.submodule"), up (".2")
    std::vector<std::string> prefixes(net->nl);
    for (int d = 0; d < net->nl; ++d) {
        prefixes[d] = pfx;
        build_down(net, ar, net->down[d], pfx + ".0", d == 0 ? net->in_ch : net->ch[d - 1], net->ch[d], 2);
        pfx += ".1.submodule";
    }
    build_down(net, ar, net->bottom, pfx, net->ch[net->nl - 1], net->ch[net->nl], 1);
    for (int d = net->nl - 1; d >= 0; --d) {
        const int cin_sub = (d == net->nl - 1) ? net->ch[d + 1] : net->ch[d];
        build_up(net, ar, net->up[d], prefixes[d] + ".2", net->ch[d], cin_sub, d == 0 ? net->out_ch : net->ch[d - 1], d == 0);
    }
}

// ---------------------------------------------------------------------------------------------------
// forward over n windows (roi_h x roi_w); `src` says where the pixels come from
// ---------------------------------------------------------------------------------------------------
static int finish_norm(hipStream_t s, const cs_unet* net, const NormSite& ns, NormRt& rt, int n, int c_total, int count, bool dry) {
    if (dry) return CS_OK;
    if (net->norm == CS_NORM_BATCH && net->batch_stats)      // nn.BatchNorm2d, train mode: momentum 0.1, eps 1e-5 (torch defaults)
        return launch_bn_train_finalize(s, rt.stats, ns.bn_w, ns.bn_b, ns.bn_m, ns.bn_v, rt.tab, rt.xtab, n, ns.c, c_total, count, 1e-5f, 0.1f);
    if (net->norm == CS_NORM_BATCH) return launch_bn_table(s, ns.bn_w, ns.bn_b, ns.bn_m, ns.bn_v, rt.tab, n, ns.c, c_total, 1e-5f);
    return launch_norm_finalize(s, rt.stats, rt.tab, n, c_total, count, 1e-5f, /*snap_constant=*/!net->training);
}

static ConvSrc lazy_src(const float* ptr, int c, int pitch, const NormRt& rt, const NormSite& ns, int tab_c, bool h16 = false) {
    ConvSrc s;
    s.ptr = ptr; s.c = c; s.pitch = pitch; s.tab = rt.tab; s.slope = ns.slope; s.tab_c = tab_c; s.tab_off = 0; s.h16 = h16;
    return s;
}
static ConvSrc raw_src(const float* ptr, int c, int pitch, bool h16 = false) {
    ConvSrc s;
    s.ptr = ptr; s.c = c; s.pitch = pitch; s.h16 = h16;
    return s;
}
// element offset into an activation tensor whose elements are fp32 or (bf16 datapath) 16 bits wide
static const float* act_off(const float* p, size_t elems, bool h16) {
    return h16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p) + elems) : p + elems;
}

// forward conv on the matrix cores: fp32 MFMA or split-bf16, by net->planes
static int conv_fwd(hipStream_t s, const cs_unet* net, int ksize, int stride, const ConvSrc& a, const ConvW& cw, float* out,
                    int pitch_out, int cout, double* stats, int n, int hin, int win) {
    if (net->planes == 5 && cw.w8 && cw.f8) {
        CS_REQUIRE(a.q8, "conv(fp8): the layer's input was not materialised as e4m3");
        return launch_conv_fp8(s, ksize, stride, a, ConvSrc(), cw.w8, cw.wscale, cw.bias, out, pitch_out, cout, stats, n, hin, win);
    }
    if (net->planes > 0 && net->planes <= 5 && cw.wb) {         // (fp8 mode: the layers below CS_FP8_MIN_C channels, on the bf16 kernels)
        CS_REQUIRE(!a.q8, "conv(bf16): a layer on the bf16 kernels met an e4m3 input");
        return launch_conv_bf16s(s, ksize, stride, net->planes == 5 ? 1 : net->planes, a, ConvSrc(), cw.wb, cw.bias, out, pitch_out, cout, stats, n, hin,
                                 win, a.h16);
    }
    CS_REQUIRE(!a.h16 && !a.q8, "conv: a layer without low-precision weights met bf16 / e4m3 activations");
    // fp32, wide layers: Winograd F(4x4,3x3) - 2.25 multiplies per output pixel and channel pair instead of F(2x2)'s 4 (CS_WINO=1: F(2x2) everywhere)
    if (net->planes == 0 && cw.ww4 && cw.ww4_fresh && ksize == 3 && stride == 1 && wino_level() >= 2 && wino4_fits(a.c, cout, hin, win) &&
        wino_fits(hin, win, a.pitch))
        return launch_conv_wino4(s, a, cw.ww4, cw.bias, out, pitch_out, cout, stats, n, hin, win);
    if (net->planes == 0 && cw.ww && ksize == 3 && stride == 1 && wino_enabled() && wino_fits(hin, win, a.pitch))   // fp32: Winograd F(2x2,3x3), 2.25x fewer multiplies
        return launch_conv_wino(s, a, cw.ww, cw.bias, out, pitch_out, cout, stats, n, hin, win);
    return launch_conv_mfma(s, ksize, stride, a, ConvSrc(), cw.w, cw.bias, out, pitch_out, cout, stats, n, hin, win);
}

// data gradient of a 3x3 stride-1 conv = a 3x3 stride-1 conv of dy with the channel-swapped, tap-reversed weights: Winograd form
// when the layer has one (same kernel as the forward), else the direct kernel on cw.wd
static int dgrad_s1(hipStream_t s, const ConvW& cw, const float* dy, int c_dy, float* out, int c_out, int n, int h, int w,
                    const ConvOutExtra* ex = nullptr, int planes16 = 0, const float* dy_amax = nullptr) {
    ConvSrc src;
    src.ptr = dy; src.c = c_dy; src.pitch = c_dy;
    // training on the 16-bit pipe: split-fp16 (planes16 == 4; dy under its dynamic power-of-two scale) or plain bf16 (1)
    if (planes16 && cw.wbd && !(ex && (ex->out2 || ex->addend2)))
        return launch_conv_bf16s(s, 3, 1, planes16, src, ConvSrc(), cw.wbd, nullptr, out, c_out, c_out, nullptr, n, h, w, false,
                                 planes16 == 4 ? dy_amax : nullptr, ex);
    if (cw.wwd && wino_enabled() && !(ex && ex->out2) && wino_fits(h, w, c_dy))
        return launch_conv_wino(s, src, cw.wwd, nullptr, out, c_out, c_out, nullptr, n, h, w, ex);
    return launch_conv_mfma(s, 3, 1, src, ConvSrc(), cw.wd, nullptr, out, c_out, c_out, nullptr, n, h, w, ex);
}

#define CS_TRY(expr)            \
    do {                        \
        int rc_ = (expr);       \
        if (rc_ != CS_OK) return rc_; \
    } while (0)

// bf16 activations in HBM?  The plain-bf16 and fp8 inference modes, and - round 5 - training in the bf16 mode (the tape then holds bf16
// tensors; CS_TAPE16=0 keeps the fp32 tape of rounds 2-4).  One predicate for the forward, its sizing pass and the pixel-record format.
static bool act16_mode(const cs_unet* net) {
    static const bool tape16_env = !(getenv("CS_TAPE16") && atoi(getenv("CS_TAPE16")) == 0);
    return (net->planes == 1 || net->planes == 5) && (!net->training || (net->planes == 1 && tape16_env));
}

// Runs (or, with a dry Bump, only sizes) the forward.  Output: logits NHWC [n, h, w, out_ch].
static int unet_forward_impl(const cs_unet* net, const WindowSrc& src, int n, int h, int w, float* logits, Bump& ws, hipStream_t s,
                             Tape* tape = nullptr) {
    const bool dry = ws.dry();
    const int nl = net->nl;
    CS_REQUIRE(h % (1 << nl) == 0 && w % (1 << nl) == 0 && h > 0 && w > 0, "unet: window %d x %d must be a multiple of %d", h, w,
               1 << nl);
    const bool inst = net->norm == CS_NORM_INSTANCE || net->batch_stats;   // statistics of the running batch are needed
    const bool want_xtab = net->norm == CS_NORM_BATCH && net->batch_stats;
    // bf16 datapath (plain-bf16 mode): every activation tensor between the first conv and the 3-channel top layers is bf16 in HBM - half
    // the traffic of every HBM-bound kernel and of the conv inputs / outputs.  Training in the bf16 mode (BASELINE configs[1]) records
    // the same bf16 tensors on the tape (round 5: the norm-backward and weight-gradient kernels read bf16 activations; CS_TAPE16=0 keeps
    // the fp32 tape of rounds 2-4); the fp16x3 / fp32 training modes keep fp32 activations.
    const bool a16 = act16_mode(net);
    CS_REQUIRE(!(net->planes == 5 && net->training), "unet: the fp8 mode is an inference mode (cs_unet_set_training(0))");
    auto take_act = [&](size_t count) -> float* {
        return a16 ? reinterpret_cast<float*>(ws.take<unsigned short>(count)) : ws.take<float>(count);
    };
    const bool tracing = net->trace && !dry;
    if (tracing) net->trace_log.clear();
    auto tap = [&](const std::string& name, const void* ptr, int hh, int ww, int c, int pitch, int kind) {
        if (!tracing || !ptr) return;
        TraceEntry e;
        e.name = name; e.off = (const char*)ptr - ws.base; e.n = n; e.h = hh; e.w = ww; e.c = c; e.pitch = pitch; e.kind = kind;
        net->trace_log.push_back(e);
    };
    const int k16 = a16 ? 1 : 0;
    if (tape) {
        tape->valid = false;
        tape->E.assign(nl, nullptr); tape->Y.assign(nl, {}); tape->U.assign(nl, nullptr); tape->V.assign(nl, nullptr);
        tape->D.assign(nl, nullptr);
    }
    // all statistics of one forward live in one arena so a single memset clears them
    const size_t stats_mark = ws.off;
    std::vector<std::pair<NormRt*, int>> pending;   // (site, channels in its stats/tab)
    auto new_norm = [&](NormRt& rt, int c_total) {
        rt.stats = inst ? ws.take<double>((size_t)n * c_total * 2) : nullptr;
        (void)pending;
    };
    // pre-allocate statistics: walk the net once
    std::vector<std::vector<NormRt>> rt_down(nl);
    std::vector<NormRt> rt_bottom(net->bottom.subunits);
    std::vector<NormRt> rt_upt(nl), rt_upu(nl);
    for (int d = 0; d < nl; ++d) {
        rt_down[d].resize(net->down[d].subunits);
        for (int su = 0; su < net->down[d].subunits; ++su) new_norm(rt_down[d][su], su == 0 ? net->down[d].unit0.cout : net->down[d].c);
    }
    for (int su = 0; su < net->bottom.subunits; ++su) new_norm(rt_bottom[su], net->bottom.c);
    for (int d = 0; d < nl; ++d) {
        new_norm(rt_upt[d], net->up[d].cout);
        if (!net->up[d].top) new_norm(rt_upu[d], net->up[d].cout);
    }
    const size_t stats_bytes = ws.off - stats_mark;
    if (!dry && inst) {
        if (!ws.ok()) { set_error("unet: workspace too small"); return CS_ERR_WORKSPACE; }
        ProfScope prof(K_MEMSET, s, 0, (double)stats_bytes);
        CS_TRY(launch_zero(s, ws.base + stats_mark, stats_bytes));
    }
    auto new_tab = [&](NormRt& rt, int c_total) {
        rt.tab = ws.take<float2>((size_t)n * c_total);
        rt.xtab = want_xtab ? ws.take<float2>((size_t)n * c_total) : nullptr;
    };

    // fp8 mode: conv outputs stay bf16 (raw, pre-norm); every tensor a conv READS is materialised once as activated e4m3
    // (launch_act_quant for the lazily normalised ones, the residual add writes its sum as e4m3), so the fp8 convs stage plain bytes
    const bool q8 = a16 && net->planes == 5;
    auto take_q8 = [&](size_t count) -> unsigned char* { return q8 ? ws.take<unsigned char>(count) : nullptr; };
    auto raw8_src = [&](const unsigned char* ptr, int c) {
        ConvSrc r;
        r.ptr = reinterpret_cast<const float*>(ptr); r.c = c; r.pitch = c; r.q8 = true;
        return r;
    };

    // ---- encoder ----
    std::vector<float*> X(nl);   // materialised ResidualUnit outputs (skip connections)
    std::vector<unsigned char*> Xq(nl, nullptr);
    const float* prev = nullptr;
    const unsigned char* prevq = nullptr;
    int ph = h, pw = w, pc = net->in_ch;
    for (int d = 0; d < nl; ++d) {
        const DownUnit& u = net->down[d];
        const int oh = ph / 2, ow = pw / 2, c = u.c, c0 = u.unit0.cout;
        // fp32 inference / bf16 datapath with pixel records: the first unit's residual branch is recomputed by the sum launch (below),
        // E then holds y0 alone
        static const int fuse_first_env = getenv("CS_FUSE_FIRST") ? atoi(getenv("CS_FUSE_FIRST")) : 1;
        const bool use_rec = u.first && !tape && src.rec && src.rec16 == a16 && pc == 3;
        const bool fuse_first_shape = fuse_first_env && u.first && !tape && u.subunits >= 2 && c0 == 2 * c && (c == 16 || c == 32 || c == 64);
        // (the sizing pass may stand for a training forward: it takes c0)
        const int ec = !dry && fuse_first_shape && (!a16 || use_rec) && !(q8 && ((d + 1 < nl ? net->down[d + 1].unit0.f8 : net->bottom.unit0.f8) || net->up[d].convt.f8))
                           ? c : c0;                            // channels (= pixel pitch) of E
        float* E = take_act((size_t)n * oh * ow * ec);          // [y0 || r], or y0
        std::vector<float*> Y(u.subunits, nullptr);
        for (int su = 1; su < u.subunits; ++su) Y[su] = take_act((size_t)n * oh * ow * c);
        // X[d] as bf16 / fp32: everywhere except the fp8 mode, where each consumer (the next level's unit0, this level's transposed conv)
        // reads the form its kernel family takes: bf16 for the bf16 kernels and the 3-channel top layer, e4m3 for the fp8 kernels
        const bool next8 = q8 && (d + 1 < nl ? net->down[d + 1].unit0.f8 : net->bottom.unit0.f8);
        const bool skip8 = q8 && net->up[d].convt.f8;
        X[d] = (!q8 || d == 0 || !next8 || !skip8) ? take_act((size_t)n * oh * ow * c) : nullptr;
        Xq[d] = (next8 || skip8) ? take_q8((size_t)n * oh * ow * c) : nullptr;
        bool any8 = false;
        for (int su = 1; su < u.subunits; ++su) any8 = any8 || u.units[su - 1].f8;
        unsigned char* Aq = (q8 && any8) ? take_q8((size_t)n * oh * ow * c) : nullptr;   // activated input of the next subunit
        for (int su = 0; su < u.subunits; ++su) new_tab(rt_down[d][su], su == 0 ? c0 : c);
        if (!ws.ok()) { set_error("unet: workspace too small"); return CS_ERR_WORKSPACE; }
        // the residual branch of the first unit (a 3 -> c conv of the image, 27 multiplies per output) is not written by the unit0 launch
        // and read back by norm_act_add: a second first-layer launch computes it from the image and adds act(norm(Y)) in its epilogue
        // (one c-channel tensor less written and read; CS_FUSE_FIRST=0 restores the [y0 || r] launch).  With pixel records from the
        // window pass (swi_run_windows) both launches run on the record kernels of kernels_first.hip (fp32 and the bf16 datapath; at
        // most 64 channels per launch)
        const bool fuse_first = ec == c && c0 != c;
        if (!dry) {
            if (u.first) {
                if (use_rec && ec <= 64)
                    CS_TRY(launch_conv_first_rec(s, src, u.unit0.w, u.unit0.bias, E, ec, ec, rt_down[d][0].stats, n, c, c0, 0));
                else if (a16 && u.unit0.wb)
                    CS_TRY(launch_conv_first16(s, src, pc, u.unit0.wb, u.unit0.bias, E, c0, c0, rt_down[d][0].stats, n, c));
                else if (fuse_first)
                    CS_TRY(launch_conv_first(s, 2, src, pc, u.unit0.w, u.unit0.bias, E, ec, c, rt_down[d][0].stats, n, false, c, c0, 0));
                else
                    CS_TRY(launch_conv_first(s, 2, src, pc, u.unit0.w, u.unit0.bias, E, c0, c0, rt_down[d][0].stats, n, a16, c));   // statistics for the unit0 half only
            } else {
                CS_TRY(conv_fwd(s, net, 3, 2, (q8 && u.unit0.f8) ? raw8_src(prevq, pc) : raw_src(prev, pc, pc, a16), u.unit0, E, c0, c0, rt_down[d][0].stats, n, ph, pw));
            }
            CS_TRY(finish_norm(s, net, u.norms[0], rt_down[d][0], n, c0, oh * ow, dry));
            const float* cur = E;
            int cur_pitch = u.first ? ec : c0, cur_tabc = c0;
            for (int su = 1; su < u.subunits; ++su) {
                const ConvSrc lz = lazy_src(cur, c, cur_pitch, rt_down[d][su - 1], u.norms[su - 1], cur_tabc, a16);
                const bool in8 = q8 && u.units[su - 1].f8;
                if (in8) CS_TRY(launch_act_quant(s, lz, Aq, c, (int64_t)oh * ow, n));
                CS_TRY(conv_fwd(s, net, 3, 1, in8 ? raw8_src(Aq, c) : lz, u.units[su - 1], Y[su], c, c, rt_down[d][su].stats, n, oh, ow));
                CS_TRY(finish_norm(s, net, u.norms[su], rt_down[d][su], n, c, oh * ow, dry));
                cur = Y[su];
                cur_pitch = c;
                cur_tabc = c;
            }
            const int last = u.subunits - 1;
            if (fuse_first) {
                const ConvSrc ly = lazy_src(cur, c, cur_pitch, rt_down[d][last], u.norms[last], cur_tabc, a16);
                if (use_rec) CS_TRY(launch_conv_first_rec(s, src, u.unit0.w, u.unit0.bias, X[d], c, c, nullptr, n, 0, c0, c, &ly));
                else CS_TRY(launch_conv_first(s, 2, src, pc, u.unit0.w, u.unit0.bias, X[d], c, c, nullptr, n, false, 0, c0, c, &ly));
            } else
            CS_TRY(launch_norm_act_add(s, lazy_src(cur, c, cur_pitch, rt_down[d][last], u.norms[last], cur_tabc, a16),
                                       raw_src(act_off(E, c, a16), c, c0, a16), X[d], c, (int64_t)oh * ow, n, a16, Xq[d], c));
        }
        if (tracing) {
            const std::string pfx = "down" + std::to_string(d) + ".";
            tap(pfx + "E", E, oh, ow, ec, ec, k16);
            for (int su = 1; su < u.subunits; ++su) tap(pfx + "Y" + std::to_string(su), Y[su], oh, ow, c, c, k16);
            tap(pfx + "X", X[d], oh, ow, c, c, k16);
            tap(pfx + "Xq", Xq[d], oh, ow, c, c, 2);
            tap(pfx + "Aq", Aq, oh, ow, c, c, 2);
            for (int su = 0; su < u.subunits; ++su) tap(pfx + "tab" + std::to_string(su), rt_down[d][su].tab, 1, 1, su == 0 ? c0 : c, su == 0 ? c0 : c, 3);
        }
        prev = X[d];
        prevq = Xq[d];
        ph = oh; pw = ow; pc = c;
        if (tape) { tape->E[d] = E; tape->Y[d] = Y; }
    }
    // fp32 inference: the residual sum of a decoder level (D[d] = act(norm(V)) + act(norm(U))) has ONE
    // consumer, the transposed conv of the level above.  When that is the MFMA kernel (every level but the 3-channel top one) it reads
    // the two tensors itself (a SUM source) and the norm_act_add pass - two reads and one write of the tensor - is not launched.
    // Training keeps the sums (the tape needs them).  CS_FUSE_SUM=0 materialises them as before.
    static const int fuse_env = getenv("CS_FUSE_SUM") ? atoi(getenv("CS_FUSE_SUM")) : 1;
    // fp32 path, bf16 datapath and - for the levels whose consumer runs on a bf16 kernel - the mixed fp8 mode
    const bool fuse_sum = fuse_env && !tape && (a16 ? (net->planes == 1 || net->planes == 5) : net->planes == 0);
    ConvSrc sub;                                                 // "sub" input of the next transposed conv
    // ---- bottom ----
    float* S = nullptr;   // "sub" input of the next up layer
    unsigned char* Sq = nullptr;
    {
        const DownUnit& u = net->bottom;
        const int c = u.c;
        std::vector<float*> Y(u.subunits);
        for (int su = 0; su < u.subunits; ++su) Y[su] = take_act((size_t)n * ph * pw * c);
        float* R = u.res_conv ? take_act((size_t)n * ph * pw * c) : nullptr;
        const bool up8 = q8 && net->up[nl - 1].convt.f8;               // the consumer of the bottom unit's sum
        S = !up8 ? take_act((size_t)n * ph * pw * c) : nullptr;        // bf16 form where the consumer is a bf16 kernel (or the 3-channel top layer)
        Sq = up8 ? take_q8((size_t)n * ph * pw * c) : nullptr;
        bool any8 = false;
        for (int su = 1; su < u.subunits; ++su) any8 = any8 || u.units[su - 1].f8;
        unsigned char* Aq = (q8 && any8) ? take_q8((size_t)n * ph * pw * c) : nullptr;
        for (int su = 0; su < u.subunits; ++su) new_tab(rt_bottom[su], c);
        if (!ws.ok()) { set_error("unet: workspace too small"); return CS_ERR_WORKSPACE; }
        CS_REQUIRE(!q8 || u.res_conv, "unet(fp8): an identity residual at the bottom is not supported");
        if (!dry) {
            const ConvSrc in = (q8 && u.unit0.f8) ? raw8_src(prevq, pc) : raw_src(prev, pc, pc, a16);      // (res1x1 has the same channel counts: same family)
            CS_TRY(conv_fwd(s, net, 3, 1, in, u.unit0, Y[0], c, c, rt_bottom[0].stats, n, ph, pw));
            CS_TRY(finish_norm(s, net, u.norms[0], rt_bottom[0], n, c, ph * pw, dry));
            if (u.res_conv)
                CS_TRY(conv_fwd(s, net, 1, 1, in, u.res1x1, R, c, c, nullptr, n, ph, pw));
            for (int su = 1; su < u.subunits; ++su) {
                const ConvSrc lz = lazy_src(Y[su - 1], c, c, rt_bottom[su - 1], u.norms[su - 1], c, a16);
                const bool in8 = q8 && u.units[su - 1].f8;
                if (in8) CS_TRY(launch_act_quant(s, lz, Aq, c, (int64_t)ph * pw, n));
                CS_TRY(conv_fwd(s, net, 3, 1, in8 ? raw8_src(Aq, c) : lz, u.units[su - 1], Y[su], c, c, rt_bottom[su].stats, n, ph, pw));
                CS_TRY(finish_norm(s, net, u.norms[su], rt_bottom[su], n, c, ph * pw, dry));
            }
            const int last = u.subunits - 1;
            const ConvSrc ly = lazy_src(Y[last], c, c, rt_bottom[last], u.norms[last], c, a16);
            // (the bottom unit's sum act(norm(Y)) + R is materialised: handed to the level above as a SUM source it saved 0.27 ms of
            // norm_act_add per slide and cost 0.32 ms in that transposed conv, 32 of whose 48 input chunks then stage two tensors)
            CS_TRY(launch_norm_act_add(s, ly, u.res_conv ? raw_src(R, c, c, a16) : raw_src(prev, c, pc, a16), S, c, (int64_t)ph * pw, n,
                                       a16, Sq, c));
            sub = raw_src(S, c, c, a16);
        }
        if (tracing) {
            for (int su = 0; su < u.subunits; ++su) {
                tap("bottom.Y" + std::to_string(su), Y[su], ph, pw, c, c, k16);
                tap("bottom.tab" + std::to_string(su), rt_bottom[su].tab, 1, 1, c, c, 3);
            }
            tap("bottom.R", R, ph, pw, c, c, k16);
            tap("bottom.S", S, ph, pw, c, c, k16);
            tap("bottom.Sq", Sq, ph, pw, c, c, 2);
            tap("bottom.Aq", Aq, ph, pw, c, c, 2);
        }
        if (tape) { tape->Yb = Y; tape->Rb = R; tape->Sb = S; }
    }
    // ---- decoder ----
    int sc = net->bottom.c;
    for (int d = nl - 1; d >= 0; --d) {
        const UpUnit& u = net->up[d];
        const int oh = ph * 2, ow = pw * 2, c = u.cout;
        const bool u16 = a16 && !u.top;                           // the 3-channel top tensors stay fp32
        float* U = u16 ? take_act((size_t)n * oh * ow * c) : ws.take<float>((size_t)n * oh * ow * c);
        new_tab(rt_upt[d], c);
        float *V = nullptr, *D = nullptr;
        unsigned char *Dq = nullptr, *Uq = nullptr;
        if (!u.top) {
            V = take_act((size_t)n * oh * ow * c);
            const bool up8 = q8 && net->up[d - 1].convt.f8;               // the consumer of this level's sum: the transposed conv above
            D = !up8 ? take_act((size_t)n * oh * ow * c) : nullptr;       // fp8 mode: bf16 where a bf16 kernel (or the top layer) reads it
            Dq = up8 ? take_q8((size_t)n * oh * ow * c) : nullptr;
            Uq = (q8 && u.unit0.f8) ? take_q8((size_t)n * oh * ow * c) : nullptr;
            new_tab(rt_upu[d], c);
        }
        if (!ws.ok()) { set_error("unet: workspace too small"); return CS_ERR_WORKSPACE; }
        if (!dry) {
            const bool in8 = q8 && !u.top && u.convt.f8;
            const ConvSrc a = in8 ? raw8_src(Xq[d], u.cin_skip) : raw_src(X[d], u.cin_skip, u.cin_skip, a16);
            const ConvSrc b = in8 ? raw8_src(Sq, sc) : sub;
            if (c % 16 == 0 && net->planes == 5 && u.convt.w8 && u.convt.f8) {
                CS_TRY(launch_convt_fp8(s, a, b, u.convt.w8, u.convt.wscale, u.convt.bias, U, c, c, rt_upt[d].stats, n, ph, pw));
            } else if (c % 16 == 0 && net->planes > 0 && net->planes <= 5 && u.convt.wb) {
                CS_TRY(launch_convt_bf16s(s, net->planes == 5 ? 1 : net->planes, a, b, u.convt.wb, u.convt.bias, U, c, c, rt_upt[d].stats, n, ph, pw, a16));
            } else if (c % 16 == 0) {
                CS_REQUIRE(!a16, "convT: a layer without 16-bit weights met bf16 activations");
                if (net->planes == 0 && u.convt.wtw && u.convt.wtw_fresh && convt_wino_fits(a.c, b.c, c, ph, pw) && !a.tab && (b.ptr2 || !b.tab))
                    CS_TRY(launch_convt_wino(s, a, b, u.convt.wtw, u.convt.bias, U, c, c, rt_upt[d].stats, n, ph, pw));   // 25 instead of 36 multiplies per 2 x 2 pixels
                else
                    CS_TRY(launch_convt_mfma(s, a, b, u.convt.w, u.convt.bias, U, c, c, rt_upt[d].stats, n, ph, pw));
            } else {
                CS_TRY(launch_convt_small(s, a, b, u.convt.w, u.convt.bias, U, c, rt_upt[d].stats, n, ph, pw));
            }
            CS_TRY(finish_norm(s, net, u.norm_t, rt_upt[d], n, c, oh * ow, dry));
            const ConvSrc lu = lazy_src(U, c, c, rt_upt[d], u.norm_t, c, u16);
            if (u.top) {
                CS_TRY(launch_conv_last(s, lu, u.unit0.w, u.unit0.bias, logits, c, n, oh, ow));
            } else {
                const bool v8 = q8 && u.unit0.f8;
                if (v8) CS_TRY(launch_act_quant(s, lu, Uq, c, (int64_t)oh * ow, n));
                CS_TRY(conv_fwd(s, net, 3, 1, v8 ? raw8_src(Uq, c) : lu, u.unit0, V, c, c, rt_upu[d].stats, n, oh, ow));
                CS_TRY(finish_norm(s, net, u.norm_u, rt_upu[d], n, c, oh * ow, dry));
                const ConvSrc lv = lazy_src(V, c, c, rt_upu[d], u.norm_u, c, a16);
                // (consumers with 64-channel blocks are left alone: the second tensor costs that kernel variant a third of its occupancy -
                // measured on the 32-base slide: level 2 fused -0.47 ms, level 3 fused -0.06 ms, CS_FUSE_SUM=2 fuses every level)
                const int c_up = net->up[d - 1].cout;
                const bool fused_mid = fuse_sum && d >= 2 && c_up % 16 == 0 && rt_upt[d].tab && rt_upu[d].tab && !(q8 && net->up[d - 1].convt.f8) &&
                                       (a16 ? (c % 32 == 0 && net->up[d - 1].convt.wb != nullptr) : (c_up % 64 != 0 || fuse_env == 2));
                // ... and the 3-channel top layer: its GEMM + gather kernel stages act(norm(V)) + act(norm(U)) itself for the 32 channels
                // that come from below - fp32 path: measured -0.9 ms of norm_act_add for +0.3 ms in that kernel per slide; bf16 datapath /
                // mixed fp8 mode (round 5): the sum is rounded to bf16 once while it is staged, bitwise what norm_act_add16x8 would have stored
                const int cin_top = net->up[0].cin_skip + c;
                const bool fused_top = fuse_sum && d == 1 && (!a16 || c % 16 == 0) && net->up[0].top && rt_upt[d].tab && rt_upu[d].tab &&
                                       c % 4 == 0 && net->up[0].cin_skip == c && (cin_top == 64 || cin_top == 128);
                const bool fused = fused_mid || fused_top;
                if (fused) {
                    sub = lv;
                    sub.ptr2 = U; sub.tab2 = rt_upt[d].tab; sub.slope2 = u.norm_t.slope;
                    D = nullptr;                                 // never written: the consumer reads V and U itself
                    Dq = nullptr;
                } else {
                    CS_TRY(launch_norm_act_add(s, lv, lu, D, c, (int64_t)oh * ow, n, a16, Dq, c));
                    sub = raw_src(D, c, c, a16);
                }
            }
        }
        if (tracing) {
            const std::string pfx = "up" + std::to_string(d) + ".";
            tap(pfx + "U", U, oh, ow, c, c, u16 ? 1 : 0);
            tap(pfx + "tabt", rt_upt[d].tab, 1, 1, c, c, 3);
            tap(pfx + "V", V, oh, ow, c, c, k16);
            tap(pfx + "Uq", Uq, oh, ow, c, c, 2);
            tap(pfx + "tabu", rt_upu[d].tab, 1, 1, c, c, 3);
            tap(pfx + "D", D, oh, ow, c, c, k16);
            tap(pfx + "Dq", Dq, oh, ow, c, c, 2);
        }
        S = D;
        Sq = Dq;
        sc = c;
        ph = oh; pw = ow;
        if (tape) { tape->U[d] = U; tape->V[d] = V; tape->D[d] = D; }
    }
    if (tape) {
        tape->valid = !dry;
        tape->n = n; tape->h = h; tape->w = w; tape->src = src;
        tape->X = X;
        tape->rt_down = rt_down; tape->rt_bottom = rt_bottom; tape->rt_upt = rt_upt; tape->rt_upu = rt_upu;
        tape->logits = logits;
        tape->h16 = a16;
    }
    return CS_OK;
}

// ---------------------------------------------------------------------------------------------------
// backward over the batch recorded in net->tape.  `acc` = fp64 / fp32 accumulators (zeroed by the caller's memset),
// `buf` = gradient activations.  Gradients land in `flat` at params[i].grad_off, in PyTorch layouts.
// ---------------------------------------------------------------------------------------------------
static int unet_backward_impl(const cs_unet* net, const float* dlogits, Bump& acc, Bump& buf, float* flat, hipStream_t s,
                              const hipStream_t* side = nullptr, int n_side = 0, const hipEvent_t* side_ev = nullptr, int n_side_ev = 0) {
    const bool dry = buf.dry();
    const Tape& T = net->tape;
    const int nl = net->nl, n = T.n, h = T.h, w = T.w;
    const bool bn = net->norm == CS_NORM_BATCH;
    CS_REQUIRE(!bn || dry || T.rt_upt[0].xtab != nullptr,
               "backward: the recorded forward used running statistics (eval-mode BatchNorm); call cs_unet_set_batch_stats(1) first");
    CS_REQUIRE(dry || T.src.origins == nullptr, "backward: the recorded forward must be a plain batched forward");
    auto gptr = [&](int pidx) { return flat + net->params[pidx].grad_off; };
    auto oom = [&]() { set_error("backward: workspace too small"); return CS_ERR_WORKSPACE; };
    // weight gradients (+ their unpack passes) go to the side stream `sw`: wg_begin() makes it wait for everything issued on `s` so far
    // (the dy the launches read).  They only read finished tensors and write their own partial slabs / gradient slots, and the
    // workspace is a bump allocator (nothing is reused inside one backward), so the main chain never has to wait for them before
    // the join at the end.  HBM-bound norm-backward kernels of the main chain then run beside matrix-bound weight gradients.
    const bool side_on = side != nullptr && n_side > 0 && n_side_ev > 0 && !dry;
    hipStream_t sw = s;                        // stream of the current weight-gradient group (the groups alternate over the side streams)
    int side_used = 0, groups = 0;
    // bias gradients (column sums of a finished dy) are nobody's input: with side streams they wait for the next weight-gradient group and
    // run in front of it, off the main chain (5 x 20 us of the 1.1 ms backward chain of the batch-8 reference-width step)
    struct PendingColsum { const float* dy; int pitch, c; int64_t npix; float* partial; float* dst; };
    PendingColsum pending[8];
    int n_pending = 0;
    auto flush_colsums = [&](hipStream_t st) -> int {
        for (int i = 0; i < n_pending; ++i)
            CS_TRY(launch_colsum(st, pending[i].dy, pending[i].pitch, pending[i].c, pending[i].npix, pending[i].partial, pending[i].dst));
        n_pending = 0;
        return CS_OK;
    };
    auto wg_begin = [&]() -> int {
        if (!side_on) return CS_OK;
        sw = side[groups++ % n_side];
        const hipEvent_t ev = side_ev[side_used++ % n_side_ev];
        CS_HIP(hipEventRecord(ev, s));
        CS_HIP(hipStreamWaitEvent(sw, ev, 0));
        return flush_colsums(sw);
    };

    // fp64 accumulators that end up as fp32 gradients (slope gradients, the last conv's weight / bias sums) are handed over by ONE
    // launch at the end of the pass
    F64Jobs f64_jobs;
    auto f64_job = [&](const double* src, float* dst, int count, int slots) -> int {
        CS_REQUIRE(f64_jobs.n < (int)(sizeof(f64_jobs.j) / sizeof(f64_jobs.j[0])), "backward: more than %d fp64 hand-overs", f64_jobs.n);
        f64_jobs.j[f64_jobs.n++] = F64Job{src, dst, count, slots};
        return CS_OK;
    };
    // helpers -------------------------------------------------------------------------------------------
    // `before_norm`: the conv feeds an InstanceNorm, which removes any per-(image, channel) constant - the bias gradient is
    // analytically zero (autograd returns rounding noise there), so it is written as exact zeros instead of summing dy
    auto bias_grad = [&](const float* dy, int pitch, int c, int64_t npix_total, int pidx, bool before_norm = false) -> int {
        if (before_norm) return CS_OK;      // exact zeros: the flat gradient buffer was cleared by cs_unet_backward's memset
        float* partial = buf.take<float>((size_t)colsum_blocks(npix_total) * c);
        if (dry) return CS_OK;
        if (!buf.ok()) return oom();
        if (side_on && n_pending < (int)(sizeof(pending) / sizeof(pending[0]))) {
            pending[n_pending++] = PendingColsum{dy, pitch, c, npix_total, partial, gptr(pidx)};
            return CS_OK;
        }
        return launch_colsum(s, dy, pitch, c, npix_total, partial, gptr(pidx));
    };
    // Weight gradients of the 3x3 layers on the 16-bit matrix pipe (kernels_bwd16.hip) when the handle's precision asks for it:
    // 0 = split fp16, 3 terms (cs_unet_set_precision 4, fp32-grade), 1 = plain bf16 (precision 1: bf16 forward AND backward
    // arithmetic, fp32 master weights - BASELINE configs[1]), -1 = the fp32 kernels.  Gradient operands of the fp16 mode carry a
    // dynamic power-of-two scale taken from max|dy|, which the producer of dy maintains in a device float (`amax` slots below).
    const int bwd16 = net->planes == 4 ? 0 : net->planes == 1 ? 1 : -1;
    // Data gradients of the 3x3 stride-1 layers: plain bf16 in the bf16 mode (conv_bf16s on the layer's data-gradient planes: 0.72 ->
    // 0.35 ms per step); the fp16x3 mode keeps the fp32 Winograd form - measured: the split-fp16 kernels (3 terms) need 0.64 ms for
    // the same launches, no faster than Winograd's 16/36 of the fp32 multiplies, and their slope gradients came out 1.4e-4 off the
    // fp64 yardstick (20x torch-fp32's error: gradient tensors span many decades, elements far below the tensor maximum lose
    // relative precision in the second fp16 plane) - so fp32-grade training keeps fp32 data gradients.
    const int dg16 = bwd16 == 1 ? 1 : 0;
    // bias_pidx >= 0: `da` is also the gradient of that conv bias (the raw residual branch of a ResidualUnit) - its column sums are taken
    // by this pass (one more accumulator per channel) and handed over with the other fp64 sums at the end, instead of a column-sum pass
    // of their own on the weight-gradient stream (5 x 2 launches, 0.19 ms of that stream at 32-base x 32; CS_FUSE_BIAS=0: the old passes)
    static const bool fuse_bias_env = !(getenv("CS_FUSE_BIAS") && atoi(getenv("CS_FUSE_BIAS")) == 0);
    auto norm_bwd = [&](const NormSite& ns, const NormRt& rt, const ConvSrc& y, const float* da, int da_pitch, float* dy, int dy_pitch,
                        int64_t npix, float** amax_out = nullptr, int bias_pidx = -1) -> int {
        double* red = acc.take<double>((size_t)n * y.c * 2);
        double* dsl = acc.take<double>(NORM_DSLOPE_SLOTS);
        float* amax = (bwd16 == 0 && amax_out) ? acc.take<float>(AMAX_SLOTS) : nullptr;
        const bool ride = bias_pidx >= 0 && fuse_bias_env && y.c != 3;
        double* dasum = ride ? acc.take<double>((size_t)n * y.c) : nullptr;
        if (amax_out) *amax_out = amax;
        if (bias_pidx >= 0 && !ride) CS_TRY(bias_grad(da, da_pitch, y.c, (int64_t)n * npix, bias_pidx));
        if (dry) return CS_OK;
        CS_TRY(launch_norm_bwd(s, y, da, da_pitch, dy, dy_pitch, red, dsl, npix, n, bn ? rt.xtab : nullptr, bn,
                               bn && ns.p_bn_w >= 0 ? gptr(ns.p_bn_w) : nullptr, bn && ns.p_bn_b >= 0 ? gptr(ns.p_bn_b) : nullptr, amax, dasum));
        if (ride) CS_TRY(f64_job(dasum, gptr(bias_pidx), y.c, n));
        if (ns.p_slope >= 0) return f64_job(dsl, gptr(ns.p_slope), 1, NORM_DSLOPE_SLOTS);
        return CS_OK;
    };
    auto rsrc = [&](const float* p, int c, int pitch) { return raw_src(p, c, pitch); };
    // tensors of the tape: bf16 elements when the forward ran the bf16 datapath (Tape::h16), fp32 otherwise
    const bool t16 = T.h16;
    auto tsrc = [&](const float* p, int c, int pitch) { return raw_src(p, c, pitch, t16); };
    auto tlazy = [&](const float* p, int c, int pitch, const NormRt& rt, const NormSite& ns, int tab_c, bool h16) {
        return lazy_src(p, c, pitch, rt, ns, tab_c, h16);
    };
    // max |t| of a gradient tensor that no norm backward produced (the residual-branch gradients), on the weight-gradient stream
    auto amax_of = [&](hipStream_t st, const float* t, size_t numel, float** out) -> int {
        *out = bwd16 == 0 ? acc.take<float>(AMAX_SLOTS) : nullptr;
        if (dry || !*out) return CS_OK;
        return launch_amax(st, t, (int64_t)numel, *out);
    };
    // weight-gradient partial sums: [split_k][taps][rows][cols] in the (un-zeroed) buffer arena
    struct WG { float* part; int k; bool h16; };
    auto wg_alloc = [&](int a_c1, int a_c2, int b_c, int taps, int rows, int cols, int ho, int wo, int ks = 3, int st = 2) {
        WG g;
        g.h16 = bwd16 >= 0 && ks == 3 && b_c % 16 == 0 && a_c1 % 16 == 0 && (a_c2 <= 0 || a_c2 % 16 == 0);
        auto kof = [&](int a_c) { return g.h16 ? wgrad16_split_k(a_c, b_c, n, ho, wo) : wgrad_split_k(a_c, b_c, n, ho, wo, ks, st); };
        g.k = kof(a_c1);
        if (a_c2 > 0) { const int k2 = kof(a_c2); g.k = k2 < g.k ? k2 : g.k; }
        g.part = buf.take<float>((size_t)g.k * taps * rows * cols);
        return g;
    };
    // one 3x3 weight-gradient launch on whichever pipe `g` was sized for; a_amax / b_amax: the dynamic-scale slots of gradient operands
    // a_h16: the A operand is an ACTIVATION tensor of the bf16 tape (transposed convs: the activations are the A operand)
    auto wgrad3 = [&](hipStream_t st, const WG& g, int stride, const float* a, int a_c, const float* a_amax, const ConvSrc& b, const float* b_amax,
                      int co_total, int ci_total, int co_off, int ho, int wo, int hi, int wi, bool a_h16 = false) -> int {
        if (g.h16)
            return launch_wgrad16(st, bwd16, stride, a, a_c, a_c, a_amax, b, b_amax, g.part, g.k, co_total, ci_total, co_off, 0, n, ho, wo, hi, wi,
                                  a_h16);
        return launch_wgrad(st, 3, stride, a, a_c, a_c, b, nullptr, g.part, g.k, co_total, ci_total, co_off, 0, n, ho, wo, hi, wi, a_h16);
    };
    auto unpack3 = [&](hipStream_t st, const WG& g, int rows_total, int cols_total, int row_off, int rows, int cols, float* dst, const float* a_amax,
                       const float* b_amax) -> int {
        WgradUnscale us;
        us.a_amax = a_amax; us.b_amax = b_amax;
        return launch_unpack_wgrad(st, g.part, g.k, rows_total, cols_total, row_off, 0, rows, cols, 9, dst, (g.h16 && bwd16 == 0) ? &us : nullptr);
    };

    std::vector<float*> dX(nl, nullptr);
    float* dS = nullptr;       // gradient w.r.t. the "sub" input of the level above (D[d+1] or the bottom output)

    // ---- decoder, top level first ----
    for (int d = 0; d < nl; ++d) {
        const UpUnit& u = net->up[d];
        const int c = u.cout, cs = u.cin_skip, cb = u.cin_sub, cin_t = cs + cb;
        const int oh = h >> d, ow = w >> d, ph = oh / 2, pw = ow / 2;
        const float* Ssub = (d == nl - 1) ? T.Sb : T.D[d + 1];
        const ConvSrc lu = tlazy(T.U[d], c, c, T.rt_upt[d], u.norm_t, c, t16 && !u.top);      // (the 3-channel top tensor is fp32 in every mode)
        float* d_au = buf.take<float>((size_t)n * oh * ow * c);
        float* dU = buf.take<float>((size_t)n * oh * ow * c);
        dX[d] = buf.take<float>((size_t)n * ph * pw * cs);
        float* dSn = buf.take<float>((size_t)n * ph * pw * cb);
        const WG gt = wg_alloc(cs, cb, c, 9, cin_t, c, ph, pw);
        if (!buf.ok()) return oom();
        if (u.top) {
            double* dw_last = acc.take<double>(84);                   // [81 weights | 3 biases]
            float* last_part = buf.take<float>(conv_last_bwd_partial_floats(n, oh, ow));
            if (!buf.ok()) return oom();
            if (!dry) {
                CS_TRY(launch_conv_last_bwd(s, dlogits, lu, u.unit0.w, d_au, dw_last, last_part, n, oh, ow));
                CS_TRY(f64_job(dw_last, gptr(u.unit0.p_w[0]), c * c * 9, 1));
                CS_TRY(f64_job(dw_last + 81, gptr(u.unit0.p_b[0]), c, 1));
            }
        } else {
            float* dV = buf.take<float>((size_t)n * oh * ow * c);
            const WG gu = wg_alloc(c, 0, c, 9, c, c, oh, ow, 3, 1);
            if (!buf.ok()) return oom();
            float* ax_dV = nullptr;
            CS_TRY(norm_bwd(u.norm_u, T.rt_upu[d], tlazy(T.V[d], c, c, T.rt_upu[d], u.norm_u, c, t16), dS, c, dV, c, (int64_t)oh * ow, &ax_dV));
            CS_TRY(bias_grad(dV, c, c, (int64_t)n * oh * ow, u.unit0.p_b[0], true));
            if (!dry) {
                CS_TRY(wg_begin());
                CS_TRY(wgrad3(sw, gu, 1, dV, c, ax_dV, lu, nullptr, c, c, 0, oh, ow, oh, ow));
                CS_TRY(unpack3(sw, gu, c, c, 0, c, c, gptr(u.unit0.p_w[0]), ax_dV, nullptr));
                ConvOutExtra ex;
                ex.addend = dS; ex.pitch_add = c;          // identity residual of the up ResidualUnit
                CS_TRY(dgrad_s1(s, u.unit0, dV, c, d_au, c, n, oh, ow, &ex, dg16, ax_dV));
            }
        }
        float* ax_dU = nullptr;
        CS_TRY(norm_bwd(u.norm_t, T.rt_upt[d], lu, d_au, c, dU, c, (int64_t)oh * ow, u.top ? nullptr : &ax_dU));
        CS_TRY(bias_grad(dU, c, c, (int64_t)n * oh * ow, u.convt.p_b[0], true));
        if (!dry) {
            if (u.top) {
                CS_TRY(launch_convt_small_bwd_data(s, dU, u.convt.w, cin_t, 0, cs, dX[d], cs, false, n, ph, pw));
                CS_TRY(launch_convt_small_bwd_data(s, dU, u.convt.w, cin_t, cs, cb, dSn, cb, false, n, ph, pw));
                WgradNarrow nb;
                nb.ptr = dU; nb.c = c; nb.sn = (int64_t)oh * ow * c; nb.sc = 1; nb.sh = (int64_t)ow * c; nb.sw = c;
                CS_TRY(wg_begin());
                CS_TRY(launch_wgrad(sw, 3, 2, T.X[d], cs, cs, ConvSrc(), &nb, gt.part, gt.k, cin_t, c, 0, 0, n, ph, pw, oh, ow, t16));
                CS_TRY(launch_wgrad(sw, 3, 2, Ssub, cb, cb, ConvSrc(), &nb, gt.part, gt.k, cin_t, c, cs, 0, n, ph, pw, oh, ow, t16));
            } else {
                ConvOutExtra ex;
                ex.out2 = dSn; ex.pitch_out2 = cb; ex.split = cs;
                if (dg16 && u.convt.wbd)
                    CS_TRY(launch_conv_bf16s(s, 3, 2, dg16, rsrc(dU, c, c), ConvSrc(), u.convt.wbd, nullptr, dX[d], cs, cin_t, nullptr, n, oh, ow, false,
                                             nullptr, &ex));
                else
                    CS_TRY(launch_conv_mfma(s, 3, 2, rsrc(dU, c, c), ConvSrc(), u.convt.wd, nullptr, dX[d], cs, cin_t, nullptr, n, oh, ow, &ex));
                CS_TRY(wg_begin());
                // (transposed conv: the ACTIVATIONS are the A operand, the gradient dU the shifted B operand)
                CS_TRY(wgrad3(sw, gt, 2, T.X[d], cs, nullptr, rsrc(dU, c, c), ax_dU, cin_t, c, 0, ph, pw, oh, ow, t16));
                CS_TRY(wgrad3(sw, gt, 2, Ssub, cb, nullptr, rsrc(dU, c, c), ax_dU, cin_t, c, cs, ph, pw, oh, ow, t16));
            }
            CS_TRY(unpack3(sw, gt, cin_t, c, 0, cin_t, c, gptr(u.convt.p_w[0]), nullptr, u.top ? nullptr : ax_dU));
        }
        dS = dSn;
    }

    // ---- bottom ----
    {
        const DownUnit& u = net->bottom;
        const int c = u.c, cin = u.cin, ph = h >> nl, pw = w >> nl;
        const int64_t npix = (int64_t)ph * pw;
        const float* Xin = T.X[nl - 1];
        float* dXin = dX[nl - 1];
        if (u.res_conv) {
            const WG gr = wg_alloc(c, 0, cin, 1, c, cin, ph, pw, 1, 1);
            // (the 1x1 residual conv's bias gradient = column sums of dS: taken by the norm backward below, whose `da` dS is)
            if (!dry) {
                CS_TRY(wg_begin());
                CS_TRY(launch_wgrad(sw, 1, 1, dS, c, c, tsrc(Xin, cin, cin), nullptr, gr.part, gr.k, c, cin, 0, 0, n, ph, pw, ph, pw));
                CS_TRY(launch_unpack_wgrad(sw, gr.part, gr.k, c, cin, 0, 0, c, cin, 1, gptr(u.res1x1.p_w[0])));
            }
        }
        const float* da = dS;
        for (int su = u.subunits - 1; su >= 0; --su) {
            const ConvW& cw = su == 0 ? u.unit0 : u.units[su - 1];
            float* dY = buf.take<float>((size_t)n * npix * c);
            const WG ga = wg_alloc(c, 0, su == 0 ? cin : c, 9, c, su == 0 ? cin : c, ph, pw, 3, 1);
            float* da_new = su > 0 ? buf.take<float>((size_t)n * npix * c) : nullptr;
            if (!buf.ok()) return oom();
            float* ax_dY = nullptr;
            CS_TRY(norm_bwd(u.norms[su], T.rt_bottom[su], tlazy(T.Yb[su], c, c, T.rt_bottom[su], u.norms[su], c, t16), da, c, dY, c, npix, &ax_dY,
                            (su == u.subunits - 1 && u.res_conv) ? u.res1x1.p_b[0] : -1));
            CS_TRY(bias_grad(dY, c, c, n * npix, cw.p_b[0], true));
            if (dry) { da = da_new; continue; }
            if (su > 0) {
                const ConvSrc prev = tlazy(T.Yb[su - 1], c, c, T.rt_bottom[su - 1], u.norms[su - 1], c, t16);
                CS_TRY(wg_begin());
                CS_TRY(wgrad3(sw, ga, 1, dY, c, ax_dY, prev, nullptr, c, c, 0, ph, pw, ph, pw));
                CS_TRY(unpack3(sw, ga, c, c, 0, c, c, gptr(cw.p_w[0]), ax_dY, nullptr));
                CS_TRY(dgrad_s1(s, cw, dY, c, da_new, c, n, ph, pw, nullptr, dg16, ax_dY));
                da = da_new;
            } else {
                CS_TRY(wg_begin());
                CS_TRY(wgrad3(sw, ga, 1, dY, c, ax_dY, tsrc(Xin, cin, cin), nullptr, c, cin, 0, ph, pw, ph, pw));
                CS_TRY(unpack3(sw, ga, c, cin, 0, c, cin, gptr(cw.p_w[0]), ax_dY, nullptr));
                ConvOutExtra ex;
                ex.addend = dXin; ex.pitch_add = cin;
                CS_TRY(dgrad_s1(s, cw, dY, c, dXin, cin, n, ph, pw, &ex, dg16, ax_dY));
            }
        }
        if (!dry) {
            if (u.res_conv) {
                ConvOutExtra ex;
                ex.addend = dXin; ex.pitch_add = cin;
                CS_TRY(launch_conv_mfma(s, 1, 1, rsrc(dS, c, c), ConvSrc(), u.res1x1.wd, nullptr, dXin, cin, cin, nullptr, n, ph, pw, &ex));
            } else {
                CS_TRY(launch_norm_act_add(s, rsrc(dXin, cin, cin), rsrc(dS, c, c), dXin, cin, npix, n));
            }
        }
    }

    // ---- encoder, deepest level first ----
    for (int d = nl - 1; d >= 0; --d) {
        const DownUnit& u = net->down[d];
        const int c = u.c, c0 = u.unit0.cout, cin = u.cin;
        const int oh = h >> (d + 1), ow = w >> (d + 1), ph = oh * 2, pw = ow * 2;
        const int64_t npix = (int64_t)oh * ow;
        const float* dXd = dX[d];
        // residual conv bias (the residual branch is raw): column sums of dX[d] - taken by the first norm backward below, whose `da` it is
        const float* da = dXd;
        for (int su = u.subunits - 1; su >= 1; --su) {
            const ConvW& cw = u.units[su - 1];
            float* dY = buf.take<float>((size_t)n * npix * c);
            float* da_new = buf.take<float>((size_t)n * npix * c);
            const WG ga = wg_alloc(c, 0, c, 9, c, c, oh, ow, 3, 1);
            if (!buf.ok()) return oom();
            float* ax_dY = nullptr;
            CS_TRY(norm_bwd(u.norms[su], T.rt_down[d][su], tlazy(T.Y[d][su], c, c, T.rt_down[d][su], u.norms[su], c, t16), da, c, dY, c, npix, &ax_dY,
                            su == u.subunits - 1 ? u.unit0.p_b[1] : -1));
            CS_TRY(bias_grad(dY, c, c, n * npix, cw.p_b[0], true));
            if (!dry) {
                const ConvSrc prev = su - 1 == 0 ? tlazy(T.E[d], c, c0, T.rt_down[d][0], u.norms[0], c0, t16)
                                                 : tlazy(T.Y[d][su - 1], c, c, T.rt_down[d][su - 1], u.norms[su - 1], c, t16);
                CS_TRY(wg_begin());
                CS_TRY(wgrad3(sw, ga, 1, dY, c, ax_dY, prev, nullptr, c, c, 0, oh, ow, oh, ow));
                CS_TRY(unpack3(sw, ga, c, c, 0, c, c, gptr(cw.p_w[0]), ax_dY, nullptr));
                CS_TRY(dgrad_s1(s, cw, dY, c, da_new, c, n, oh, ow, nullptr, dg16, ax_dY));
            }
            da = da_new;
        }
        float* dY0 = buf.take<float>((size_t)n * npix * c);
        const WG gf = wg_alloc(c, 0, cin, 9, c0, cin, oh, ow);
        if (!buf.ok()) return oom();
        float *ax_dY0 = nullptr, *ax_dXd = nullptr;
        CS_TRY(norm_bwd(u.norms[0], T.rt_down[d][0], tlazy(T.E[d], c, c0, T.rt_down[d][0], u.norms[0], c0, t16), da, c, dY0, c, npix,
                        u.first ? nullptr : &ax_dY0, u.subunits == 1 ? u.unit0.p_b[1] : -1));
        CS_TRY(bias_grad(dY0, c, c, n * npix, u.unit0.p_b[0], true));
        if (!u.first && gf.h16) {                                  // the residual conv's gradient is the raw dX[d]: no norm backward measured it
            if (!dry) CS_TRY(wg_begin());
            CS_TRY(amax_of(sw, dXd, (size_t)n * npix * c, &ax_dXd));
        }
        if (dry) continue;
        if (!u.first) {
            const ConvSrc xin = tsrc(T.X[d - 1], cin, cin);
            if (!gf.h16) CS_TRY(wg_begin());
            CS_TRY(wgrad3(sw, gf, 2, dY0, c, ax_dY0, xin, nullptr, c0, cin, 0, oh, ow, ph, pw));
            CS_TRY(wgrad3(sw, gf, 2, dXd, c, ax_dXd, xin, nullptr, c0, cin, c, oh, ow, ph, pw));
            ConvOutExtra ex;
            ex.addend = dX[d - 1]; ex.pitch_add = cin;
            if (dg16 && u.unit0.wbd)
                CS_TRY(launch_convt_bf16s(s, dg16, rsrc(dY0, c, c), rsrc(dXd, c, c), u.unit0.wbd, nullptr, dX[d - 1], cin, cin, nullptr, n, oh, ow, false, &ex));
            else
                CS_TRY(launch_convt_mfma(s, rsrc(dY0, c, c), rsrc(dXd, c, c), u.unit0.wd, nullptr, dX[d - 1], cin, cin, nullptr, n, oh, ow, &ex));
        } else {
            WgradNarrow nb;
            nb.ptr = T.src.image; nb.c = cin; nb.sn = T.src.sn; nb.sc = T.src.sc; nb.sh = T.src.sh; nb.sw = T.src.sw;
            CS_TRY(wg_begin());
            CS_TRY(launch_wgrad(sw, 3, 2, dY0, c, c, ConvSrc(), &nb, gf.part, gf.k, c0, cin, 0, 0, n, oh, ow, ph, pw));
            CS_TRY(launch_wgrad(sw, 3, 2, dXd, c, c, ConvSrc(), &nb, gf.part, gf.k, c0, cin, c, 0, n, oh, ow, ph, pw));
        }
        {   // both halves of the pair in ONE launch: rows [0, c) -> unit0's weight gradient, rows [c, 2c) -> the residual conv's
            WgradUnscale us0, us1;
            us0.a_amax = ax_dY0; us1.a_amax = ax_dXd;
            const bool scaled = gf.h16 && bwd16 == 0;
            CS_TRY(launch_unpack_wgrad(sw, gf.part, gf.k, c0, cin, 0, 0, c, cin, 9, gptr(u.unit0.p_w[0]), scaled ? &us0 : nullptr,
                                       gptr(u.unit0.p_w[1]), scaled ? &us1 : nullptr));
        }
    }
    if (!dry) CS_TRY(flush_colsums(s));        // column sums no weight-gradient group picked up
    if (!dry) CS_TRY(launch_f64_jobs(s, f64_jobs));
    if (side_on) {                             // join: the caller's stream continues behind the last weight gradients
        for (int i = 0; i < n_side; ++i) {
            const hipEvent_t ev = side_ev[side_used++ % n_side_ev];
            CS_HIP(hipEventRecord(ev, side[i]));
            CS_HIP(hipStreamWaitEvent(s, ev, 0));
        }
    }
    return CS_OK;
}

// a handle's weight arena, packing tables and side streams live on the device that was current when it was created
static int check_device(const cs_unet* net) {
    int dev = -1;
    CS_HIP(hipGetDevice(&dev));
    CS_REQUIRE(dev == net->device, "this cs_unet handle belongs to device %d but device %d is current (hipSetDevice / torch.cuda.device)", net->device, dev);
    return CS_OK;
}

static int check_loaded(const cs_unet* net, bool for_backward = false) {
    for (const auto& p : net->params) {
        if (!p.loaded) {
            set_error("parameter '%s' was never loaded", p.key.c_str());
            return CS_ERR_NOT_LOADED;
        }
        if (net->planes > 0 && p.has_bf16 && p.wb_format != (net->planes == 5 ? 3 : net->planes == 4 ? 2 : 1)) {
            set_error("parameter '%s' was loaded under a different cs_unet_set_precision family: load it again", p.key.c_str());
            return CS_ERR_NOT_LOADED;
        }
        if (for_backward && p.has_dgrad && !p.dgrad_packed) {
            set_error("parameter '%s' was loaded before cs_unet_set_training(1): load it again", p.key.c_str());
            return CS_ERR_NOT_LOADED;
        }
    }
    return CS_OK;
}

__global__ void swi_origins_kernel(cs_swi_plan pl, int k_lo, int count, int2* out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const int ih = (k_lo + k) / pl.n_w, iw = (k_lo + k) % pl.n_w;
    int hs = ih * pl.interval_h, wsx = iw * pl.interval_w;
    hs = hs < pl.pad_h - pl.roi_h ? hs : pl.pad_h - pl.roi_h;
    wsx = wsx < pl.pad_w - pl.roi_w ? wsx : pl.pad_w - pl.roi_w;
    out[k] = make_int2(hs - pl.pad_top, wsx - pl.pad_left);   // image coordinates (negative inside the zero padding)
}

}  // namespace cs

// ===================================================================================================
// C-ABI
// ===================================================================================================
extern "C" {

const char* cs_last_error(void) { return cs::g_err; }
int cs_version(void) { return 100; }

int cs_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        cs::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return CS_ERR_HIP;
    }
    return n;
}

int cs_unet_create(cs_unet** out, int in_channels, int out_channels, const int* channels_host, int n_channels, int num_res_units,
                   int norm_kind, int act_kind) {
    CS_REQUIRE(out && channels_host, "unet_create: null argument");
    CS_REQUIRE(in_channels == 3 && out_channels == 3, "unet_create: only the reference's 3 -> 3 channel contract is supported");
    CS_REQUIRE(n_channels >= 2 && n_channels <= 8, "unet_create: n_channels %d out of range", n_channels);
    CS_REQUIRE(num_res_units >= 1 && num_res_units <= 4, "unet_create: num_res_units %d out of range", num_res_units);
    CS_REQUIRE(norm_kind == CS_NORM_INSTANCE || norm_kind == CS_NORM_BATCH, "unet_create: bad norm kind");
    CS_REQUIRE(act_kind == CS_ACT_PRELU || act_kind == CS_ACT_RELU, "unet_create: bad act kind");
    for (int i = 0; i < n_channels; ++i)
        CS_REQUIRE(channels_host[i] % 16 == 0 && channels_host[i] >= 16 && channels_host[i] <= 4096,
                   "unet_create: channels[%d]=%d must be a multiple of 16", i, channels_host[i]);
    CS_REQUIRE(2 * channels_host[0] <= 128, "unet_create: channels[0]=%d too wide for the first-layer kernel", channels_host[0]);
    cs_unet* net = new cs_unet();
    net->in_ch = in_channels; net->out_ch = out_channels; net->nl = n_channels - 1; net->nru = num_res_units;
    net->norm = norm_kind; net->act = act_kind;
    net->ch.assign(channels_host, channels_host + n_channels);
    ArenaPlan count;
    build_net(net, count);
    net->arena_floats = count.off;
    hipError_t e = hipMalloc((void**)&net->arena, net->arena_floats * sizeof(float));
    if (e != hipSuccess) {
        cs::set_error("hipMalloc(%zu) for the weight arena failed: %s", net->arena_floats * sizeof(float), hipGetErrorString(e));
        delete net;
        return CS_ERR_HIP;
    }
    (void)hipMemset(net->arena, 0, net->arena_floats * sizeof(float));
    ArenaPlan real;
    real.base = net->arena;
    build_net(net, real);
    if (real.off != net->arena_floats) {       // a build step whose allocations depend on pointers would overrun the arena silently
        cs::set_error("unet_create: the two planning passes disagree (%zu vs %zu floats)", real.off, net->arena_floats);
        cs_unet_destroy(net);
        return CS_ERR_INVALID;
    }
    int64_t off = 0;
    for (auto& p : net->params) {
        p.grad_off = off;
        off += (p.numel() + 3) / 4 * 4;          // keep every gradient 16-byte aligned inside the flat bucket
    }
    net->grad_floats = off;
    (void)hipGetDevice(&net->device);
    net->pack = cs::pack_ctx_create(4 * (int)net->params.size() + 16);
    if (!net->pack) {
        cs::set_error("unet_create: allocating the weight-packing tables failed");
        cs_unet_destroy(net);
        return CS_ERR_HIP;
    }
    *out = net;
    return CS_OK;
}

int cs_unet_set_precision(cs_unet* net, int planes) {
    CS_REQUIRE(net && planes >= 0 && planes <= 5, "set_precision: 0 (fp32 MFMA), 1..3 (split-bf16 planes), 4 (split-fp16, 3 terms) or 5 (fp8)");
    net->planes = planes;
    return CS_OK;
}

int cs_unet_load_begin(cs_unet* net) {
    CS_REQUIRE(net, "load_begin: null handle");
    cs::pack_batch_begin(net->pack);
    return CS_OK;
}

int cs_unet_load_end(cs_unet* net, void* stream) {
    CS_REQUIRE(net, "load_end: null handle");
    return cs::pack_batch_flush(net->pack, (hipStream_t)stream);
}

int cs_unet_set_training(cs_unet* net, int on) {
    CS_REQUIRE(net, "set_training: null handle");
    net->training = on != 0;
    return CS_OK;
}

int cs_unet_set_batch_stats(cs_unet* net, int on) {
    CS_REQUIRE(net, "set_batch_stats: null handle");
    CS_REQUIRE(!on || net->norm == CS_NORM_BATCH, "set_batch_stats: the network has no BatchNorm layers");
    net->batch_stats = on != 0;
    return CS_OK;
}

int cs_unet_read_param(cs_unet* net, const char* key_host, float* dst, int64_t numel, void* stream) {
    CS_REQUIRE(net && key_host && dst, "read_param: null argument");
    for (auto& p : net->params) {
        if (p.key == key_host) {
            CS_REQUIRE(p.raw != nullptr, "read_param: '%s' is kept in a packed layout only (readable: BatchNorm tensors, PReLU slopes)", key_host);
            CS_REQUIRE(numel == p.numel(), "read_param: '%s' holds %lld elements, got %lld", key_host, (long long)p.numel(), (long long)numel);
            return cs::pack_copy(net->pack, (hipStream_t)stream, p.raw, dst, (size_t)numel);
        }
    }
    cs::set_error("read_param: unknown key '%s'", key_host);
    return CS_ERR_UNKNOWN_KEY;
}

int cs_unet_trace(cs_unet* net, int on) {
    CS_REQUIRE(net, "unet_trace: null handle");
    net->trace = on != 0;
    net->trace_log.clear();
    return CS_OK;
}

int cs_unet_trace_count(const cs_unet* net) { return net ? (int)net->trace_log.size() : CS_ERR_INVALID; }

int cs_unet_trace_entry(const cs_unet* net, int index, char* name_out_host, int64_t* info_out_host) {
    CS_REQUIRE(net && name_out_host && info_out_host && index >= 0 && index < (int)net->trace_log.size(), "unet_trace_entry: bad argument");
    const cs::TraceEntry& e = net->trace_log[index];
    strncpy(name_out_host, e.name.c_str(), 63);
    name_out_host[63] = 0;
    const int64_t info[8] = {e.off, e.n, e.h, e.w, e.c, e.pitch, e.kind, 0};
    for (int i = 0; i < 8; ++i) info_out_host[i] = info[i];
    return CS_OK;
}

int cs_unet_set_streams(cs_unet* net, int n) {
    CS_REQUIRE(net && n >= 1 && n <= 4, "set_streams: n must be 1..4");
    for (int i = 0; i < n && n > 1; ++i) {          // created here, never on the launch path
        if (!net->side[i]) CS_HIP(hipStreamCreateWithFlags(&net->side[i], hipStreamNonBlocking));
        if (!net->ev_join[i]) CS_HIP(hipEventCreateWithFlags(&net->ev_join[i], hipEventDisableTiming));
    }
    if (n > 1 && !net->ev_fork) CS_HIP(hipEventCreateWithFlags(&net->ev_fork, hipEventDisableTiming));
    net->n_streams = n;
    return CS_OK;
}

int cs_unet_destroy(cs_unet* net) {
    if (!net) return CS_OK;
    for (int i = 0; i < 4; ++i) {
        if (net->side[i]) (void)hipStreamDestroy(net->side[i]);
        if (net->ev_join[i]) (void)hipEventDestroy(net->ev_join[i]);
    }
    if (net->ev_fork) (void)hipEventDestroy(net->ev_fork);
    for (auto& st : net->bwd_side)
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    for (auto& ev : net->bwd_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (net->arena) (void)hipFree(net->arena);
    cs::pack_ctx_destroy(net->pack);
    cs::pack_ctx_destroy(net->pack_bound);
    cs::pack_ctx_destroy(net->pack_readback);
    delete net;
    return CS_OK;
}

int cs_unet_num_params(const cs_unet* net) { return net ? (int)net->params.size() : CS_ERR_INVALID; }

int cs_unet_param_info(const cs_unet* net, int index, char* key_out_host, int64_t* shape_out_host) {
    CS_REQUIRE(net && index >= 0 && index < (int)net->params.size(), "param_info: index %d out of range", index);
    const ParamEntry& p = net->params[index];
    if (key_out_host) {
        strncpy(key_out_host, p.key.c_str(), 159);
        key_out_host[159] = 0;
    }
    if (shape_out_host)
        for (int i = 0; i < 4; ++i) shape_out_host[i] = p.shape[i];
    return p.rank;
}

int cs_unet_load_param(cs_unet* net, const char* key_host, const float* data, int64_t numel, void* stream) {
    CS_REQUIRE(net && key_host && data, "load_param: null argument");
    if (int rcd = check_device(net)) return rcd;
    for (auto& p : net->params) {
        if (p.key == key_host) {
            CS_REQUIRE(numel == p.numel(), "load_param: '%s' expects %lld elements, got %lld", key_host, (long long)p.numel(),
                       (long long)numel);
            int rc = p.load(data, (hipStream_t)stream);
            if (rc == CS_OK) {
                p.loaded = true;
                p.dgrad_packed = net->training;
                p.wb_format = net->planes == 5 ? 3 : net->planes == 4 ? 2 : net->planes > 0 ? 1 : 0;
            }
            return rc;
        }
    }
    cs::set_error("load_param: unknown key '%s'", key_host);
    return CS_ERR_UNKNOWN_KEY;
}

int cs_unet_bind_params(cs_unet* net, const float* const* param_ptrs_host, int n_params, void* stream) {
    CS_REQUIRE(net && param_ptrs_host, "bind_params: null argument");
    CS_REQUIRE(n_params == (int)net->params.size(), "bind_params: %d pointers for %d parameters", n_params, (int)net->params.size());
    if (int rcd = check_device(net)) return rcd;
    hipStream_t s = (hipStream_t)stream;
    cs::pack_ctx_destroy(net->pack_bound);
    cs::pack_ctx_destroy(net->pack_readback);
    net->pack_bound = cs::pack_ctx_create(4 * (int)net->params.size() + 16);
    net->pack_readback = cs::pack_ctx_create((int)net->params.size() + 16);
    CS_REQUIRE(net->pack_bound && net->pack_readback, "bind_params: allocating the job tables failed");
    // the parameters' own load functions record into whatever context the handle points at: point it at the bound table
    cs::PackCtx* const own = net->pack;
    net->pack = net->pack_bound;
    cs::pack_batch_begin(net->pack);
    int rc = CS_OK;
    for (size_t i = 0; i < net->params.size() && rc == CS_OK; ++i) {
        ParamEntry& p = net->params[i];
        if (!param_ptrs_host[i]) { cs::set_error("bind_params: null pointer for '%s'", p.key.c_str()); rc = CS_ERR_INVALID; break; }
        rc = p.load(param_ptrs_host[i], s);
        if (rc == CS_OK) {
            p.loaded = true;
            p.dgrad_packed = net->training;
            p.wb_format = net->planes == 5 ? 3 : net->planes == 4 ? 2 : net->planes > 0 ? 1 : 0;
        }
    }
    if (rc == CS_OK) rc = cs::pack_batch_freeze(net->pack, s);
    net->pack = own;
    if (rc != CS_OK) return rc;
    cs::pack_batch_begin(net->pack_readback);
    for (size_t i = 0; i < net->params.size() && rc == CS_OK; ++i) {
        const ParamEntry& p = net->params[i];
        const size_t kl = p.key.size();
        const bool running = p.raw && ((kl > 13 && p.key.compare(kl - 13, 13, ".running_mean") == 0) || (kl > 12 && p.key.compare(kl - 12, 12, ".running_var") == 0));
        if (running) rc = cs::pack_copy(net->pack_readback, s, p.raw, const_cast<float*>(param_ptrs_host[i]), (size_t)p.numel());
    }
    if (rc == CS_OK) rc = cs::pack_batch_freeze(net->pack_readback, s);
    net->bound_planes = net->planes;
    net->bound_training = net->training;
    return rc;
}

int cs_unet_repack(cs_unet* net, void* stream) {
    CS_REQUIRE(net && net->pack_bound, "repack: call cs_unet_bind_params first");
    CS_REQUIRE(net->bound_planes == net->planes && net->bound_training == net->training,
               "repack: the parameters were bound under another precision / training mode: bind them again");
    return cs::pack_replay(net->pack_bound, (hipStream_t)stream);
}

int cs_unet_readback(cs_unet* net, void* stream) {
    CS_REQUIRE(net && net->pack_readback, "readback: call cs_unet_bind_params first");
    return cs::pack_replay(net->pack_readback, (hipStream_t)stream);
}

int cs_graph_begin(void* stream) {
    CS_REQUIRE(!cs::prof_enabled(), "graph_begin: per-kernel profiling (cs_prof_enable) records events around every launch and cannot be captured");
    CS_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed));
    return CS_OK;
}

int cs_graph_end(void* stream, cs_graph** out) {
    CS_REQUIRE(out, "graph_end: null argument");
    cs_graph* g = new cs_graph();
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g->graph);
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        cs::set_error("graph_end: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        cs_graph_destroy(g);
        return CS_ERR_HIP;
    }
    *out = g;
    return CS_OK;
}

int cs_graph_abort(void* stream) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture((hipStream_t)stream, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    return CS_OK;
}

int cs_graph_launch(cs_graph* g, void* stream) {
    CS_REQUIRE(g && g->exec, "graph_launch: null graph");
    CS_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return CS_OK;
}

int cs_graph_num_nodes(const cs_graph* g) {
    if (!g || !g->graph) return CS_ERR_INVALID;
    size_t n = 0;
    if (hipGraphGetNodes(g->graph, nullptr, &n) != hipSuccess) return CS_ERR_HIP;
    return (int)n;
}

int cs_graph_destroy(cs_graph* g) {
    if (!g) return CS_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return CS_OK;
}

size_t cs_unet_workspace_bytes(const cs_unet* net, int n, int h, int w) {
    if (!net || n <= 0) return 0;
    Bump b(nullptr, 0);
    WindowSrc src;
    if (unet_forward_impl(net, src, n, h, w, nullptr, b, nullptr) != CS_OK) return 0;
    const size_t records = ((size_t)n * h * w * 16 + 255) / 256 * 256;      // the batch as pixel records (inference: cs_unet_forward)
    return b.off + records + 256;
}

int cs_unet_forward(cs_unet* net, const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int n, int h, int w,
                    float* logits_nhwc, void* workspace, size_t workspace_bytes, void* stream) {
    CS_REQUIRE(net && x && logits_nhwc && workspace, "unet_forward: null argument");
    CS_REQUIRE(n > 0, "unet_forward: n = %d", n);
    if (int rcd = check_device(net)) return rcd;
    int rc = check_loaded(net);
    if (rc) return rc;
    WindowSrc src;
    src.image = x; src.sn = sn; src.sc = sc; src.sh = sh; src.sw = sw;
    src.origins = nullptr; src.img_h = h; src.img_w = w; src.roi_h = h; src.roi_w = w;
    Bump b(workspace, workspace_bytes);
    // inference: the batch once more as (r, g, b, 0) pixel records - the first layer then runs on the record kernels of kernels_first.hip and
    // the first unit's shortcut is recomputed by its sum launch (as in the window pass, swi_run_windows)
    if (!net->training && cs::first_rec_enabled() && net->in_ch == 3 && (int64_t)n * h * w < ((int64_t)1 << 31) && h < 65536 && n < 65536) {   // grid y / z limits of the record kernel
        const bool rec16 = act16_mode(net);                                      // unet_forward_impl's bf16 datapath
        void* rec = b.take<char>((size_t)n * h * w * 16);
        if (!b.ok()) { set_error("unet_forward: workspace too small"); return CS_ERR_WORKSPACE; }
        if (int rc0 = cs::launch_image_records((hipStream_t)stream, x, sn, sc, sh, sw, n, h, w, rec, rec16)) return rc0;
        src.rec = rec;
        src.rec16 = rec16;
    }
    // the activation tape is recorded for cs_unet_backward only in training mode (cs_unet_set_training)
    net->tape.valid = false;
    return unet_forward_impl(net, src, n, h, w, logits_nhwc, b, (hipStream_t)stream, net->training ? &net->tape : nullptr);
}

int64_t cs_unet_grad_numel(const cs_unet* net) { return net ? net->grad_floats : CS_ERR_INVALID; }

int64_t cs_unet_grad_offset(const cs_unet* net, int index) {
    if (!net || index < 0 || index >= (int)net->params.size()) return CS_ERR_INVALID;
    return net->params[index].grad_off;
}

static int backward_sizes(const cs_unet* net, size_t* acc_bytes, size_t* buf_bytes) {
    Bump acc(nullptr, 0), buf(nullptr, 0);
    int rc = unet_backward_impl(net, nullptr, acc, buf, nullptr, nullptr);
    *acc_bytes = acc.off;
    *buf_bytes = buf.off;
    return rc;
}

size_t cs_unet_backward_workspace_bytes(const cs_unet* net, int n, int h, int w) {
    if (!net || n <= 0) return 0;
    // sizes depend only on (n, h, w): run the planner on a scratch tape
    cs_unet* m = const_cast<cs_unet*>(net);
    Tape saved = m->tape;
    Bump fb(nullptr, 0);
    WindowSrc src;
    size_t a = 0, b = 0;
    int rc = unet_forward_impl(net, src, n, h, w, nullptr, fb, nullptr, &m->tape);
    if (rc == CS_OK) rc = backward_sizes(net, &a, &b);
    m->tape = saved;
    return rc == CS_OK ? a + b + 512 : 0;
}

int cs_unet_backward(cs_unet* net, const float* grad_logits_nhwc, void* workspace, size_t workspace_bytes, float* flat_grad,
                     void* stream) {
    CS_REQUIRE(net && grad_logits_nhwc && workspace && flat_grad, "unet_backward: null argument");
    CS_REQUIRE(net->tape.valid, "unet_backward: no forward has been recorded (cs_unet_set_training(1), then cs_unet_forward, and keep its workspace intact)");
    if (int rcd = check_device(net)) return rcd;
    if (int rc0 = check_loaded(net, true)) return rc0;
    hipStream_t s = (hipStream_t)stream;
    size_t acc_bytes = 0, buf_bytes = 0;
    int rc = backward_sizes(net, &acc_bytes, &buf_bytes);
    if (rc) return rc;
    acc_bytes = (acc_bytes + 255) / 256 * 256;
    if (acc_bytes + buf_bytes > workspace_bytes) {
        cs::set_error("unet_backward: workspace %zu < %zu bytes", workspace_bytes, acc_bytes + buf_bytes);
        return CS_ERR_WORKSPACE;
    }
    if (int zrc = launch_zero((hipStream_t)s, workspace, acc_bytes, flat_grad, (size_t)net->grad_floats * sizeof(float))) return zrc;
    Bump acc(workspace, acc_bytes), buf((char*)workspace + acc_bytes, workspace_bytes - acc_bytes);
    static const int side_n = [] { const char* e = getenv("CS_BWD_SIDE"); const int v = e ? atoi(e) : 1; return v < 0 ? 0 : v > 2 ? 2 : v; }();
    static const bool side_env = side_n != 0;     // A/B: CS_BWD_SIDE=0 = one stream (7.38 vs 6.90 ms per step), 2 = the weight-gradient groups alternate over two side streams
    if (side_env && !prof_enabled() && !net->bwd_side[0]) {      // (per-kernel HIP-event profiling keeps everything on one stream)
        for (auto& st : net->bwd_side) CS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (auto& ev : net->bwd_ev) CS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const bool use_side = side_env && !prof_enabled() && net->bwd_side[0];
    net->tape.valid = false;      // the gradient buffers reuse nothing of the tape, but one backward per forward (also when it fails)
    rc = unet_backward_impl(net, grad_logits_nhwc, acc, buf, flat_grad, s, net->bwd_side, use_side ? side_n : 0, net->bwd_ev, 48);
    if (rc != CS_OK && use_side) {
        // an early return skipped the join at the end of the pass: side-stream weight gradients may still be reading / writing the
        // caller's buffers, which the caller is about to free - drain them before reporting the error
        for (auto& st : net->bwd_side)
            if (st) (void)hipStreamSynchronize(st);
    }
    return rc;
}

size_t cs_dice_ce_workspace_bytes(int n, int classes) { return (n > 0 && classes > 0) ? dice_ce_workspace_bytes(n, classes) : 0; }

int cs_dice_ce_forward(const float* logits, const int64_t* logit_strides, const float* target, const int64_t* target_strides, int n, int classes,
                       int h, int w, float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce, float* loss_out, void* workspace,
                       size_t workspace_bytes, void* stream) {
    CS_REQUIRE(logit_strides && target_strides, "dice_ce_forward: null strides");
    return dice_ce_forward((hipStream_t)stream, logits, logit_strides, target, target_strides, n, classes, h, w, smooth_nr, smooth_dr, lambda_dice,
                           lambda_ce, loss_out, workspace, workspace_bytes);
}

int cs_dice_ce_backward(const float* logits, const int64_t* logit_strides, const float* target, const int64_t* target_strides, int n, int classes,
                        int h, int w, float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce, const float* grad_loss,
                        const void* workspace, size_t workspace_bytes, float* grad_logits, void* stream) {
    CS_REQUIRE(logit_strides && target_strides, "dice_ce_backward: null strides");
    return dice_ce_backward((hipStream_t)stream, logits, logit_strides, target, target_strides, n, classes, h, w, smooth_nr, smooth_dr, lambda_dice,
                            lambda_ce, grad_loss, workspace, workspace_bytes, grad_logits);
}

int cs_adamw_step(const int64_t* table, int n_tensors, int64_t total, const float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
    return adamw_step((hipStream_t)stream, table, n_tensors, total, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step);
}

int cs_adamw_step_dev(const int64_t* table, int n_tensors, int64_t total, const float* grad, float* exp_avg, float* exp_avg_sq, float* hyper,
                      int32_t* step, void* stream) {
    return adamw_step_dev((hipStream_t)stream, table, n_tensors, total, grad, exp_avg, exp_avg_sq, hyper, step);
}

int cs_swi_window_start(int i, int interval, int padded_size, int roi) {
    const int s = i * interval;
    return s < padded_size - roi ? s : padded_size - roi;
}

int cs_swi_make_plan(int img_h, int img_w, int roi_h, int roi_w, double overlap, cs_swi_plan* pl) {
    CS_REQUIRE(pl && img_h > 0 && img_w > 0 && roi_h > 0 && roi_w > 0, "swi plan: bad sizes");
    CS_REQUIRE(overlap >= 0.0 && overlap < 1.0, "swi plan: overlap %.3f out of range", overlap);
    pl->img_h = img_h; pl->img_w = img_w; pl->roi_h = roi_h; pl->roi_w = roi_w;
    pl->pad_h = img_h > roi_h ? img_h : roi_h;
    pl->pad_w = img_w > roi_w ? img_w : roi_w;
    pl->pad_top = (pl->pad_h - img_h) / 2;
    pl->pad_left = (pl->pad_w - img_w) / 2;
    auto interval = [&](int size, int roi) {
        int iv = size == roi ? roi : (int)(roi * (1.0 - overlap));
        return iv > 0 ? iv : 1;
    };
    pl->interval_h = interval(pl->pad_h, roi_h);
    pl->interval_w = interval(pl->pad_w, roi_w);
    auto count = [](int size, int roi, int iv) {
        int n = 0;
        while ((long long)n * iv + roi < size) ++n;
        return n + 1;
    };
    pl->n_h = count(pl->pad_h, roi_h, pl->interval_h);
    pl->n_w = count(pl->pad_w, roi_w, pl->interval_w);
    return CS_OK;
}

static void swi_row_windows(const cs_swi_plan* pl, int row_begin, int row_end, int* ih_lo, int* ih_hi) {
    // window rows whose padded extent [hs, hs + roi) intersects padded rows [row_begin + pad_top, row_end + pad_top)
    const int p0 = row_begin + pl->pad_top, p1 = row_end + pl->pad_top;
    int lo = pl->n_h, hi = -1;
    for (int ih = 0; ih < pl->n_h; ++ih) {
        const int hs = cs_swi_window_start(ih, pl->interval_h, pl->pad_h, pl->roi_h);
        if (hs < p1 && hs + pl->roi_h > p0) {
            if (ih < lo) lo = ih;
            if (ih > hi) hi = ih;
        }
    }
    *ih_lo = lo;
    *ih_hi = hi;
}

size_t cs_swi_windows_workspace_bytes(const cs_unet* net, const cs_swi_plan* pl, int engine_batch) {
    if (!net || !pl || engine_batch <= 0) return 0;
    const size_t nwin = (size_t)pl->n_h * pl->n_w;
    const size_t eb = (size_t)engine_batch < nwin ? (size_t)engine_batch : nwin;
    const size_t unet = cs_unet_workspace_bytes(net, (int)eb, pl->roi_h, pl->roi_w) * (size_t)net->n_streams;
    if (unet == 0) return 0;
    const size_t records = ((size_t)pl->img_h * pl->img_w * 16 + 255) / 256 * 256;      // the image as pixel records (swi_run_windows)
    return unet + (nwin * 8 + 255) / 256 * 256 + records + 512;
}

size_t cs_swi_workspace_bytes(const cs_unet* net, const cs_swi_plan* pl, int engine_batch) {
    const size_t win = cs_swi_windows_workspace_bytes(net, pl, engine_batch);
    if (win == 0) return 0;
    const size_t nwin = (size_t)pl->n_h * pl->n_w;
    return win + (nwin * pl->roi_h * pl->roi_w * net->out_ch * 4 + 255) / 256 * 256;
}

// Window pass over the windows [k_lo, k_hi) of the row-major window list: logits of window k go to win_logits[k - k_lo].
static int swi_run_windows(cs_unet* net, const cs_swi_plan* pl, const float* image, int64_t sc, int64_t sh, int64_t sw, int engine_batch,
                           int k_lo, int k_hi, float* win_logits, cs::Bump& b, hipStream_t s) {
    const int nwin = k_hi - k_lo;
    const size_t win_floats = (size_t)pl->roi_h * pl->roi_w * net->out_ch;
    int2* origins = b.take<int2>((size_t)nwin);
    if (!b.ok()) { cs::set_error("swi: workspace too small"); return CS_ERR_WORKSPACE; }
    hipLaunchKernelGGL(cs::swi_origins_kernel, dim3((unsigned)cs::ceil_div(nwin, 256)), dim3(256), 0, s, *pl, k_lo, nwin, origins);
    CS_HIP(hipGetLastError());
    WindowSrc src;
    src.image = image; src.sn = 0; src.sc = sc; src.sh = sh; src.sw = sw;
    src.img_h = pl->img_h; src.img_w = pl->img_w; src.roi_h = pl->roi_h; src.roi_w = pl->roi_w;
    // the image once more as (r, g, b, 0) pixel records: every window's first layer gathers its patches from these (kernels_first.hip)
    if (cs::first_rec_enabled() && net->in_ch == 3 && (int64_t)pl->img_h * pl->img_w < ((int64_t)1 << 31) && pl->img_h < 65536) {   // taller images keep the planar path
        const bool rec16 = act16_mode(net);                                        // unet_forward_impl's bf16 datapath
        void* rec = b.take<char>((size_t)pl->img_h * pl->img_w * 16);
        if (!b.ok()) { cs::set_error("swi: workspace too small"); return CS_ERR_WORKSPACE; }
        // only the image rows the windows [k_lo, k_hi) read (a rank's share of a slide: an eighth of the rows + a window's height)
        const int i_lo = k_lo / pl->n_w, i_hi = (k_hi - 1) / pl->n_w;
        int r0 = cs_swi_window_start(i_lo, pl->interval_h, pl->pad_h, pl->roi_h) - pl->pad_top;
        int r1 = cs_swi_window_start(i_hi, pl->interval_h, pl->pad_h, pl->roi_h) - pl->pad_top + pl->roi_h;
        r0 = r0 < 0 ? 0 : (r0 > pl->img_h ? pl->img_h : r0);
        r1 = r1 > pl->img_h ? pl->img_h : (r1 < r0 ? r0 : r1);
        if (int rc0 = cs::launch_image_records(s, image, 0, sc, sh, sw, 1, pl->img_h, pl->img_w, rec, rec16, r0, r1)) return rc0;
        src.rec = rec;
        src.rec16 = rec16;
    }
    // sliding-window inference is an inference call: BatchNorm layers use their running statistics whatever mode the handle was
    // left in (a train-mode handle would otherwise normalise every window chunk with that chunk's own statistics and update the
    // running ones once per chunk, from several streams)
    struct BatchStatsOff {
        cs_unet* n; bool saved;
        explicit BatchStatsOff(cs_unet* m) : n(m), saved(m->batch_stats) { m->batch_stats = false; }
        ~BatchStatsOff() { n->batch_stats = saved; }
    } stats_guard(net);
    // chunks of `engine_batch` windows; with side streams the chunks are dealt round-robin to them, each stream owning
    // its own slice of the activation workspace (stream order serialises the chunks that share a slice).  Per-kernel
    // profiling (cs_prof_enable) forces one stream so that event-timed durations are not inflated by overlap.
    const int ns = (net->n_streams > 1 && !cs::prof_enabled() && nwin > engine_batch) ? net->n_streams : 1;
    const size_t mark = b.off;
    const size_t slice = ns > 1 ? ((b.cap - mark) / ns) / 256 * 256 : 0;
    if (ns > 1) {
        CS_HIP(hipEventRecord(net->ev_fork, s));
        for (int i = 0; i < ns; ++i) CS_HIP(hipStreamWaitEvent(net->side[i], net->ev_fork, 0));
    }
    int chunk = 0, rc = CS_OK;
    for (int k0 = 0; k0 < nwin && rc == CS_OK; k0 += engine_batch, ++chunk) {
        const int nb = nwin - k0 < engine_batch ? nwin - k0 : engine_batch;
        src.origins = origins + k0;
        if (ns > 1) {
            cs::Bump bs(b.base + mark + (size_t)(chunk % ns) * slice, slice);
            rc = unet_forward_impl(net, src, nb, pl->roi_h, pl->roi_w, win_logits + (size_t)k0 * win_floats, bs, net->side[chunk % ns]);
        } else {
            b.off = mark;   // every chunk reuses the same activation workspace (stream order serialises them)
            rc = unet_forward_impl(net, src, nb, pl->roi_h, pl->roi_w, win_logits + (size_t)k0 * win_floats, b, s);
        }
    }
    if (ns > 1) {                                  // joined on every path: the caller may free the workspace after an error
        for (int i = 0; i < ns; ++i) {
            CS_HIP(hipEventRecord(net->ev_join[i], net->side[i]));
            CS_HIP(hipStreamWaitEvent(s, net->ev_join[i], 0));
        }
    }
    return rc;
}

int cs_swi_windows(cs_unet* net, const cs_swi_plan* pl, const float* image, int64_t sc, int64_t sh, int64_t sw, int engine_batch,
                   int win_row_begin, int win_row_end, float* win_logits, void* workspace, size_t workspace_bytes, void* stream) {
    CS_REQUIRE(net && pl && image && workspace && win_logits, "swi_windows: null argument");
    CS_REQUIRE(engine_batch > 0, "swi_windows: engine_batch %d", engine_batch);
    CS_REQUIRE(win_row_begin >= 0 && win_row_end <= pl->n_h && win_row_begin <= win_row_end, "swi_windows: bad window-row range [%d, %d) of %d",
               win_row_begin, win_row_end, pl->n_h);
    if (int rcd = check_device(net)) return rcd;
    if (int rc = check_loaded(net)) return rc;
    if (win_row_begin == win_row_end) return CS_OK;
    cs::Bump b(workspace, workspace_bytes);
    return swi_run_windows(net, pl, image, sc, sh, sw, engine_batch, win_row_begin * pl->n_w, win_row_end * pl->n_w, win_logits, b, (hipStream_t)stream);
}

int cs_swi_windows_range(cs_unet* net, const cs_swi_plan* pl, const float* image, int64_t sc, int64_t sh, int64_t sw, int engine_batch,
                         int win_begin, int win_end, float* win_logits, void* workspace, size_t workspace_bytes, void* stream) {
    CS_REQUIRE(net && pl && image && workspace && win_logits, "swi_windows_range: null argument");
    CS_REQUIRE(engine_batch > 0, "swi_windows_range: engine_batch %d", engine_batch);
    CS_REQUIRE(win_begin >= 0 && win_end <= pl->n_h * pl->n_w && win_begin <= win_end, "swi_windows_range: bad window range [%d, %d) of %d", win_begin,
               win_end, pl->n_h * pl->n_w);
    if (int rcd = check_device(net)) return rcd;
    if (int rc = check_loaded(net)) return rc;
    if (win_begin == win_end) return CS_OK;
    cs::Bump b(workspace, workspace_bytes);
    return swi_run_windows(net, pl, image, sc, sh, sw, engine_batch, win_begin, win_end, win_logits, b, (hipStream_t)stream);
}

int cs_swi_partition(const cs_swi_plan* pl, int rank, int world, cs_swi_strip* out) {
    CS_REQUIRE(pl && out && world >= 1 && rank >= 0 && rank < world, "swi_partition: bad argument (rank %d of %d)", rank, world);
    auto first_row = [&](int r) {                     // first window row of rank r: shares differ by at most one, the LARGER ones last -
        const int base = pl->n_h / world, rem = pl->n_h % world;   // rank 0, which also labels the gathered mask, gets a smaller one
        const int extra = r - (world - rem);
        return r * base + (extra > 0 ? extra : 0);
    };
    auto first_out = [&](int r) {                     // first output row (image coordinates) rank r finalises
        const int a = r >= world ? pl->n_h : first_row(r);
        if (a <= 0) return 0;
        if (a >= pl->n_h) return pl->img_h;
        int y = cs_swi_window_start(a, pl->interval_h, pl->pad_h, pl->roi_h) - pl->pad_top;
        y = y < 0 ? 0 : y;
        return y > pl->img_h ? pl->img_h : y;
    };
    out->win_row_begin = first_row(rank);
    out->win_row_end = rank + 1 >= world ? pl->n_h : first_row(rank + 1);
    out->row_begin = first_out(rank);
    out->row_end = first_out(rank + 1);
    if (out->win_row_begin == out->win_row_end) out->row_end = out->row_begin;      // a rank without windows finalises nothing
    // earlier ranks' window rows that still cover this rank's first output row
    int halo = out->win_row_begin;
    if (out->row_end > out->row_begin) {
        const int p0 = out->row_begin + pl->pad_top;
        while (halo > 0 && cs_swi_window_start(halo - 1, pl->interval_h, pl->pad_h, pl->roi_h) + pl->roi_h > p0) --halo;
    }
    out->halo_begin = halo;
    return CS_OK;
}

int cs_swi_infer(cs_unet* net, const cs_swi_plan* pl, const float* image, int64_t sc, int64_t sh, int64_t sw, int engine_batch,
                 int row_begin, int row_end, float* logits_nchw, float* prob1, uint8_t* mask, void* workspace,
                 size_t workspace_bytes, void* stream) {
    CS_REQUIRE(net && pl && image && workspace, "swi_infer: null argument");
    CS_REQUIRE(engine_batch > 0, "swi_infer: engine_batch %d", engine_batch);
    CS_REQUIRE(row_begin >= 0 && row_end <= pl->img_h && row_begin < row_end, "swi_infer: bad row range [%d, %d)", row_begin, row_end);
    if (int rcd = check_device(net)) return rcd;
    int rc = check_loaded(net);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    int ih_lo, ih_hi;
    swi_row_windows(pl, row_begin, row_end, &ih_lo, &ih_hi);
    CS_REQUIRE(ih_hi >= ih_lo, "swi_infer: no window intersects rows [%d, %d)", row_begin, row_end);
    const int nwin = (ih_hi - ih_lo + 1) * pl->n_w;
    Bump b(workspace, workspace_bytes);
    const size_t win_floats = (size_t)pl->roi_h * pl->roi_w * net->out_ch;
    float* win_logits = b.take<float>((size_t)nwin * win_floats);
    if (!b.ok()) { cs::set_error("swi_infer: workspace too small"); return CS_ERR_WORKSPACE; }
    rc = swi_run_windows(net, pl, image, sc, sh, sw, engine_batch, ih_lo * pl->n_w, (ih_hi + 1) * pl->n_w, win_logits, b, s);
    if (rc) return rc;
    SwiFinalizeArgs fa{};
    fa.win_logits = win_logits;
    fa.plan = *pl;
    fa.c = net->out_ch;
    fa.ih_base = ih_lo;
    fa.row_begin = row_begin;
    fa.row_end = row_end;
    fa.logits_nchw = logits_nchw;
    fa.prob1 = prob1;
    fa.mask = mask;
    return launch_swi_finalize(s, fa);
}

int cs_swi_finalize(const cs_swi_plan* pl, const float* win_logits, int classes, int ih_base, int row_begin, int row_end,
                    float* logits_nchw, float* prob1, uint8_t* mask, void* stream) {
    CS_REQUIRE(pl && win_logits, "swi_finalize: null argument");
    SwiFinalizeArgs fa{};
    fa.win_logits = win_logits;
    fa.plan = *pl;
    fa.c = classes;
    fa.ih_base = ih_base;
    fa.row_begin = row_begin;
    fa.row_end = row_end;
    fa.logits_nchw = logits_nchw;
    fa.prob1 = prob1;
    fa.mask = mask;
    return launch_swi_finalize((hipStream_t)stream, fa);
}

int cs_swi_finalize_rect(const cs_swi_plan* pl, const float* win_logits, int classes, int ih_base, int row_begin, int row_end, int col_begin,
                         int col_end, float* logits_nchw, float* prob1, uint8_t* mask, void* stream) {
    CS_REQUIRE(pl && win_logits, "swi_finalize_rect: null argument");
    if (col_end <= col_begin || row_end <= row_begin) return CS_OK;
    SwiFinalizeArgs fa{};
    fa.win_logits = win_logits;
    fa.plan = *pl;
    fa.c = classes;
    fa.ih_base = ih_base;
    fa.row_begin = row_begin;
    fa.row_end = row_end;
    fa.col_begin = col_begin;
    fa.col_end = col_end;
    fa.logits_nchw = logits_nchw;
    fa.prob1 = prob1;
    fa.mask = mask;
    return launch_swi_finalize((hipStream_t)stream, fa);
}

int cs_swi_partition_windows(const cs_swi_plan* pl, int rank, int world, cs_swi_share* out) {
    CS_REQUIRE(pl && out && world >= 1 && rank >= 0 && rank < world, "swi_partition_windows: bad argument (rank %d of %d)", rank, world);
    const int nwin = pl->n_h * pl->n_w;
    auto first_win = [&](int r) { return (int)(((long long)r * nwin) / world); };            // shares differ by at most one window
    // image row / column at which the band of window row / column i begins (the pixels whose LAST covering window row / column is i)
    auto band = [&](int i, int n, int interval, int pad_size, int roi, int pad_before, int img) {
        if (i <= 0) return 0;
        if (i >= n) return img;
        int v = cs_swi_window_start(i, interval, pad_size, roi) - pad_before;
        v = v < 0 ? 0 : v;
        return v > img ? img : v;
    };
    auto row_of = [&](int i) { return band(i, pl->n_h, pl->interval_h, pl->pad_h, pl->roi_h, pl->pad_top, pl->img_h); };
    auto col_of = [&](int j) { return band(j, pl->n_w, pl->interval_w, pl->pad_w, pl->roi_w, pl->pad_left, pl->img_w); };
    memset(out, 0, sizeof(*out));
    const int a = first_win(rank), b = first_win(rank + 1);
    out->win_begin = a;
    out->win_end = b;
    if (a == b) {                                       // a rank without windows owns nothing
        out->slot_row_begin = out->slot_row_end = a / (pl->n_w > 0 ? pl->n_w : 1);
        return CS_OK;
    }
    const int j0 = a / pl->n_w, c0 = a % pl->n_w, j1 = (b - 1) / pl->n_w, c1 = (b - 1) % pl->n_w;
    int n = 0;
    auto add = [&](int r0, int r1, int x0, int x1) {
        if (r1 > r0 && x1 > x0) {
            out->rect[n][0] = r0; out->rect[n][1] = r1; out->rect[n][2] = x0; out->rect[n][3] = x1;
            ++n;
        }
    };
    if (j0 == j1) {
        add(row_of(j0), row_of(j0 + 1), col_of(c0), col_of(c1 + 1));
    } else {
        add(row_of(j0), row_of(j0 + 1), col_of(c0), pl->img_w);
        add(row_of(j0 + 1), row_of(j1), 0, pl->img_w);
        add(row_of(j1), row_of(j1 + 1), 0, col_of(c1 + 1));
    }
    out->n_rects = n;
    // earlier window rows that still cover the first owned image row
    int halo = j0;
    const int p0 = row_of(j0) + pl->pad_top;
    while (halo > 0 && cs_swi_window_start(halo - 1, pl->interval_h, pl->pad_h, pl->roi_h) + pl->roi_h > p0) --halo;
    out->slot_row_begin = halo;
    out->slot_row_end = j1 + 1;
    return CS_OK;
}

size_t cs_normalize_workspace_bytes(int h, int w) { return cs::normalize_workspace_bytes(h, w); }

int cs_normalize_image(const void* image, int bytes_per_sample, int h, int w, int channels, float* out_hwc, uint8_t* u8_hwc,
                       void* workspace, size_t workspace_bytes, void* stream) {
    CS_REQUIRE(image && out_hwc && workspace, "normalize_image: null argument");
    CS_REQUIRE(bytes_per_sample == 1 || bytes_per_sample == 2, "normalize_image: only uint8 / uint16 samples (got %d bytes)", bytes_per_sample);
    return cs::normalize_image((hipStream_t)stream, image, bytes_per_sample == 2, h, w, channels, out_hwc, u8_hwc, workspace,
                               workspace_bytes);
}

int cs_label_max(const int32_t* a, const int32_t* b, int64_t n, int32_t* out2, void* stream) {
    CS_REQUIRE(a && out2, "label_max: null argument");
    return cs::label_max((hipStream_t)stream, a, b, n, out2);
}

int cs_label_overlap(const int32_t* gt, const int32_t* seg, int64_t n, int32_t n_gt, int32_t n_seg, uint32_t* overlap, void* stream) {
    CS_REQUIRE(gt && seg && overlap && n_gt >= 0 && n_seg >= 0, "label_overlap: bad argument");
    return cs::label_overlap((hipStream_t)stream, gt, seg, n, n_gt + 1, n_seg + 1, overlap);
}

size_t cs_remove_boundary_workspace_bytes(int32_t n_labels) { return cs::remove_boundary_workspace_bytes(n_labels); }

int cs_remove_boundary_cells(const int32_t* labels, int h, int w, int border, int32_t n_labels, int32_t* out, int32_t* n_out,
                             void* workspace, size_t workspace_bytes, void* stream) {
    CS_REQUIRE(labels && out && workspace && h >= 0 && w >= 0 && border >= 0 && n_labels >= 0, "remove_boundary_cells: bad argument");
    return cs::remove_boundary_cells((hipStream_t)stream, labels, h, w, border, n_labels, out, n_out, workspace, workspace_bytes);
}

size_t cs_augment_workspace_bytes(int n, int channels, int size) { return cs::augment_workspace_bytes(n, channels, size); }

int cs_augment_batch(const float* img_arena, const uint8_t* lab_arena, const cs_aug_params* params_device, int n, int channels, int size,
                     float* out_img, uint8_t* out_lab, void* workspace, size_t workspace_bytes, void* stream) {
    CS_REQUIRE(img_arena && lab_arena && params_device && out_img && out_lab && workspace, "augment_batch: null argument");
    return cs::augment_batch((hipStream_t)stream, img_arena, lab_arena, params_device, n, channels, size, out_img, out_lab, workspace,
                             workspace_bytes);
}

size_t cs_label_deflate_workspace_bytes(int h, int w, int rows_per_strip) { return cs::label_deflate_workspace_bytes(h, w, rows_per_strip); }

size_t cs_label_deflate_bound(int h, int w, int rows_per_strip) { return cs::label_deflate_bound(h, w, rows_per_strip); }

int cs_label_deflate(const int32_t* labels, int h, int w, int rows_per_strip, uint8_t* out, size_t out_capacity, uint32_t* meta,
                     void* workspace, size_t workspace_bytes, void* stream) {
    CS_REQUIRE(labels && out && meta && workspace, "label_deflate: null argument");
    return cs::label_deflate((hipStream_t)stream, labels, h, w, rows_per_strip, out, out_capacity, meta, workspace, workspace_bytes);
}

size_t cs_ccl_workspace_bytes(int h, int w) { return cs::ccl_workspace_bytes(h, w); }

int cs_ccl_postprocess(const uint8_t* mask, int h, int w, int hole_area, int min_size, int32_t* labels, int32_t* n_labels,
                       void* workspace, size_t workspace_bytes, void* stream) {
    CS_REQUIRE(mask && labels && workspace, "ccl_postprocess: null argument");
    return cs::ccl_postprocess((hipStream_t)stream, mask, h, w, hole_area, min_size, labels, n_labels, workspace, workspace_bytes);
}

int cs_interior_map(const int32_t* inst, int h, int w, int min_size, uint8_t* out, void* workspace, size_t workspace_bytes,
                    void* stream) {
    CS_REQUIRE(inst && out && workspace, "interior_map: null argument");
    return cs::interior_map((hipStream_t)stream, inst, h, w, min_size, out, workspace, workspace_bytes);
}

int cs_ccl_label(const uint8_t* mask, int h, int w, int connectivity, int32_t* labels, int32_t* n_labels, void* workspace,
                 size_t workspace_bytes, void* stream) {
    CS_REQUIRE(mask && labels && workspace, "ccl_label: null argument");
    return cs::ccl_label((hipStream_t)stream, mask, h, w, connectivity, labels, n_labels, workspace, workspace_bytes);
}

int cs_ccl_strip_halo(int hole_area, int min_size) { return cs::ccl_strip_halo(hole_area, min_size); }
int cs_ccl_strip_edge_ints(int w) { return cs::ccl_strip_edge_ints(w); }
size_t cs_ccl_seam_workspace_bytes(int world, int w) { return cs::ccl_seam_workspace_bytes(world, w); }

int cs_ccl_strip_local(const uint8_t* mask_rows, int ext_h, int w, int strip_row0, int strip_h, int hole_area, int min_size,
                       int32_t* edge_record, void* workspace, size_t workspace_bytes, void* stream) {
    CS_REQUIRE(w == 0 || (edge_record && workspace && (mask_rows || ext_h == 0)), "ccl_strip_local: null argument");
    return cs::ccl_strip_local((hipStream_t)stream, mask_rows, ext_h, w, strip_row0, strip_h, hole_area, min_size, edge_record, workspace,
                               workspace_bytes);
}

int cs_ccl_strip_resolve(const int32_t* all_records, int world, int rank, int ext_h, int w, int strip_h, int32_t* edge_ids, int32_t* n_labels,
                         void* workspace, size_t workspace_bytes, void* seam_workspace, size_t seam_workspace_bytes, void* stream) {
    CS_REQUIRE((all_records && edge_ids && workspace && seam_workspace) || w == 0, "ccl_strip_resolve: null argument");
    return cs::ccl_strip_resolve((hipStream_t)stream, all_records, world, rank, ext_h, w, strip_h, edge_ids, n_labels, workspace,
                                 workspace_bytes, seam_workspace, seam_workspace_bytes);
}

int cs_ccl_strip_finish(const int32_t* all_edge_ids, int world, int rank, int ext_h, int w, int strip_h, int32_t* labels, void* workspace,
                        size_t workspace_bytes, void* stream) {
    CS_REQUIRE((all_edge_ids && labels && workspace) || w == 0 || strip_h == 0, "ccl_strip_finish: null argument");
    return cs::ccl_strip_finish((hipStream_t)stream, all_edge_ids, world, rank, ext_h, w, strip_h, labels, workspace, workspace_bytes);
}

int cs_prof_enable(int on) {
    cs::prof_state().on = on != 0;
    if (on) cs::g_prof_any = true;
    return CS_OK;
}

int cs_prof_is_enabled(void) { return cs::prof_enabled() ? 1 : 0; }

int cs_prof_reset(void) {
    cs::ProfState& ps = cs::prof_state();
    cs::prof_drain(ps);
    for (int i = 0; i < K_NUM_IDS; ++i) {
        ps.flops[i] = ps.bytes[i] = ps.ms[i] = 0.0;
        ps.launches[i] = 0;
    }
    return CS_OK;
}

int cs_prof_collect(char* names_out_host, double* ms_out_host, int64_t* launches_out_host, double* flops_out_host,
                    double* bytes_out_host) {
    cs::ProfState& ps = cs::prof_state();
    cs::prof_drain(ps);
    for (int i = 0; i < K_NUM_IDS; ++i) {
        if (names_out_host) {
            strncpy(names_out_host + 48 * i, cs::kernel_name(i), 47);
            names_out_host[48 * i + 47] = 0;
        }
        if (ms_out_host) ms_out_host[i] = ps.ms[i];
        if (launches_out_host) launches_out_host[i] = ps.launches[i];
        if (flops_out_host) flops_out_host[i] = ps.flops[i];
        if (bytes_out_host) bytes_out_host[i] = ps.bytes[i];
    }
    return K_NUM_IDS;
}

}  // extern "C"
