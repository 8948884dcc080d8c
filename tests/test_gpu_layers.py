"""Per-layer oracle for the 16-bit and fp8 datapaths (bf16 / fp8 modes of cellseg_amd.UNet).

The fp32 path is held to 1e-4 against the oracle net.  The low-precision datapaths cannot be: their logits differ from fp32 by the
rounding they are allowed to do.  What CAN be held tightly is every single layer: given the layer's own inputs exactly as the
engine stored them (read out of the workspace through cs_unet_trace), a torch-CPU float64 convolution over the operands rounded
the way the kernel is specified to round them - activations: lazy InstanceNorm + PReLU in fp32, then bf16 (RNE) or e4m3; weights:
bf16 (RNE) or e4m3 with one scale per output channel (amax / 448) - must reproduce the stored output to one rounding step of the
output type, and the layer's InstanceNorm table must be the statistics of the unrounded accumulators.  A dropped tap, a wrong
rounding of the staged activations, a mis-scaled channel or a stale weight plane shows up here whatever the other layers do.

Reference call sites: the conv / transposed-conv / norm / PReLU dispatches of baseline/predict.py:109 -> :45-52 (MONAI UNet blocks,
oracle/unet_ref.py for their composition).
"""
import json
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16_ULP = 2.0 ** -8          # one RNE rounding to bf16 (8 significand bits) moves a value by at most this fraction of itself
E4M3_ULP = 2.0 ** -4          # ... to e4m3 (4 significand bits)


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda:0")


def bf16r(t):
    """Round to bf16 (RNE) and return float64."""
    return t.float().to(torch.bfloat16).double()


def e4m3r(t):
    return t.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).double()


def nchw(t):
    return t.permute(0, 3, 1, 2)


def nhwc(t):
    return t.permute(0, 2, 3, 1)


def activated(raw, tab, slope):
    """act(norm(x)) as the kernels compute it: prelu(fma(x, scale, shift)) in fp32.  raw [n,h,w,c] (stored type), tab [n,c,2]."""
    x = raw.double()
    y = (x * tab[:, None, None, :, 0].double() + tab[:, None, None, :, 1].double()).float()      # one rounding = the fp32 FMA
    return torch.where(y >= 0, y, y * float(slope)).float()


def table_of(acc, eps=1e-5):
    """(scale, shift) of InstanceNorm2d(affine=False) from unrounded conv outputs acc [n,h,w,c] (float64)."""
    mean = acc.mean(dim=(1, 2))
    var = acc.var(dim=(1, 2), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + eps)
    return torch.stack([rstd, -mean * rstd], -1)


def fp8_weight(w, out_dim):
    """e4m3 weights with one fp32 scale per OUTPUT channel, as pack_conv_weights_fp8 quantises them -> (dequantisable values float64, scale)."""
    dims = [d for d in range(w.dim()) if d != out_dim]
    amax = w.abs().amax(dim=dims, keepdim=True).float()
    scale = torch.where(amax > 0, amax * torch.tensor(1.0 / 448.0, dtype=torch.float32), torch.ones_like(amax))
    inv = (torch.ones_like(scale) / scale).float()
    q = e4m3r(w.float() * inv)
    return q, scale.double()


class Net:
    """Weights of a cellseg_amd.UNet by role (MONAI state-dict keys, oracle/unet_ref.py)."""

    def __init__(self, model):
        self.sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        self.nl = len(model.channels) - 1
        self.ch = model.channels

    def blk(self, d):
        return "model" + ".1.submodule" * d

    def down(self, d):
        return self.blk(d) + ".0"

    def bottom(self):
        return self.blk(self.nl - 1) + ".1.submodule"

    def up(self, d):
        return self.blk(d) + ".2"

    def w(self, key):
        return self.sd[key + ".weight"].double(), self.sd[key + ".bias"].double()

    def slope(self, key):
        return float(self.sd[key + ".adn.A.weight"])


class Report:
    def __init__(self):
        self.rows = []

    def compare(self, name, got, ref, ulp, what="out"):
        """got: stored tensor; ref: float64 unrounded reference.  ONE rounding of the output type around ref (|got - ref| <= ulp * |ref|),
        plus half of that at the tensor's rms for values near zero (e4m3 subnormals) and fp32 accumulation-order flips at rounding ties."""
        got, ref = got.double().cpu(), ref.double()
        rms = float(ref.pow(2).mean().sqrt())
        tol = ulp * ref.abs() + 0.5 * ulp * rms + 1e-30
        err = (got - ref).abs()
        worst = float((err / tol).max())
        self.rows.append({"layer": name, "what": what, "worst_over_tol": round(worst, 3), "max_abs": float(err.max()), "rms": rms,
                          "frac_over_half_tol": float((err > 0.5 * tol).double().mean())})
        return worst

    def compare_abs(self, name, got, ref, rel, what):
        got, ref = got.double().cpu(), ref.double()
        scale = float(ref.abs().max()) + 1e-30
        worst = float((got - ref).abs().max()) / (rel * scale)
        self.rows.append({"layer": name, "what": what, "worst_over_tol": round(worst, 3), "max_abs": float((got - ref).abs().max()), "rms": scale,
                          "frac_over_half_tol": 0.0})
        return worst

    def compare_table(self, name, tab, acc, eps=1e-5):
        """The lazy-norm table (scale, shift) = (rstd, -mean * rstd) against the statistics of the unrounded accumulators acc [n,h,w,c].
        The 16-bit datapaths sum a lane's <= 32 values in fp32 before going to fp64 (relative error <= 2e-6 of sum x^2), and
        var = E[x^2] - mean^2 turns that into d(rstd) <= 1e-6 * E[x^2] * rstd^3 - invisible on real maps, dominant on 2 x 2 maps of
        near-constant channels (rstd in the hundreds).  There the fp32 accumulation of the convolution itself shows too: an absolute
        perturbation delta of the outputs (K <= 9 216 products of O(1) operands summed in fp32: a few 1e-6) moves rstd by rstd^2 * delta.
        Tolerance: 2e-4 of the value + four times the first bound + the second with delta = 8e-6 (on maps of >= 64 pixels the tables sit below a tenth of this tolerance)."""
        tab, acc = tab.double().cpu(), acc.double()
        mean, ex2 = acc.mean(dim=(1, 2)), acc.pow(2).mean(dim=(1, 2))
        rstd = 1.0 / torch.sqrt(ex2 - mean * mean + eps)
        d_rstd = 4e-6 * ex2 * rstd ** 3 + 8e-6 * rstd ** 2
        tol_scale = 2e-4 * rstd + d_rstd
        tol_shift = 2e-4 * (1.0 + (mean * rstd).abs()) + mean.abs() * d_rstd + 4e-6 * ex2.sqrt() * rstd
        worst = max(float(((tab[..., 0] - rstd).abs() / tol_scale).max()), float(((tab[..., 1] + mean * rstd).abs() / tol_shift).max()))
        self.rows.append({"layer": name, "what": "norm table", "worst_over_tol": round(worst, 3), "max_abs": float((tab[..., 0] - rstd).abs().max()),
                          "rms": float(rstd.max()), "frac_over_half_tol": 0.0})
        return worst

    def failures(self):
        return [r for r in self.rows if not (r["worst_over_tol"] <= 1.0)]

    def dump(self, tag):
        path = os.environ.get("LAYER_REPORT")
        if path:
            with open(path, "a") as f:
                for r in self.rows:
                    f.write(json.dumps(dict(r, tag=tag)) + "\n")


def conv(x_nhwc, w, b, stride):
    return nhwc(F.conv2d(nchw(x_nhwc.double()), w, b, stride=stride, padding=w.shape[-1] // 2))


def convt(x_nhwc, w, b):
    return nhwc(F.conv_transpose2d(nchw(x_nhwc.double()), w, b, stride=2, padding=1, output_padding=1))


def check_net(model, x, mode, rep):
    """Walk the forward layer by layer; every layer is recomputed from the engine's own stored inputs."""
    model.precision = mode
    logits, t = model.traced_forward(x)
    t = {k: v.cpu() for k, v in t.items()}
    net = Net(model)
    nl, ch = net.nl, net.ch
    img = x.cpu()
    a_in = lambda v: bf16r(v)                     # noqa: E731  every 16-bit kernel stages bf16 activations

    def conv_layer(name, inp_act, w, b, stride, out, tab, f8=False, transposed=False, out_ulp=BF16_ULP, wq=True):
        """inp_act: fp32 activated input (or the stored e4m3 tensor when f8); out: stored output; tab: stored table or None."""
        if f8:
            qw, scale = fp8_weight(w, 1 if transposed else 0)
            acc = (convt(inp_act, qw, None) if transposed else conv(inp_act, qw, None, stride))
            acc = acc * scale.reshape(1, 1, 1, -1) + b.reshape(1, 1, 1, -1)
        else:
            ww = bf16r(w) if wq else w
            acc = convt(a_in(inp_act), ww, b) if transposed else conv(a_in(inp_act), ww, b, stride)
        rep.compare(name, out, acc, out_ulp)
        if tab is not None:
            rep.compare_table(name, tab[:, :acc.shape[-1]], acc)
        return acc

    # ---- encoder ----
    prev = None                                                       # stored X of the level above (bf16)
    for d in range(nl):
        pfx, ru, c = f"down{d}.", net.down(d), ch[d]
        w0, b0 = net.w(ru + ".conv.unit0.conv")
        wr, br = net.w(ru + ".residual")
        E, X = t[pfx + "E"], t.get(pfx + "X")
        f8_0 = mode == "fp8" and d > 0 and min(ch[d - 1], c) >= 128
        if d == 0:
            src = nhwc(img)
            if E.shape[-1] == c:                                      # y0 alone; the residual branch is recomputed by the sum launch
                conv_layer(pfx + "unit0(first)", src, w0, b0, 2, E, t[pfx + "tab0"])
            else:
                conv_layer(pfx + "unit0||res(first)", src, torch.cat([w0, wr]), torch.cat([b0, br]), 2, E, None)
        else:
            src = t[f"down{d - 1}.Xq"] if f8_0 else prev
            acc = conv_layer(pfx + "unit0||res", src, torch.cat([w0, wr]), torch.cat([b0, br]), 2, E, None, f8=f8_0)
            rep.compare_table(pfx + "unit0||res", t[pfx + "tab0"][:, :c], acc[..., :c])
        f8_1 = mode == "fp8" and c >= 128
        a0 = activated(E[..., :c], t[pfx + "tab0"][:, :c], net.slope(ru + ".conv.unit0"))
        if f8_1:
            rep.compare(pfx + "act_quant(E)", t[pfx + "Aq"], a0, E4M3_ULP, "e4m3 activations")
        w1, b1 = net.w(ru + ".conv.unit1.conv")
        conv_layer(pfx + "unit1", t[pfx + "Aq"] if f8_1 else a0, w1, b1, 1, t[pfx + "Y1"], t[pfx + "tab1"], f8=f8_1)
        a1 = activated(t[pfx + "Y1"], t[pfx + "tab1"], net.slope(ru + ".conv.unit1"))
        if E.shape[-1] == c:                                          # first unit: X = residual conv of the image + act(Y1)
            s = conv(a_in(nhwc(img)), bf16r(wr), br, 2) + a1.double()
        else:
            s = a1.double() + E[..., c:].double()
        if pfx + "X" in t:
            rep.compare(pfx + "sum", X, s, BF16_ULP, "residual sum")
        if pfx + "Xq" in t:
            rep.compare(pfx + "sum(e4m3)", t[pfx + "Xq"], s, E4M3_ULP, "residual sum")
        prev = X if pfx + "X" in t else None
    # ---- bottom ----
    ru, c, cp = net.bottom(), ch[nl], ch[nl - 1]
    f8_b = mode == "fp8" and min(cp, c) >= 128
    src = t[f"down{nl - 1}.Xq"] if f8_b else prev
    w0, b0 = net.w(ru + ".conv.unit0.conv")
    conv_layer("bottom.unit0", src, w0, b0, 1, t["bottom.Y0"], t["bottom.tab0"], f8=f8_b)
    wr, br = net.w(ru + ".residual")
    conv_layer("bottom.res1x1", src, wr, br, 1, t["bottom.R"], None, f8=f8_b)
    a0 = activated(t["bottom.Y0"], t["bottom.tab0"], net.slope(ru + ".conv.unit0"))
    f8_1 = mode == "fp8" and c >= 128
    if f8_1:
        rep.compare("bottom.act_quant(Y0)", t["bottom.Aq"], a0, E4M3_ULP, "e4m3 activations")
    w1, b1 = net.w(ru + ".conv.unit1.conv")
    conv_layer("bottom.unit1", t["bottom.Aq"] if f8_1 else a0, w1, b1, 1, t["bottom.Y1"], t["bottom.tab1"], f8=f8_1)
    s = activated(t["bottom.Y1"], t["bottom.tab1"], net.slope(ru + ".conv.unit1")).double() + t["bottom.R"].double()
    if "bottom.S" in t:
        rep.compare("bottom.sum", t["bottom.S"], s, BF16_ULP, "residual sum")
    if "bottom.Sq" in t:
        rep.compare("bottom.sum(e4m3)", t["bottom.Sq"], s, E4M3_ULP, "residual sum")
    sub16 = t.get("bottom.S")
    sub8 = t.get("bottom.Sq")
    sub_exact = None                                                  # a sum the consumer computes itself (never stored)
    c_sub = c
    # ---- decoder ----
    for d in range(nl - 1, -1, -1):
        pfx, up, c = f"up{d}.", net.up(d), (ch[d - 1] if d > 0 else 3)
        top = d == 0
        wt, bt = net.w(up + ".0.conv")                                # ConvTranspose2d weight [cin, cout, 3, 3]
        f8_t = mode == "fp8" and not top and min(ch[d], c_sub) >= 128 and c >= 128
        skip = t[f"down{d}.Xq"] if f8_t else t[f"down{d}.X"]
        below = sub8 if f8_t else (sub16 if sub16 is not None else sub_exact)
        assert below is not None, (pfx, "no stored sub input")
        cat = torch.cat([skip.double(), below.double()], -1)
        U = t[pfx + "U"]
        if top:
            # 3-channel top layer: fp32 output; on the bf16 datapath its GEMM runs on the bf16 pipe
            conv_layer(pfx + "convT(top)", cat, wt, bt, None, U, t[pfx + "tabt"], transposed=True, out_ulp=2.0 ** -17)
            au = activated(U, t[pfx + "tabt"], net.slope(up + ".0"))
            wl, bl = net.w(up + ".1.conv.unit0.conv")
            # ResidualUnit(3 -> 3, subunits 1, last_conv_only): conv(x) + x with an identity residual; fp32 VALU arithmetic on fp32 inputs
            ref = conv(au.double(), wl, bl, 1) + au.double()
            rep.compare_abs("logits(conv_last)", nhwc(logits.cpu()), ref, 2e-5, "logits")
            break
        conv_layer(pfx + "convT", cat, wt, bt, None, U, t[pfx + "tabt"], f8=f8_t, transposed=True)
        au = activated(U, t[pfx + "tabt"], net.slope(up + ".0"))
        f8_u = mode == "fp8" and c >= 128
        if f8_u:
            rep.compare(pfx + "act_quant(U)", t[pfx + "Uq"], au, E4M3_ULP, "e4m3 activations")
        wu, bu = net.w(up + ".1.conv.unit0.conv")
        conv_layer(pfx + "unit0", t[pfx + "Uq"] if f8_u else au, wu, bu, 1, t[pfx + "V"], t[pfx + "tabu"], f8=f8_u)
        s = activated(t[pfx + "V"], t[pfx + "tabu"], net.slope(up + ".1.conv.unit0")).double() + au.double()
        sub16, sub8, sub_exact = t.get(pfx + "D"), t.get(pfx + "Dq"), None
        if sub16 is not None:
            rep.compare(pfx + "sum", sub16, s, BF16_ULP, "residual sum")
        if sub8 is not None:
            rep.compare(pfx + "sum(e4m3)", sub8, s, E4M3_ULP, "residual sum")
        if sub16 is None and sub8 is None:
            sub_exact = s.float()                                     # two-tensor source: the consumer adds in fp32, then rounds while staging
        c_sub = c
    return rep


CASES = [
    # (mode, channels, n, size, sparse): 32-base = BASELINE configs[1]/[2] (rows32 / rows64 / stride-2 streaming / block kernels, record first
    # layer); 64-base = configs[4] (64-channel first layer, fp8 from level 1 on); sparse = a 22 x 13 image in a zero window (round-5 sweep: the
    # whole-net fp8 logits of such a window were 0.53 from the fp32 oracle - the layers below show that this is e4m3 on near-constant maps, not a kernel)
    ("bf16", (32, 64, 128, 256, 512), 3, 64, False),
    ("bf16", (64, 128, 256, 512, 1024), 2, 64, False),
    ("bf16", (16, 32, 64, 128, 256), 2, 96, False),
    ("fp8", (32, 64, 128, 256, 512), 3, 64, False),
    ("fp8", (64, 128, 256, 512, 1024), 2, 64, False),
    ("bf16", (32, 64, 128, 256), 2, 128, True),
    ("fp8", (32, 64, 128, 256), 2, 128, True),
    # 1 040 windows of 32 x 32: launches of >= 1 024 blocks, where the round-5 streaming kernels (stride-2 pair, 64 + 64 -> 32 transposed conv) take over
    ("bf16", (32, 64, 128, 256, 512), 1040, 32, False),
    ("fp8", (32, 64, 128, 256, 512), 1040, 32, False),
    # 520 windows of 64 x 64: the 64-channel stride-2 pair (shared-window streaming kernel) reaches 1 040 blocks
    ("bf16", (32, 64, 128, 256, 512), 520, 64, False),
]


@pytest.mark.parametrize("mode,channels,n,size,sparse", CASES)
def test_every_layer_matches_its_own_cpu_recomputation(mode, channels, n, size, sparse):
    import cellseg_amd as ca
    dev = _dev()
    torch.manual_seed(11)
    model = ca.UNet(channels=channels, strides=(2,) * (len(channels) - 1), num_res_units=2).to(dev).eval()
    with torch.no_grad():                                              # decisive PReLU slopes and non-trivial biases
        for k, p in model.named_parameters():
            if k.endswith("adn.A.weight"):
                p.fill_(0.1 + 0.3 * torch.rand(()).item())
            elif k.endswith("bias"):
                p.uniform_(-0.2, 0.2)
    x = torch.rand(n, 3, size, size, generator=torch.Generator().manual_seed(5)).to(dev)
    if sparse:                                 # a small image inside a zero-padded window (the tiler's padding of images below the window size): most
        keep = torch.zeros_like(x)            # of every map is constant, InstanceNorm variances are tiny and every rounding is amplified downstream -
        keep[:, :, 20:42, 30:43] = x[:, :, 20:42, 30:43]     # the layers themselves must still be one rounding step from their recomputation
        x = keep
    rep = check_net(model, x, mode, Report())
    rep.dump(f"{mode}-{channels[0]}-{size}" + ("-sparse" if sparse else ""))
    bad = rep.failures()
    assert not bad, json.dumps(bad, indent=1)
    assert len(rep.rows) >= 30
