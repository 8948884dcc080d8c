"""``UNet`` - drop-in for ``monai.networks.nets.UNet`` as the reference instantiates it.

Reference call sites: /root/reference/baseline/predict.py:45-52, :80-81, :85, :109 and
/root/reference/baseline/model_training_3class.py:201-208, :236, :248, :255, :271.

Same constructor arguments, same ``forward`` contract ([N,3,H,W] float -> [N,3,H,W] logits), same
``state_dict`` key names (``model.0.conv.unit0.conv.weight`` ...), so a MONAI checkpoint loads with
``model.load_state_dict(ckpt['model_state_dict'])``.  PyTorch only stores the parameters; all arithmetic
runs in libcellseg_hip.so (hand-written gfx950 kernels).  There is no CPU path: calling the model on a
CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Dict, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, lib

_NORMS = {"instance": 0, "batch": 1}
_ACTS = {"prelu": 0, "relu": 1}


def _norm_name(v) -> str:
    name = (v[0] if isinstance(v, (tuple, list)) else v)
    name = str(getattr(name, "value", name)).lower()
    if name not in _NORMS:
        raise ValueError(f"norm '{v}' is not supported (instance | batch)")
    return name


def _act_name(v) -> str:
    name = (v[0] if isinstance(v, (tuple, list)) else v)
    name = str(getattr(name, "value", name)).lower()
    if name not in _ACTS:
        raise ValueError(f"act '{v}' is not supported (prelu | relu)")
    return name


class _Engine:
    """Owns one ``cs_unet`` handle (packed weights live on the device that was current at creation)."""

    def __init__(self, channels, num_res_units, norm, act, in_channels, out_channels):
        arr = (C.c_int * len(channels))(*channels)
        h = C.c_void_p()
        check(lib.cs_unet_create(C.byref(h), in_channels, out_channels, arr, len(channels), num_res_units,
                                 _NORMS[norm], _ACTS[act]), "cs_unet_create")
        self.handle = h

    def param_table(self):
        out = []
        key = C.create_string_buffer(160)
        shape = (C.c_int64 * 4)()
        for i in range(check(lib.cs_unet_num_params(self.handle))):
            rank = check(lib.cs_unet_param_info(self.handle, i, key, shape), "cs_unet_param_info")
            out.append((key.value.decode(), tuple(int(shape[k]) for k in range(rank))))
        return out

    def __del__(self):
        try:
            if self.handle:
                lib.cs_unet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _host_param_table(in_ch, out_ch, channels, nru, norm, act):
    """MONAI state-dict layout computed on the host (mirrors engine.hip build_net)."""
    tab = []

    def adn(prefix, c):
        if norm == "batch":
            for k in ("weight", "bias", "running_mean", "running_var"):
                tab.append((f"{prefix}.N.{k}", (c,)))
            tab.append((f"{prefix}.N.num_batches_tracked", ()))
        if act == "prelu":
            tab.append((f"{prefix}.A.weight", (1,)))

    def down(prefix, cin, c, stride):
        for su in range(nru):
            tab.append((f"{prefix}.conv.unit{su}.conv.weight", (c, cin if su == 0 else c, 3, 3)))
            tab.append((f"{prefix}.conv.unit{su}.conv.bias", (c,)))
            adn(f"{prefix}.conv.unit{su}.adn", c)
        if stride != 1 or cin != c:
            k = 3 if stride != 1 else 1
            tab.append((f"{prefix}.residual.weight", (c, cin, k, k)))
            tab.append((f"{prefix}.residual.bias", (c,)))

    def up(prefix, cin, cout, top):
        tab.append((f"{prefix}.0.conv.weight", (cin, cout, 3, 3)))
        tab.append((f"{prefix}.0.conv.bias", (cout,)))
        adn(f"{prefix}.0.adn", cout)
        tab.append((f"{prefix}.1.conv.unit0.conv.weight", (cout, cout, 3, 3)))
        tab.append((f"{prefix}.1.conv.unit0.conv.bias", (cout,)))
        if not top:
            adn(f"{prefix}.1.conv.unit0.adn", cout)

    def block(prefix, inc, outc, chs, top):
        c = chs[0]
        down(f"{prefix}.0", inc, c, 2)
        if len(chs) > 2:
            block(f"{prefix}.1.submodule", c, c, chs[1:], False)
            upc = 2 * c
        else:
            down(f"{prefix}.1.submodule", c, chs[1], 1)
            upc = c + chs[1]
        up(f"{prefix}.2", upc, outc, top)

    block("model", in_ch, out_ch, tuple(channels), True)
    return tab


_POISON = os.environ.get("CS_POISON_WORKSPACE", "0") == "1"
_GUARD = 1 << 20


class _UNetFunction(torch.autograd.Function):
    """Autograd bookkeeping only: forward and backward both run in libcellseg_hip.so.

    forward keeps the engine's activation workspace alive (ctx) until backward; backward fills ONE flat fp32 gradient
    buffer (the data-parallel bucket) and hands PyTorch per-parameter views of it."""

    @staticmethod
    def forward(ctx, module, x, *params):
        out, ws = module._forward_engine(x, tag="train", train=True)
        ctx.module, ctx.ws, ctx.x, ctx.shape, ctx.generation = module, ws, x, tuple(x.shape), module._generation
        return out

    @staticmethod
    def backward(ctx, grad_out):
        module = ctx.module
        if ctx.generation != module._generation:
            raise _lib.CellSegHipError(
                "backward of a forward whose activations have been overwritten: the engine keeps one activation tape per "
                "model, so every grad-enabled forward must be followed by its backward before the next grad-enabled forward "
                "(for gradient accumulation call backward() per micro-batch)")
        flat = module._backward_engine(grad_out, ctx.shape)
        grads = []
        for key in module._param_keys:
            off, numel, shape = module._grad_slots[key]
            grads.append(flat[off:off + numel].view(shape))
        return (None, None, *grads)


class UNet(nn.Module):
    def __init__(self, spatial_dims: int = 2, in_channels: int = 3, out_channels: int = 3,
                 channels: Sequence[int] = (16, 32, 64, 128, 256), strides: Sequence[int] = (2, 2, 2, 2),
                 kernel_size: int = 3, up_kernel_size: int = 3, num_res_units: int = 2, act="PRELU", norm="INSTANCE",
                 dropout: float = 0.0, bias: bool = True, adn_ordering: str = "NDA"):
        super().__init__()
        if spatial_dims != 2:
            raise ValueError("cellseg_amd.UNet is the reference's 2-D network (spatial_dims=2)")
        if in_channels != 3 or out_channels != 3:
            raise ValueError("cellseg_amd.UNet implements the reference's 3-channel -> 3-class contract")
        if tuple(strides) != (2,) * (len(channels) - 1):
            raise ValueError("every stride must be 2 and len(strides) == len(channels) - 1")
        if kernel_size != 3 or up_kernel_size != 3 or not bias or adn_ordering != "NDA" or dropout not in (0, 0.0, None):
            raise ValueError("only MONAI's defaults (kernel 3, bias, dropout 0, ordering 'NDA') are implemented")
        if not (1 <= num_res_units <= 4):
            raise ValueError("num_res_units must be in 1..4")
        if any(c % 16 or c < 16 for c in channels):
            raise ValueError("channels must be multiples of 16 (MFMA tile width)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.channels, self.strides, self.num_res_units = tuple(channels), tuple(strides), num_res_units
        self.norm_name, self.act_name = _norm_name(norm), _act_name(act)
        self._keys = []
        for key, shape in _host_param_table(in_channels, out_channels, self.channels, num_res_units,
                                            self.norm_name, self.act_name):
            self._register(key, shape)
        self._engine = None
        self._engine_device = None
        self._uploaded: Dict[str, Tuple[int, int]] = {}
        self._ws: Dict[Tuple, torch.Tensor] = {}
        self._param_keys = [k for k, _ in self.named_parameters()]
        self._grad_slots: Dict[str, Tuple[int, int, Tuple[int, ...]]] = {}
        self._generation = 0           # id of the grad-enabled forward that owns the engine's activation tape
        self.flat_grad = None          # the flat gradient bucket of the last backward (what data parallel all-reduces)
        self.precision = "fp32"        # "fp32" | "fp16x3", "bf16x6" (fp32-grade) | "bf16x3" | "bf16" (bf16 datapath) | "fp8": arithmetic of the convs

    # ---- parameter registration under MONAI's dotted names -------------------------------------------
    def _register(self, key: str, shape) -> None:
        parts = key.split(".")
        mod: nn.Module = self
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        leaf = parts[-1]
        if leaf in ("running_mean", "running_var", "num_batches_tracked"):
            init = {"running_mean": torch.zeros(shape), "running_var": torch.ones(shape),
                    "num_batches_tracked": torch.tensor(0, dtype=torch.long)}[leaf]
            mod.register_buffer(leaf, init)
        else:
            mod.register_parameter(leaf, nn.Parameter(self._init(key, shape)))
        self._keys.append(key)

    @staticmethod
    def _init(key: str, shape) -> torch.Tensor:
        """torch's default initialisers for Conv2d / ConvTranspose2d / PReLU / BatchNorm2d."""
        if key.endswith(".A.weight"):
            return torch.full(shape, 0.25)
        if ".N." in key:
            return torch.ones(shape) if key.endswith("weight") else torch.zeros(shape)
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            bound = 1.0 / math.sqrt(fan_in)          # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            w = torch.empty(shape).uniform_(-bound, bound)
            UNet._last_fan_in = fan_in
            return w
        bound = 1.0 / math.sqrt(getattr(UNet, "_last_fan_in", 1))
        return torch.empty(shape).uniform_(-bound, bound)

    # ---- engine plumbing ------------------------------------------------------------------------------
    def _tensor(self, key: str) -> torch.Tensor:
        obj = self
        for p in key.split("."):
            obj = getattr(obj, p)
        return obj

    def engine(self, train: bool = False) -> _Engine:
        """Create (once per device) the C engine and upload every tensor whose storage/version changed.
        `train`: also pack the data-gradient weight layouts (the caller is about to record a tape for backward)."""
        p0 = next(self.parameters())
        _lib.require_gpu(p0)
        if self._engine is None or self._engine_device != p0.device:
            with torch.cuda.device(p0.device):
                self._engine = _Engine(self.channels, self.num_res_units, self.norm_name, self.act_name,
                                       self.in_channels, self.out_channels)
            self._engine_device = p0.device
            self._uploaded.clear()
            self._pack_state = (0, False, False)
            self._ws.clear()
            self._grad_slots = {}
            for i, (key, shape) in enumerate(self._engine.param_table()):
                off = lib.cs_unet_grad_offset(self._engine.handle, i)
                numel = 1
                for d in shape:
                    numel *= d
                self._grad_slots[key] = (int(off), numel, tuple(shape))
        planes = {"fp32": 0, "bf16": 1, "bf16x3": 2, "bf16x6": 3, "fp16x3": 4, "fp8": 5}.get(self.precision)
        if planes is None:
            raise ValueError(f"precision '{self.precision}' (fp32 | fp16x3 | bf16x6 | bf16x3 | bf16 | fp8)")
        if planes == 5 and train:
            raise _lib.CellSegHipError("precision 'fp8' is an inference mode: training runs in fp32 / bf16")
        check(lib.cs_unet_set_precision(self._engine.handle, planes), "cs_unet_set_precision")
        want_train = bool(train)
        check(lib.cs_unet_set_training(self._engine.handle, int(want_train)), "cs_unet_set_training")
        if self.norm_name == "batch":                 # nn.BatchNorm2d semantics: batch statistics in train mode, running ones in eval
            check(lib.cs_unet_set_batch_stats(self._engine.handle, int(self.training)), "cs_unet_set_batch_stats")
        # The engine packs the split-bf16 / data-gradient weight layouts only when asked.  `_pack_state` says which of them
        # EVERY uploaded tensor currently has; a missing one forces a full re-upload under the new flags.
        fmt = 3 if planes == 5 else 2 if planes == 4 else 1 if planes > 0 else 0    # low-precision packing the kernels will read
        # third flag: the inference-only packings (Winograd F(4x4,3x3) weights of the fp32 path) are present.  A training-mode upload
        # skips them - a training step re-packs every weight, and 36 fp64-transformed positions per 3x3 kernel cost more per step than
        # they save - so the first inference call after training re-uploads once (and the first training call after that again)
        need = (fmt, want_train, not want_train)
        state = getattr(self, "_pack_state", (0, False, False))
        if (need[0] and state[0] != need[0]) or (need[1] and not state[1]) or (need[2] and not state[2]):
            self._uploaded.clear()
            state = need
        uploaded_any = False
        alive = []                                                   # sources must outlive the batched pack kernel's enqueue
        with torch.cuda.device(p0.device):                          # the handle lives on the parameters' device, whatever is current
            stream = _lib.stream_ptr()
            check(lib.cs_unet_load_begin(self._engine.handle), "cs_unet_load_begin")
            try:
                for key in self._keys:
                    if key.endswith("num_batches_tracked"):
                        continue
                    t = self._tensor(key)
                    tag = (t.data_ptr(), t._version)
                    if self._uploaded.get(key) != tag:
                        src = t.detach()
                        if src.dtype != torch.float32 or not src.is_contiguous():
                            src = src.float().contiguous()
                        alive.append(src)
                        check(lib.cs_unet_load_param(self._engine.handle, key.encode(), C.c_void_p(src.data_ptr()), src.numel(),
                                                     stream), f"cs_unet_load_param({key})")
                        self._uploaded[key] = tag
                        uploaded_any = True
            finally:
                check(lib.cs_unet_load_end(self._engine.handle, stream), "cs_unet_load_end")   # one kernel packs everything recorded
        del alive
        self._pack_state = (state[0] and need[0], state[1] and need[1], state[2] and need[2]) if uploaded_any else state
        return self._engine

    def invalidate_weights(self) -> None:
        """Force a re-upload of every tensor at the next call.  Uploads are keyed on (data_ptr, tensor version); an in-place edit
        through ``p.data`` (EMA, manual clipping) does not bump the version, so it needs this call - ``optimizer.step()``,
        ``load_state_dict`` and ordinary in-place ops under ``no_grad`` are picked up by themselves."""
        self._uploaded.clear()

    def workspace(self, tag, nbytes: int, device) -> torch.Tensor:
        ws = self._ws.get(tag)
        if ws is None or ws.numel() < nbytes or ws.device != device:
            self._ws = {k: v for k, v in self._ws.items() if k[0] != tag[0]}    # one buffer per kind
            if _POISON:                          # diagnostics: guard bands on both sides, checked by check_guards()
                base = torch.full((nbytes + 2 * _GUARD,), 0x5A, dtype=torch.uint8, device=device)
                self._ws_base = getattr(self, "_ws_base", {})
                self._ws_base[tag] = base
                ws = base[_GUARD:_GUARD + nbytes]
            else:
                ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._ws[tag] = ws
        return ws

    def check_guards(self) -> None:
        """CS_POISON_WORKSPACE=1 only: raise if a kernel wrote outside one of the engine workspaces."""
        for tag, base in getattr(self, "_ws_base", {}).items():
            n = base.numel() - 2 * _GUARD
            lo, hi = base[:_GUARD], base[_GUARD + n:]
            if not (bool((lo == 0x5A).all()) and bool((hi == 0x5A).all())):
                blo = (lo != 0x5A).nonzero().flatten(); bhi = (hi != 0x5A).nonzero().flatten()
                raise _lib.CellSegHipError(f"workspace {tag}: guard band overwritten ({blo.numel()} bytes below, first at -{_GUARD - int(blo[0]) if blo.numel() else 0}; "
                                           f"{bhi.numel()} bytes above, first at +{int(bhi[0]) if bhi.numel() else 0})")

    def _forward_engine(self, x: torch.Tensor, tag: str = "fwd", train: bool = False):
        n, _, h, w = x.shape
        eng = self.engine(train)
        self._generation += 1                         # the engine keeps ONE tape (cs_unet_forward records it): this forward owns it now
        with torch.cuda.device(x.device):
            nbytes = lib.cs_unet_workspace_bytes(eng.handle, n, h, w)
            if nbytes == 0:
                raise _lib.CellSegHipError(f"cs_unet_workspace_bytes: {lib.cs_last_error().decode()}")
            ws = self.workspace((tag, n, h, w), nbytes, x.device)
            out = torch.empty((n, h, w, self.out_channels), dtype=torch.float32, device=x.device)
            if _POISON:
                ws.fill_(255)
                out.fill_(float("nan"))
            sn, sc, sh, sw = x.stride()
            check(lib.cs_unet_forward(eng.handle, C.c_void_p(x.data_ptr()), sn, sc, sh, sw, n, h, w,
                                      C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                                      _lib.stream_ptr()), "cs_unet_forward")
            if self.norm_name == "batch" and self.training:
                self._pull_running_stats(eng)
        return out.permute(0, 3, 1, 2), ws        # NCHW view of the NHWC result, like a channels_last tensor

    def traced_forward(self, x: torch.Tensor):
        """Inference forward that also returns every intermediate tensor of the engine (``cs_unet_trace``): -> (logits [N,3,H,W],
        {name: tensor}) with activations as NHWC views of the workspace in their stored element type (float32 / bfloat16 /
        float8_e4m3fn, channel padding cut off) and lazy-norm tables as float32 [N, C, 2] = (scale, shift).  For the per-layer parity
        tests (the reference's layers are torch.nn modules one would hook: baseline/predict.py:45-52); not a fast path."""
        _lib.require_gpu(x)
        eng = self.engine(False)
        check(lib.cs_unet_trace(eng.handle, 1), "cs_unet_trace")
        try:
            with torch.no_grad():
                out, ws = self._forward_engine(x.float(), tag="trace")
            taps = {}
            name, info = C.create_string_buffer(64), (C.c_int64 * 8)()
            for i in range(check(lib.cs_unet_trace_count(eng.handle), "cs_unet_trace_count")):
                check(lib.cs_unet_trace_entry(eng.handle, i, name, info), "cs_unet_trace_entry")
                off, n, h, w, c, pitch, kind = (int(v) for v in info[:7])
                if kind == 3:
                    taps[name.value.decode()] = ws[off:off + n * c * 8].view(torch.float32).view(n, c, 2)
                    continue
                dt = (torch.float32, torch.bfloat16, torch.float8_e4m3fn)[kind]
                nbytes = n * h * w * pitch * dt.itemsize
                taps[name.value.decode()] = ws[off:off + nbytes].view(dt).view(n, h, w, pitch)[..., :c]
        finally:
            check(lib.cs_unet_trace(eng.handle, 0), "cs_unet_trace")
        return out, taps

    def _pull_running_stats(self, eng) -> None:
        """Train-mode BatchNorm: the engine has just updated its running_mean / running_var; mirror them into the module's buffers
        (what state_dict() saves) with one batched copy kernel, and count the batch like nn.BatchNorm2d does."""
        stream = _lib.stream_ptr()
        check(lib.cs_unet_load_begin(eng.handle), "cs_unet_load_begin")
        try:
            for key in self._keys:
                if key.endswith("running_mean") or key.endswith("running_var"):
                    t = self._tensor(key)
                    check(lib.cs_unet_read_param(eng.handle, key.encode(), C.c_void_p(t.data_ptr()), t.numel(), stream),
                          f"cs_unet_read_param({key})")      # raw-pointer write: the tensor version (and the upload cache) stay valid
        finally:
            check(lib.cs_unet_load_end(eng.handle, stream), "cs_unet_load_end")
        for key in self._keys:
            if key.endswith("num_batches_tracked"):
                self._tensor(key).add_(1)

    def _backward_engine(self, grad_out: torch.Tensor, shape) -> torch.Tensor:
        n, _, h, w = shape
        eng = self._engine
        g = grad_out.permute(0, 2, 3, 1).contiguous().float()             # NHWC, as the engine wrote the logits
        with torch.cuda.device(g.device):
            nbytes = lib.cs_unet_backward_workspace_bytes(eng.handle, n, h, w)
            if nbytes == 0:
                raise _lib.CellSegHipError(f"cs_unet_backward_workspace_bytes: {lib.cs_last_error().decode()}")
            ws = self.workspace(("bwd", n, h, w), nbytes, g.device)
            flat = torch.empty(int(lib.cs_unet_grad_numel(eng.handle)), dtype=torch.float32, device=g.device)
            if _POISON:                          # diagnostics: a kernel that reads scratch memory it has not written turns the result into NaN
                ws.fill_(255)
                flat.fill_(float("nan"))
            check(lib.cs_unet_backward(eng.handle, C.c_void_p(g.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                                       C.c_void_p(flat.data_ptr()), _lib.stream_ptr()), "cs_unet_backward")
        self.flat_grad = flat
        return flat

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _lib.require_gpu(x)
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected [N,{self.in_channels},H,W], got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            x = x.float()
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if needs_grad:
            if self.norm_name == "batch" and not self.training:
                raise _lib.CellSegHipError("backward through eval-mode BatchNorm (running statistics) is not implemented: call model.train()")
            params = [p for _, p in self.named_parameters()]
            return _UNetFunction.apply(self, x, *params)
        return self._forward_engine(x)[0]
