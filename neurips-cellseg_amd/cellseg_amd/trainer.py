"""``TrainStep`` - the loop body of the reference's training script as one lean launch sequence, optionally a captured hipGraph.

Reference: /root/reference/baseline/model_training_3class.py:250-262

    optimizer.zero_grad(); outputs = model(inputs); loss = loss_function(outputs, labels_onehot); loss.backward(); optimizer.step()

At the reference's own setting (16...256 channels, batch 8) that body is ~175 kernel launches of a few microseconds each; issued
through autograd, with a 63-tensor upload check and the optimiser's bookkeeping around them, the HOST sets the pace (2.25 ms of
enqueue per step for 1.8 ms of GPU work).  This class issues the same C-ABI calls - ``cs_unet_forward``,
``cs_dice_ce_forward / _backward``, ``cs_unet_backward`` (side-stream weight gradients included), ``cs_adamw_step_dev``,
``cs_unet_repack`` (a frozen re-pack table: two launches, no host table) - on buffers that stay allocated: 0.8 ms of enqueue,
the step is GPU-bound again (measured on one MI355X, tests/diagnostics/gpu_train_step_modes.py: autograd loop 2.25 ms,
this class 1.83 ms at 16...256 x 8; 6.18 vs 6.22 ms at 32...512 x 32, which was GPU-bound already).

``capture=True`` records that sequence once into a hipGraph (``cs_graph_begin / _end``) and replays it every step (two graphs
around the flat-bucket all-reduce under data parallel): the host enqueue drops to 0.55 ms and no longer depends on Python, but on
ROCm 7.2 the replayed graph runs its 186 nodes SLOWER on the GPU than the same launches from streams (2.13 vs 1.83 ms, 6.67 vs
6.22 ms: the graph's branch streams and per-node barriers cost more than they hide), so it is off by default - a choice for
hosts whose cores are busy elsewhere.  Same kernels in the same order either way: the gradients are bit for bit those of the
launch-by-launch sequence (``tests/test_gpu_round3.py``).

    step = ca.TrainStep(model, loss_function, optimizer)
    for batch in loader:
        loss = step(batch["img"], ca.one_hot(batch["label"], 3))        # a 0-dim device tensor, like the reference's `loss`

PyTorch is plumbing here: it owns the parameter tensors, the static buffers and the stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import check, lib
from .losses import DiceCELoss
from .optim import AdamW
from .unet import UNet


class _Shape:
    """Static buffers + captured graphs for one batch shape."""

    def __init__(self):
        self.x = self.t = self.logits = self.dlogits = self.ws_f = self.ws_b = self.ws_l = self.loss = self.loss_out = None
        self.graph_a = self.graph_b = None
        self.eager_steps = 0


class TrainStep:
    def __init__(self, model: UNet, loss_function: DiceCELoss, optimizer: AdamW, *, capture: bool = False, group=None,
                 sync_gradients: Optional[bool] = None):
        if not isinstance(model, UNet) or not isinstance(loss_function, DiceCELoss) or not isinstance(optimizer, AdamW):
            raise TypeError("TrainStep drives cellseg_amd.UNet + cellseg_amd.DiceCELoss + cellseg_amd.AdamW")
        if optimizer.model is not model:
            raise ValueError("the optimizer belongs to another model")
        self.model, self.loss_function, self.optimizer = model, loss_function, optimizer
        self.capture, self.group = bool(capture), group
        self._sync = sync_gradients
        self._shapes: Dict[Tuple, _Shape] = {}
        self._bound = None                 # (param pointers, precision, training) the engine's frozen re-pack table was recorded for
        self._opt_bound = None             # (moment buffers, offset table) of the optimiser the launch sequence was bound to
        self._hyper = self._step_dev = None
        self._hyper_host = None
        self.flat = None
        self._tensors = None               # the engine's parameter tensors in cs_unet_param_info order
        self._sig = None                   # their (data_ptr, version) pairs + precision at the last full check
        self._stream = None                # the step runs (and is captured) on a stream of its own: the legacy default stream cannot be captured
        self._nbt = []
        self._expected_step = -1           # optimizer._step after this object's last step (a load_state_dict in between is noticed)

    # ---- one-time bindings ---------------------------------------------------------------------------------------------------
    def _collective(self) -> bool:
        if self._sync is not None:
            return self._sync
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()

    def _signature(self):
        return (tuple((t.data_ptr(), t._version) for t in self._tensors), self.model.precision)

    def _bind(self, dev) -> None:
        """Full check, taken on the first step and whenever a parameter tensor moved or was modified behind this object's back
        (load_state_dict, .to(), manual edits): the usual upload path brings the engine up to date, then the frozen re-pack table
        is (re)recorded for the tensors' current addresses."""
        model, opt = self.model, self.optimizer
        for p in model.parameters():
            if not p.requires_grad:
                raise _lib.CellSegHipError("TrainStep updates every parameter: frozen tensors need the eager loop (cellseg_amd.AdamW.step)")
        eng = model.engine(True)          # creates the handle, sets precision / training mode, uploads whatever changed
        keys = [k for k, _ in eng.param_table()]
        self._tensors = [model._tensor(k) for k in keys]
        for k, t in zip(keys, self._tensors):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                raise _lib.CellSegHipError(f"TrainStep: parameter {k} must be a contiguous fp32 tensor on {dev}")
        self._nbt = [model._tensor(k) for k in model._keys if k.endswith("num_batches_tracked")]
        self._sig = self._signature()
        tag = (tuple(t.data_ptr() for t in self._tensors), model.precision)
        if tag == self._bound:
            # the parameters did not move - but the optimiser's state may have (AdamW.load_state_dict re-allocates the moments and drops
            # its offset table): re-bind it, and forget captured graphs, which replay the OLD moment addresses
            if self.flat is not None and self._opt_tag() != self._opt_bound:
                opt._bind(self.flat, frozenset(model._param_keys))
                self._opt_bound = self._opt_tag()
                for sh in self._shapes.values():
                    self._drop_graphs(sh)
                self._step_dev.fill_(int(opt._step))
            return
        arr = (C.c_void_p * len(self._tensors))(*[t.data_ptr() for t in self._tensors])
        check(lib.cs_unet_bind_params(eng.handle, arr, len(self._tensors), _lib.stream_ptr()), "cs_unet_bind_params")
        self._bound = tag
        for sh in self._shapes.values():   # graphs recorded against the old binding are stale
            self._drop_graphs(sh)
        total = int(lib.cs_unet_grad_numel(eng.handle))
        if self.flat is None or self.flat.numel() != total or self.flat.device != dev:
            self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        model.flat_grad = self.flat
        for key, p in model.named_parameters():           # p.grad are views of the static bucket, as after an eager backward
            off, numel, shape = model._grad_slots[key]
            p.grad = self.flat[off:off + numel].view(shape)
        opt._bind(self.flat, frozenset(model._param_keys))
        self._opt_bound = self._opt_tag()
        if self._hyper is None or self._hyper.device != dev:
            self._hyper = torch.zeros(16, dtype=torch.float32, device=dev)
            self._step_dev = torch.zeros((), dtype=torch.int32, device=dev)
            self._hyper_host = None
        self._step_dev.fill_(int(opt._step))
        model._pack_state = (model._pack_state[0], True, False)     # the frozen re-pack runs in training mode: inference-only packings go stale

    def _opt_tag(self):
        """Identity of the optimiser state a recorded launch sequence reads and writes: the two moment buffers and the offset table."""
        o = self.optimizer
        return (None if o._m is None else o._m.data_ptr(), None if o._v is None else o._v.data_ptr(),
                None if o._table is None else o._table.data_ptr())

    def _push_hyper(self) -> None:
        g = self.optimizer.param_groups[0]
        cur = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))
        if cur != self._hyper_host:
            self._hyper[:5].copy_(torch.tensor(cur, dtype=torch.float32), non_blocking=False)
            self._hyper_host = cur

    @staticmethod
    def _drop_graphs(sh: _Shape) -> None:
        for name in ("graph_a", "graph_b"):
            g = getattr(sh, name)
            if g is not None:
                lib.cs_graph_destroy(g)
                setattr(sh, name, None)
        sh.eager_steps = 0

    def _buffers(self, shape, dev) -> _Shape:
        sh = self._shapes.get(shape)
        if sh is not None:
            return sh
        n, c, h, w = shape
        eng = self.model._engine
        sh = _Shape()
        sh.x = torch.empty(shape, dtype=torch.float32, device=dev)
        sh.t = torch.empty((n, self.model.out_channels, h, w), dtype=torch.float32, device=dev)
        sh.logits = torch.empty((n, h, w, self.model.out_channels), dtype=torch.float32, device=dev)        # NHWC, as the engine writes them
        sh.dlogits = torch.empty_like(sh.logits)
        nf, nb = lib.cs_unet_workspace_bytes(eng.handle, n, h, w), lib.cs_unet_backward_workspace_bytes(eng.handle, n, h, w)
        if nf == 0 or nb == 0:
            raise _lib.CellSegHipError(f"TrainStep: {lib.cs_last_error().decode()}")
        sh.ws_f = torch.empty(nf, dtype=torch.uint8, device=dev)
        sh.ws_b = torch.empty(nb, dtype=torch.uint8, device=dev)
        sh.ws_l = torch.empty(int(lib.cs_dice_ce_workspace_bytes(n, self.model.out_channels)), dtype=torch.uint8, device=dev)
        sh.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self._shapes[shape] = sh
        return sh

    # ---- the launch sequence (identical eager and under capture) ----------------------------------------------------------
    def _front(self, sh: _Shape, x: torch.Tensor, t: torch.Tensor) -> None:
        """forward -> (BatchNorm: running statistics back to the module) -> DiceCE -> its gradient -> backward into the flat bucket."""
        model, lf = self.model, self.loss_function
        h = model._engine.handle
        n, _, hh, ww = x.shape
        s = _lib.stream_ptr()
        sn, sc, shh, sww = x.stride()
        check(lib.cs_unet_forward(h, C.c_void_p(x.data_ptr()), sn, sc, shh, sww, n, hh, ww, C.c_void_p(sh.logits.data_ptr()),
                                  C.c_void_p(sh.ws_f.data_ptr()), sh.ws_f.numel(), s), "cs_unet_forward")
        if model.norm_name == "batch":
            check(lib.cs_unet_readback(h, s), "cs_unet_readback")
        lg = sh.logits.permute(0, 3, 1, 2)                         # the NCHW view the loss is defined on
        ls = (C.c_int64 * 4)(*[int(v) for v in lg.stride()])
        ts = (C.c_int64 * 4)(*[int(v) for v in t.stride()])
        cls = model.out_channels
        # the loss lands in a tensor of its own per step (an allocation out of torch's cache, no launch) - a captured step replays fixed
        # addresses, so it keeps the static scalar and __call__ clones it (one runtime copy kernel + its dispatch gap per step)
        sh.loss_out = sh.loss if self.capture else torch.empty((), dtype=torch.float32, device=sh.loss.device)
        check(lib.cs_dice_ce_forward(C.c_void_p(sh.logits.data_ptr()), ls, C.c_void_p(t.data_ptr()), ts, n, cls, hh, ww, lf.smooth_nr,
                                     lf.smooth_dr, lf.lambda_dice, lf.lambda_ce, C.c_void_p(sh.loss_out.data_ptr()), C.c_void_p(sh.ws_l.data_ptr()),
                                     sh.ws_l.numel(), s), "cs_dice_ce_forward")
        check(lib.cs_dice_ce_backward(C.c_void_p(sh.logits.data_ptr()), ls, C.c_void_p(t.data_ptr()), ts, n, cls, hh, ww, lf.smooth_nr,
                                      lf.smooth_dr, lf.lambda_dice, lf.lambda_ce, C.c_void_p(), C.c_void_p(sh.ws_l.data_ptr()), sh.ws_l.numel(),
                                      C.c_void_p(sh.dlogits.data_ptr()), s), "cs_dice_ce_backward")
        check(lib.cs_unet_backward(h, C.c_void_p(sh.dlogits.data_ptr()), C.c_void_p(sh.ws_b.data_ptr()), sh.ws_b.numel(),
                                   C.c_void_p(self.flat.data_ptr()), s), "cs_unet_backward")

    def _back(self) -> None:
        """AdamW on the flat bucket (device-resident step counter and hyperparameters) -> re-pack of the updated weights."""
        opt = self.optimizer
        s = _lib.stream_ptr()
        check(lib.cs_adamw_step_dev(C.c_void_p(opt._table.data_ptr()), opt._table.shape[0], self.flat.numel(), C.c_void_p(self.flat.data_ptr()),
                                    C.c_void_p(opt._m.data_ptr()), C.c_void_p(opt._v.data_ptr()), C.c_void_p(self._hyper.data_ptr()),
                                    C.c_void_p(self._step_dev.data_ptr()), s), "cs_adamw_step_dev")
        check(lib.cs_unet_repack(self.model._engine.handle, s), "cs_unet_repack")

    def _record(self, fn):
        s = _lib.stream_ptr()
        check(lib.cs_graph_begin(s), "cs_graph_begin")
        try:
            fn()
        except Exception:
            lib.cs_graph_abort(s)
            raise
        g = C.c_void_p()
        check(lib.cs_graph_end(s, C.byref(g)), "cs_graph_end")
        return g

    # ---- one step ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, inputs: torch.Tensor, labels_onehot: torch.Tensor) -> torch.Tensor:
        _lib.require_gpu(inputs)
        model = self.model
        if not model.training:
            raise _lib.CellSegHipError("TrainStep: call model.train() first")
        if inputs.dim() != 4 or inputs.shape[1] != model.in_channels:
            raise ValueError(f"expected [N,{model.in_channels},H,W], got {tuple(inputs.shape)}")
        if tuple(labels_onehot.shape) != (inputs.shape[0], model.out_channels, inputs.shape[2], inputs.shape[3]):
            raise ValueError(f"one-hot labels {tuple(labels_onehot.shape)} do not match inputs {tuple(inputs.shape)}")
        dev = inputs.device
        with torch.cuda.device(dev):
            if not self.capture:
                # the lean launch sequence runs on the caller's stream and reads the caller's tensors in place: no event round trips
                # between steps, no staging copies (40 us of the 1.5 ms batch-8 step)
                return self._step(inputs, labels_onehot, dev)
            if self._stream is None or self._stream.device != dev:
                self._stream = torch.cuda.Stream(dev)
            caller = torch.cuda.current_stream(dev)
            self._stream.wait_stream(caller)
            with torch.cuda.stream(self._stream):
                loss = self._step(inputs, labels_onehot, dev)
            caller.wait_stream(self._stream)
            inputs.record_stream(self._stream)
            labels_onehot.record_stream(self._stream)
            return loss

    def _step(self, inputs, labels_onehot, dev) -> torch.Tensor:
        model = self.model
        if (self._tensors is None or self._signature() != self._sig or self.optimizer._m is None or int(self.optimizer._step) != self._expected_step
                or self._opt_tag() != self._opt_bound        # load_state_dict at the SAME step count: new moment buffers, no offset table
                or not getattr(model, "_pack_state", (0, False, False))[1]):     # (an inference call in between re-packed without the data-gradient layouts)
            self._bound = None
            self._bind(dev)
        h = model._engine.handle
        check(lib.cs_unet_set_training(h, 1), "cs_unet_set_training")
        if model.norm_name == "batch":
            check(lib.cs_unet_set_batch_stats(h, 1), "cs_unet_set_batch_stats")
        sh = self._buffers(tuple(inputs.shape), dev)
        if self.capture:                                        # a captured step replays fixed addresses: stage the batch
            sh.x.copy_(inputs, non_blocking=True)
            sh.t.copy_(labels_onehot, non_blocking=True)
            x, t = sh.x, sh.t
        else:                                                   # (any strides; other dtypes are converted once)
            x = inputs if inputs.dtype == torch.float32 else inputs.float()
            t = labels_onehot if labels_onehot.dtype == torch.float32 else labels_onehot.float()
        self._push_hyper()
        model._generation += 1                                  # an eager backward of an older forward must not run on this tape
        collective = self._collective()
        use_graph = self.capture and sh.eager_steps >= 1 and not lib.cs_prof_is_enabled()
        if not use_graph:
            self._front(sh, x, t)                               # (the first step of a shape runs eagerly: it creates the engine's
            if collective:                                      # side streams, which a capture must not do)
                self._exchange()
            self._back()
            sh.eager_steps += 1
        else:
            if sh.graph_a is None:
                if collective:
                    sh.graph_a, sh.graph_b = self._record(lambda: self._front(sh, x, t)), self._record(self._back)
                else:
                    sh.graph_a = self._record(lambda: (self._front(sh, x, t), self._back()))
            s = _lib.stream_ptr()
            check(lib.cs_graph_launch(sh.graph_a, s), "cs_graph_launch")
            if collective:
                self._exchange()
                check(lib.cs_graph_launch(sh.graph_b, s), "cs_graph_launch")
        self.optimizer._step += 1
        self._expected_step = int(self.optimizer._step)
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)                   # nn.BatchNorm2d counts its train-mode batches
        return sh.loss.clone() if self.capture else sh.loss_out

    def _exchange(self) -> None:
        from . import parallel
        parallel.sync_gradients(self.model, self.group, rebind=False)      # p.grad already are views of the static bucket

    def graph_nodes(self) -> int:
        """Kernel / memset nodes of the captured step(s) (0 before the capture)."""
        return sum(int(lib.cs_graph_num_nodes(g)) for sh in self._shapes.values() for g in (sh.graph_a, sh.graph_b) if g is not None)

    def __del__(self):
        try:
            for sh in self._shapes.values():
                self._drop_graphs(sh)
        except Exception:
            pass
