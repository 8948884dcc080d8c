"""``AdamW`` - the reference's optimiser (``torch.optim.AdamW(model.parameters(), args.initial_lr)``,
/root/reference/baseline/model_training_3class.py:237, ``optimizer.step()`` at :262) as ONE HIP launch per step.

Row O of SURVEY.md 8(a).  The engine's backward leaves every parameter gradient in one flat fp32 bucket
(``model.flat_grad``, the data-parallel all-reduce bucket); the two moment buffers use the same offsets, so the whole update is
one element-wise kernel (``cs_adamw_step``) instead of torch's dozen multi-tensor launches over 63 tensors.  Same update rule
and defaults as ``torch.optim.AdamW`` (lr 1e-3, betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2, amsgrad off), and
``state_dict()`` / ``load_state_dict()`` use torch's layout, so checkpoints stay interchangeable with the reference's.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib
from ._lib import check, lib
from .unet import UNet


class AdamW:
    def __init__(self, model: UNet, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if not isinstance(model, UNet):
            raise TypeError("cellseg_amd.AdamW drives a cellseg_amd.UNet (it reads the engine's flat gradient bucket); "
                            "use torch.optim.AdamW for other modules")
        self.model = model
        self.defaults = dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=float(weight_decay))
        self.param_groups = [dict(self.defaults, amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                                  params=list(range(len(model._param_keys))))]
        self._step = 0
        self._m = self._v = self._table = None
        self._ptrs = None
        self._touched = set()            # keys that have taken part in at least one step (the ones state_dict() reports, like torch)

    # -- torch.optim.Optimizer surface the training loop uses --------------------------------------------
    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.model.parameters():
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _bind(self, flat: torch.Tensor, active) -> None:
        """Kernel table of the tensors this step updates: parameters that require a gradient AND have one (torch.optim.AdamW skips
        a parameter whose .grad is None entirely - no weight decay, no moment update)."""
        model = self.model
        params = dict(model.named_parameters())
        ptrs = [(params[k].data_ptr(), k in active) for k in model._param_keys]
        if self._table is None or ptrs != self._ptrs:
            rows = []
            for k in model._param_keys:
                p = params[k]
                if k not in active:
                    continue                     # frozen tensors / tensors without a gradient are left alone
                if p.dtype != torch.float32 or not p.is_contiguous() or p.device != flat.device:
                    raise _lib.CellSegHipError(f"AdamW: parameter {k} must be a contiguous fp32 tensor on {flat.device}")
                off, numel, _ = model._grad_slots[k]
                rows.append((off, numel, p.data_ptr()))
            rows.sort()
            self._table = torch.tensor([[ptr, off, numel] for off, numel, ptr in rows], dtype=torch.int64).reshape(-1, 3).to(flat.device)
            self._ptrs = ptrs
        late = set(active) - self._touched
        if self._step > 0 and self._touched and late:
            # torch keeps a step count per parameter; this optimiser keeps ONE (cs_adamw_step): a tensor whose first gradient arrives
            # after step 1 would get the bias correction of the global count - refused rather than silently different
            raise _lib.CellSegHipError("AdamW: " + ", ".join(sorted(late)[:3]) + f" receive(s) a first gradient after step {self._step}; "
                                       "cellseg_amd.AdamW keeps one step count for all tensors (the set of updated tensors must not grow)")
        self._touched.update(active)
        if self._m is None or self._m.numel() != flat.numel() or self._m.device != flat.device:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)

    def _gradients(self):
        """-> (flat fp32 gradient bucket of the step, keys of the parameters that have a gradient) or (None, ()).  Normally the
        bucket is ``model.flat_grad`` itself (every ``p.grad`` is a view of it); when autograd has accumulated several backward
        passes into other storage the ``p.grad`` tensors are gathered into a flat copy."""
        model = self.model
        flat = model.flat_grad
        params = dict(model.named_parameters())
        grads = [(k, params[k].grad) for k in model._param_keys if params[k].requires_grad and params[k].grad is not None]
        if not grads:
            return None, ()
        active = frozenset(k for k, _ in grads)
        base = flat.data_ptr() if flat is not None else None
        if base is not None and all(g.data_ptr() == base + 4 * model._grad_slots[k][0] for k, g in grads):
            return flat, active
        model.engine(True)
        total = int(lib.cs_unet_grad_numel(model._engine.handle))
        dev = next(model.parameters()).device
        gathered = torch.zeros(total, dtype=torch.float32, device=dev)
        for k, g in grads:
            off, numel, _ = model._grad_slots[k]
            gathered[off:off + numel] = g.reshape(-1).float()
        return gathered, active

    @torch.no_grad()
    def step(self) -> None:
        flat, active = self._gradients()
        if flat is None:
            return                               # nothing to do without gradients (torch.optim.AdamW behaves the same)
        self._bind(flat, active)
        if self._table.shape[0] == 0:
            return
        g = self.param_groups[0]
        self._step += 1
        with torch.cuda.device(flat.device):
            check(lib.cs_adamw_step(C.c_void_p(self._table.data_ptr()), self._table.shape[0], flat.numel(), C.c_void_p(flat.data_ptr()),
                                    C.c_void_p(self._m.data_ptr()), C.c_void_p(self._v.data_ptr()), g["lr"], g["betas"][0], g["betas"][1],
                                    g["eps"], g["weight_decay"], self._step, _lib.stream_ptr()), "cs_adamw_step")
        self.model.invalidate_weights()          # the parameters changed behind torch's version counters

    # -- checkpoints in torch.optim.AdamW's layout ----------------------------------------------------------
    def state_dict(self) -> Dict:
        state = {}
        if self._m is not None:
            for i, k in enumerate(self.model._param_keys):
                if k not in self._touched:
                    continue                     # torch.optim keeps no state for a parameter it has never updated (frozen / no gradient)
                off, numel, shape = self.model._grad_slots[k]
                state[i] = {"step": torch.tensor(float(self._step)), "exp_avg": self._m[off:off + numel].view(shape).clone(),
                            "exp_avg_sq": self._v[off:off + numel].view(shape).clone()}
        return {"state": state, "param_groups": [dict(g) for g in self.param_groups]}

    def load_state_dict(self, sd: Dict) -> None:
        g = sd["param_groups"][0]
        for key in ("lr", "betas", "eps", "weight_decay"):
            self.param_groups[0][key] = tuple(g[key]) if key == "betas" else float(g[key])
        state = sd.get("state", {})
        self._touched = set()
        if not state:
            self._step, self._m, self._v = 0, None, None
            return
        model = self.model
        model.engine(True)                        # the gradient slots come with the engine
        dev = next(model.parameters()).device
        total = int(lib.cs_unet_grad_numel(model._engine.handle))
        self._m = torch.zeros(total, dtype=torch.float32, device=dev)
        self._v = torch.zeros(total, dtype=torch.float32, device=dev)
        for i, k in enumerate(model._param_keys):
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue                         # a torch checkpoint holds no entry for a parameter that never had a gradient: moments stay zero
            off, numel, _ = model._grad_slots[k]
            self._m[off:off + numel] = st["exp_avg"].reshape(-1).to(dev, torch.float32)
            self._v[off:off + numel] = st["exp_avg_sq"].reshape(-1).to(dev, torch.float32)
            self._step = max(self._step if self._touched else 0, int(float(st["step"])))
            self._touched.add(k)
        self._table = None
