"""Training data that lives in HBM and is augmented there (csrc/kernels_augment.hip).

The reference feeds its training loop from DataLoader workers that decode every image again each epoch and run the MONAI
transform chain on the CPU (baseline/model_training_3class.py:119-188).  The host mirror of that chain (`cellseg_amd.data`) manages
~130 samples/s per worker; the training step of the reference's own network consumes 5 800 images/s.  An MI355X has 288 GB:
a training set of a few thousand images fits many times over, so

* `DeviceCellDataset` decodes every (image, label) pair ONCE, applies the whole-image min-max scaling (ScaleIntensity, :126) and keeps
  all of them back to back in two device arenas (float32 CHW images, uint8 labels);
* `GpuAugmenter.batch(indices)` draws one parameter record per sample on the host - `data.draw_params`, the same draws in the same
  order as the host pipeline - uploads the records (a few hundred bytes each) and lets the library do everything that touches pixels
  (`cs_augment_batch`): pad / crop / flip / rotate / noise / gamma / Gaussian smoothing / histogram shift / zoom.

The deterministic stages agree with `data.apply_params` to float rounding (tests/test_gpu_augment.py); the noise is Gaussian with the
drawn standard deviation but comes from the device generator, not numpy's stream."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib, data
from ._lib import check, lib

MAX_TAPS = 17


class AugParams(C.Structure):
    """Mirror of ``cs_aug_params`` (include/cellseg_hip.h)."""
    _fields_ = [("img_off", C.c_int64), ("lab_off", C.c_int64), ("src_h", C.c_int32), ("src_w", C.c_int32), ("pad_top", C.c_int32),
                ("pad_left", C.c_int32), ("crop_y", C.c_int32), ("crop_x", C.c_int32), ("flip_axis", C.c_int32), ("rot_k", C.c_int32),
                ("noise_std", C.c_float), ("noise_seed", C.c_uint32), ("gamma", C.c_float), ("taps_y_n", C.c_int32), ("taps_x_n", C.c_int32),
                ("taps_y", C.c_float * MAX_TAPS), ("taps_x", C.c_float * MAX_TAPS), ("hist_n", C.c_int32), ("hist_y", C.c_double * 4),
                ("zoom", C.c_float), ("zoom_h", C.c_int32), ("zoom_w", C.c_int32)]


def fill_record(rec: AugParams, p: dict, img_off: int, lab_off: int, size: int) -> None:
    """One `data.draw_params` record -> the C struct the kernels read."""
    rec.img_off, rec.lab_off = img_off, lab_off
    rec.src_h, rec.src_w = p["src_h"], p["src_w"]
    rec.pad_top, rec.pad_left = p["pad_top"], p["pad_left"]
    rec.crop_y, rec.crop_x = p["crop_y"], p["crop_x"]
    rec.flip_axis, rec.rot_k = p["flip_axis"], p["rot_k"]
    rec.noise_std, rec.noise_seed = p["noise_std"], p["noise_seed"]
    rec.gamma = p["gamma"]
    rec.taps_y_n = rec.taps_x_n = 0
    for name, sigma in (("y", p["sigma_y"]), ("x", p["sigma_x"])):
        if sigma > 0:
            taps = data.gaussian_kernel_1d(sigma)
            if len(taps) > MAX_TAPS:
                raise ValueError(f"Gaussian smoothing with sigma {sigma}: {len(taps)} taps exceed the kernel's {MAX_TAPS}")
            setattr(rec, f"taps_{name}_n", len(taps))
            arr = getattr(rec, f"taps_{name}")
            for j, t in enumerate(taps):
                arr[j] = float(t)
    rec.hist_n = 0
    if p["hist"] is not None:
        if len(p["hist"]) > 4:
            raise ValueError("histogram shift: at most 4 control points")
        rec.hist_n = len(p["hist"])
        for j, v in enumerate(p["hist"]):
            rec.hist_y[j] = float(v)
    rec.zoom = p["zoom"]
    rec.zoom_h = rec.zoom_w = 0
    if p["zoom"] > 0:
        rec.zoom_h = rec.zoom_w = int(np.floor(size * p["zoom"]))


class DeviceCellDataset:
    """All (image, label) pairs of a training set, decoded once and resident in HBM.

    `samples`: a sequence of (float32 [C,H,W] image already intensity-scaled, uint8 [1,H,W] or [H,W] label) numpy pairs, or use
    `from_files` for the reference's directory layout."""

    def __init__(self, samples: Sequence[Tuple[np.ndarray, np.ndarray]], device):
        _lib.require_gpu()
        self.device = torch.device(device)
        self.shapes: List[Tuple[int, int]] = []
        self.img_off: List[int] = []
        self.lab_off: List[int] = []
        n_img = n_lab = 0
        chans = None
        for img, lab in samples:
            c, h, w = img.shape
            chans = c if chans is None else chans
            if c != chans:
                raise ValueError("all images of a DeviceCellDataset need the same channel count")
            self.shapes.append((h, w))
            self.img_off.append(n_img)
            self.lab_off.append(n_lab)
            n_img += c * h * w
            n_lab += h * w
        if not self.shapes:
            raise ValueError("DeviceCellDataset: no samples")
        self.channels = int(chans)
        img_host = np.empty(n_img, np.float32)
        lab_host = np.empty(n_lab, np.uint8)
        for (img, lab), io, lo, (h, w) in zip(samples, self.img_off, self.lab_off, self.shapes):
            img_host[io:io + self.channels * h * w] = np.ascontiguousarray(img, np.float32).ravel()
            lab_host[lo:lo + h * w] = np.ascontiguousarray(lab, np.uint8).reshape(-1)
        self.images = torch.from_numpy(img_host).to(self.device)
        self.labels = torch.from_numpy(lab_host).to(self.device)

    @classmethod
    def from_files(cls, files: Sequence[Tuple[str, str]], device):
        """`files` as `data.split_files` returns them: decoded and scaled exactly like `data.CellDataset` does per access."""
        return cls([data._load(i, g) for i, g in files], device)

    def __len__(self) -> int:
        return len(self.shapes)

    def nbytes(self) -> int:
        return self.images.numel() * 4 + self.labels.numel()


class GpuAugmenter:
    """Batches of augmented crops straight from a `DeviceCellDataset`: `batch(indices)` -> (img [n,C,s,s] float32, label [n,1,s,s] uint8),
    both on the device (the training CLI keeps its host loader for validation: whole images, no augmentation)."""

    def __init__(self, dataset: DeviceCellDataset, input_size: int, seed: int = 0):
        self.ds, self.size = dataset, int(input_size)
        self.rng = np.random.default_rng(seed)
        self._cap = 0
        self._host = self._dev = self._ws = self._copied = None
        self._turn = 0

    def _reserve(self, n: int) -> None:
        if n <= self._cap:
            return
        dev = self.ds.device
        # two pinned staging buffers take turns: a batch's records are written while the previous batch's copy may still be in flight
        self._host = [torch.empty(n * C.sizeof(AugParams), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._dev = [torch.empty(n * C.sizeof(AugParams), dtype=torch.uint8, device=dev) for _ in range(2)]
        self._copied = [None, None]
        with torch.cuda.device(dev):
            self._ws = torch.empty(lib.cs_augment_workspace_bytes(n, self.ds.channels, self.size), dtype=torch.uint8, device=dev)
        self._cap = n

    def draw(self, indices: Sequence[int]) -> List[dict]:
        return [data.draw_params(self.rng, *self.ds.shapes[int(i)], self.size) for i in indices]

    def batch(self, indices: Sequence[int], params: List[dict] = None):
        """`params` (optional): records drawn elsewhere (tests compare the device result with `data.apply_params` on the same records)."""
        n = len(indices)
        if n == 0:
            raise ValueError("GpuAugmenter.batch: empty batch")
        params = self.draw(indices) if params is None else params
        self._reserve(n)
        t = self._turn
        self._turn ^= 1
        if self._copied[t] is not None:
            self._copied[t].synchronize()                      # the copy (and the kernels) that last used this pair of buffers
        host, devbuf = self._host[t], self._dev[t]
        recs = (AugParams * n).from_buffer(host.numpy())
        for k, (i, p) in enumerate(zip(indices, params)):
            fill_record(recs[k], p, self.ds.img_off[int(i)], self.ds.lab_off[int(i)], self.size)
        dev = self.ds.device
        nbytes = n * C.sizeof(AugParams)
        with torch.cuda.device(dev):
            devbuf[:nbytes].copy_(host[:nbytes], non_blocking=True)
            img = torch.empty((n, self.ds.channels, self.size, self.size), dtype=torch.float32, device=dev)
            lab = torch.empty((n, 1, self.size, self.size), dtype=torch.uint8, device=dev)
            check(lib.cs_augment_batch(C.c_void_p(self.ds.images.data_ptr()), C.c_void_p(self.ds.labels.data_ptr()),
                                       C.c_void_p(devbuf.data_ptr()), n, self.ds.channels, self.size, C.c_void_p(img.data_ptr()),
                                       C.c_void_p(lab.data_ptr()), C.c_void_p(self._ws.data_ptr()), self._ws.numel(), _lib.stream_ptr()),
                  "cs_augment_batch")
            ev = torch.cuda.Event()
            ev.record()
            self._copied[t] = ev
        return img, lab
