"""``predict`` - command-line drop-in for /root/reference/baseline/predict.py (same flags, same outputs).

    python -m cellseg_amd.predict -i ./inputs -o ./outputs --model_path ./work_dir/unet_3class --model_name unet

Per image: read -> per-channel percentile normalisation (host) -> H2D -> sliding-window U-Net inference, softmax,
> 0.5, remove_small_holes, remove_small_objects, 8-connected labelling (all on the MI355X) -> D2H -> int32 zlib
TIFF ``<stem>_label.tiff``.  Only ``--model_name unet`` is accelerated (unetr / swinunetr are out of scope and
refused).  Under ``torchrun`` the images are dealt to the ranks (one process per GPU, no collective); with ``--split_slides`` every
image is instead spread over ALL ranks (whole-slide images: its sliding windows are split by the window list, window parts travel to
the later neighbour, the mask rectangles are all-gathered, every rank labels its own rows - strip-local connected components with a
seam merge - and rank 0 collects and writes the label map; the file is bit-identical to the single-GPU one).
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

join = os.path.join


_PINNED = {}


def _to_host(t: torch.Tensor) -> np.ndarray:
    """Device label map -> numpy through a cached pinned staging buffer (a pageable 67 MB copy costs ~12 ms, a pinned one ~2)."""
    key = (tuple(t.shape), t.dtype)
    buf = _PINNED.get(key)
    if buf is None:
        _PINNED.clear()                                      # one staging buffer at a time
        buf = _PINNED[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    buf.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return buf.numpy()


def main(argv=None):
    parser = argparse.ArgumentParser("Baseline for Microscopy image segmentation (MI355X engine)", add_help=True)
    parser.add_argument("-i", "--input_path", default="./inputs", type=str, help="input images")
    parser.add_argument("-o", "--output_path", default="./outputs", type=str, help="output path")
    parser.add_argument("--model_path", default="./work_dir/unet_3class", help="directory holding best_Dice_model.pth")
    parser.add_argument("--show_overlay", required=False, default=False, action="store_true", help="save segmentation overlay")
    parser.add_argument("--model_name", default="unet", help="only 'unet' runs on the MI355X engine")
    parser.add_argument("--num_class", default=3, type=int, help="segmentation classes")
    parser.add_argument("--input_size", default=256, type=int, help="sliding-window size")
    parser.add_argument("--overlap", default=0.25, type=float, help="window overlap (MONAI default 0.25)")
    parser.add_argument("--random_init", action="store_true", help="no checkpoint: random weights (smoke runs)")
    parser.add_argument("--split_slides", action="store_true",
                        help="under torchrun: spread EVERY image over all ranks (RCCL seam exchange, mask all-gather, strip-local connected "
                             "components) instead of dealing whole images to the ranks (engine option; the reference is single-device)")
    parser.add_argument("--precision", default="fp32", choices=["fp32", "fp16x3", "bf16x6", "bf16x3", "bf16", "fp8"],
                        help="matrix-core arithmetic of the convolutions (engine option; fp16x3 / bf16x6 are fp32-grade)")
    args = parser.parse_args(argv)

    if args.model_name.lower() != "unet":
        raise SystemExit(f"--model_name {args.model_name}: only 'unet' is implemented by the MI355X engine")
    if args.num_class != 3:
        raise SystemExit("--num_class: the engine implements the reference's 3-class contract")

    import cellseg_amd as ca
    from cellseg_amd import parallel, preprocess

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("predict needs an MI355X: there is no CPU path")
    local = int(os.environ.get("CELLSEG_DEVICE", local))          # (functional checks on a one-GPU box: all ranks on one device, gloo)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    split = bool(args.split_slides) and world > 1
    transport = "p2p"
    if split:
        import datetime
        import torch.distributed as dist
        backend = os.environ.get("CELLSEG_BACKEND", "nccl")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=int(os.environ.get("CELLSEG_PG_TIMEOUT_S", "600"))), **kw)
        # a library whose send/recv fails: one all-gather instead (CELLSEG_SEAM_TRANSPORT forces either, for tests)
        transport = os.environ.get("CELLSEG_SEAM_TRANSPORT") or ("p2p" if parallel.p2p_selfcheck(device) else "allgather")
        if transport not in ("p2p", "allgather"):
            raise SystemExit(f"CELLSEG_SEAM_TRANSPORT: {transport!r} is neither 'p2p' nor 'allgather'")

    os.makedirs(args.output_path, exist_ok=True)
    img_names = preprocess.list_images(args.input_path)
    if args.random_init:
        torch.manual_seed(0)                                  # smoke runs are repeatable, and every rank of a split slide holds the same net
    model = ca.UNet(spatial_dims=2, in_channels=3, out_channels=args.num_class, channels=(16, 32, 64, 128, 256),
                    strides=(2, 2, 2, 2), num_res_units=2)
    if not args.random_init:
        checkpoint = torch.load(join(args.model_path, "best_Dice_model.pth"), map_location="cpu")
        model.load_state_dict(checkpoint["model_state_dict"])
    model = model.to(device).eval()
    model.precision = args.precision
    roi_size = (args.input_size, args.input_size)

    # file I/O either side of the GPU path overlaps with it: the next image is decoded by a reader thread while the current one is
    # on the GPU, label maps (and overlays) are written by a writer thread; the order of the printed lines stays the input order
    from concurrent.futures import ThreadPoolExecutor
    todo = list(range(len(img_names))) if split else list(parallel.shard_items(len(img_names), rank, world))
    say = print if (not split or rank == 0) else (lambda *a, **k: None)     # split slides: every rank sees every image, rank 0 reports
    reader, writer = ThreadPoolExecutor(1), ThreadPoolExecutor(1)
    pending_writes = []

    def save(img_name, img_data, strips, labels):
        # the label map was deflated on the device (csrc/kernels_deflate.hip): the writer thread only wraps the strips into a TIFF
        data, sizes, (h, w), rps = strips
        preprocess.write_tiff_strips(join(args.output_path, preprocess.label_name(img_name)), h, w, 1, np.dtype("<i4"), rps, data, sizes)
        if args.show_overlay:
            # predict.py:118-122: inner boundaries of the instances, dilated by disk(2), painted 255 into the image as read,
            # saved as overlay_<img_name> in the input's own format
            preprocess.save_image(join(args.output_path, "overlay_" + img_name), preprocess.overlay_boundaries(img_data, labels))

    def read(k):
        """Reader thread: decode, and stage integer images in PINNED memory so that the main thread's upload is an asynchronous 2 ms
        DMA instead of a synchronous 10 ms copy out of pageable memory (50 MB for a 4096^2 RGB image)."""
        try:
            img = preprocess.read_image(join(args.input_path, img_names[k]))
            staged = None
            if img.dtype in (np.uint8, np.uint16) and img.ndim in (2, 3):
                flat = np.ascontiguousarray(img)
                staged = torch.from_numpy(flat.view(np.int16) if flat.dtype == np.uint16 else flat).pin_memory()
            return img, staged
        except Exception as e:                                   # one undecodable file must not end the run (reader thread)
            return e

    failed = []
    nxt = reader.submit(read, todo[0]) if todo else None
    with torch.no_grad():
        for pos, k in enumerate(todo):
            img_name = img_names[k]
            got = nxt.result()
            nxt = reader.submit(read, todo[pos + 1]) if pos + 1 < len(todo) else None
            why = None
            if isinstance(got, Exception):
                why = str(got)
            elif got[0].ndim not in (2, 3) or (got[0].ndim == 3 and got[0].shape[-1] < 3):
                why = f"unsupported image shape {got[0].shape} ({got[0].dtype})"
            # split slides: the ranks agree on the image's status BEFORE the first data-path collective (a read error on one rank only
            # would otherwise leave its peers inside the seam exchange until the process-group timeout)
            if split and not parallel.all_ok(why is None, device) and why is None:
                why = "another rank could not read or prepare this image"
            if why is not None:
                say(f"Prediction FAILED: {img_name}: {why}")
                failed.append(img_name)
                continue
            img_data, staged = got
            t0 = time.time()
            if img_data.dtype in (np.uint8, np.uint16):
                # integer images: percentile normalisation on the device (bit-exact with the host path below)
                x = preprocess.normalize_image_gpu(staged.to(device, non_blocking=True))      # [H,W,3] float32
                shape = tuple(x.shape)
            else:
                pre_img_data = preprocess.prepare_image(img_data)
                x = torch.from_numpy(preprocess.to_unit_float(pre_img_data)).to(device)      # [H,W,3]: NHWC, no copy needed
                shape = pre_img_data.shape
            if split:
                # one slide over all ranks: this rank's share of the window list, the seam exchange, its rectangles of the mask; ONE
                # all-gather of the uint8 rectangles; its rows filtered and labelled here (two small all-gathers around the seam
                # union-find); the int32 strips gathered on rank 0, which deflates and writes like the single-GPU path
                image = x.permute(2, 0, 1)
                plan = ca.make_plan(image.shape[1], image.shape[2], roi_size, args.overlap)
                _, _, _, part = ca.swi_infer_share(model, image, roi_size, args.overlap, rank, world,
                                                   exchange=lambda win, pl, r, w: parallel.exchange_share(win, pl, r, w, transport=transport))
                mask = parallel.gather_shares(part, plan, world)
                strip, _, _ = parallel.label_strips(mask)
                lab = parallel.gather_label_strips(strip, mask.shape[0], dst=0)
                if rank != 0:
                    continue
            else:
                _, _, mask = ca.swi_infer_image(model, x.permute(2, 0, 1), roi_size, args.overlap, want_logits=False, want_mask=True)
                lab = ca.mask_to_instances(mask)
            rps = min(64, lab.shape[0])
            buf, sizes = preprocess.deflate_labels_gpu(lab, rps)
            strips = (bytes(memoryview(buf.numpy())), sizes, tuple(lab.shape), rps)   # the staging buffer is reused by the next image
            labels = _to_host(lab).copy() if args.show_overlay else None             # the overlay is drawn on the host
            pending_writes.append((img_name, writer.submit(save, img_name, img_data, strips, labels)))
            t1 = time.time()
            say(f"Prediction finished: {img_name}; img size = {shape}; costing: {t1 - t0:.2f}s")
    for img_name, f in pending_writes:
        try:
            f.result()                                                    # surface writer errors, finish before exit
        except Exception as e:
            print(f"Writing FAILED: {img_name}: {type(e).__name__}: {e}")
            failed.append(img_name)
    reader.shutdown()
    writer.shutdown()
    if split:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        raise SystemExit(f"{len(failed)} of {len(todo)} images failed: {', '.join(failed)}")


if __name__ == "__main__":
    main()
