"""``model_training_3class`` - command-line drop-in for /root/reference/baseline/model_training_3class.py.

Same flags (:54-80), same loop (:247-330): AdamW(lr 6e-4) on DiceCE(softmax) over 3-class one-hot labels, per-epoch
checkpoint dict {epoch, model_state_dict, optimizer_state_dict, loss}, sliding-window validation with Dice (classes
1..2) every ``val_interval`` epochs after epoch 20, best_Dice_model.pth / final_model.pth / train_log.npz.
Forward and backward of the U-Net run on the MI355X engine.  Under ``torchrun`` (one process per GPU, RCCL) each rank
takes a shard of every epoch and the flat gradient bucket is all-reduced once per step.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch
from torch.utils.data import DataLoader

join = os.path.join


def dice_excluding_background(pred_onehot: torch.Tensor, gt_onehot: torch.Tensor) -> float:
    """monai DiceMetric(include_background=False, reduction='mean') on one image; NaN classes are skipped."""
    vals = []
    for c in range(1, pred_onehot.shape[0]):
        g, p = gt_onehot[c] > 0, pred_onehot[c] > 0
        if g.sum() == 0:
            continue
        vals.append((2.0 * (g & p).sum() / (g.sum() + p.sum())).item())
    return float(np.mean(vals)) if vals else float("nan")


def main(argv=None):
    parser = argparse.ArgumentParser("Baseline for Microscopy image segmentation (MI355X engine)", add_help=True)
    parser.add_argument("--data_path", default="./data/Train_Pre_3class/", type=str, help="training data path; subfolders: images, labels")
    parser.add_argument("--work_dir", default="./work_dir", help="path where to save models and logs")
    parser.add_argument("--seed", default=2022, type=int)
    parser.add_argument("--num_workers", default=4, type=int)
    parser.add_argument("--model_name", default="unet", help="only 'unet' runs on the MI355X engine")
    parser.add_argument("--num_class", default=3, type=int, help="segmentation classes")
    parser.add_argument("--input_size", default=256, type=int, help="segmentation classes")
    parser.add_argument("--batch_size", default=8, type=int, help="Batch size per GPU")
    parser.add_argument("--max_epochs", default=2000, type=int)
    parser.add_argument("--val_interval", default=2, type=int)
    parser.add_argument("--epoch_tolerance", default=100, type=int)
    parser.add_argument("--initial_lr", type=float, default=6e-4, help="learning rate")
    parser.add_argument("--synthetic", type=int, default=0, help="train on N seeded synthetic images instead of --data_path")
    parser.add_argument("--val_start_epoch", type=int, default=20, help="validate only after this epoch (reference: 20)")
    parser.add_argument("--gpu_data", action="store_true", help="keep the decoded training set in HBM and run the augmentation chain there "
                        "(cellseg_amd.gpu_data, csrc/kernels_augment.hip) instead of in DataLoader workers")
    parser.add_argument("--eager", action="store_true", help="run the training step through autograd as the reference writes it instead of "
                                                              "cellseg_amd.TrainStep (the same launches on static buffers)")
    parser.add_argument("--precision", default="fp32", choices=["fp32", "fp16x3", "bf16"],
                        help="matrix-core arithmetic of the training step (engine option; fp32 = the reference's arithmetic, fp16x3 = fp32-grade "
                             "split fp16, bf16 = bf16 convolutions / gradients / activations on the tape with fp32 master weights)")
    parser.add_argument("--graph", action="store_true", help="replay the training step as a captured hipGraph (cellseg_amd.TrainStep(capture=True))")
    args = parser.parse_args(argv)
    if args.model_name.lower() != "unet" or args.num_class != 3:
        raise SystemExit("only --model_name unet --num_class 3 is implemented by the MI355X engine")

    import cellseg_amd as ca
    from cellseg_amd import data, parallel

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("training needs an MI355X: there is no CPU path")
    local = int(os.environ.get("CELLSEG_DEVICE", local))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # CELLSEG_DEVICE / CELLSEG_BACKEND exist only to exercise the multi-rank path on a 1-GPU box (ranks share one GPU, gloo)
        backend = os.environ.get("CELLSEG_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    model_path = join(args.work_dir, args.model_name + "_3class")
    os.makedirs(model_path, exist_ok=True)
    if args.synthetic:
        n_val = max(1, args.synthetic // 10)
        train_ds = data.SyntheticCells(args.synthetic - n_val, args.input_size, args.seed)
        val_ds = data.SyntheticCells(n_val, max(args.input_size, 384), args.seed + 10_000)
    else:
        train_files, val_files = data.split_files(args.data_path, args.seed)
        train_ds = data.CellDataset(train_files, args.input_size, True, args.seed + rank)
        val_ds = data.CellDataset(val_files, args.input_size, False)
    if rank == 0:
        print(f"training image num: {len(train_ds)}, validation image num: {len(val_ds)}")
    sampler = torch.utils.data.distributed.DistributedSampler(train_ds, world, rank, shuffle=True, seed=args.seed) if world > 1 else None
    if args.gpu_data:
        # the training set decoded once into HBM - on EVERY rank the whole set (the DistributedSampler deals a different share of the
        # indices to a rank each epoch, so a rank needs every image over time; 288 GB hold thousands of images many times over);
        # every batch is cropped / augmented there from records the host draws
        from cellseg_amd import gpu_data
        if args.synthetic:
            samples = [(train_ds[i]["img"].numpy(), train_ds[i]["label"].numpy()) for i in range(len(train_ds))]
            dev_ds = gpu_data.DeviceCellDataset(samples, device)
        else:
            dev_ds = gpu_data.DeviceCellDataset.from_files(train_files, device)
        augmenter = gpu_data.GpuAugmenter(dev_ds, args.input_size, seed=args.seed + rank)
        if rank == 0:
            print(f"training set resident in HBM: {len(dev_ds)} images, {dev_ds.nbytes() / 2 ** 20:.1f} MiB")

        class _DeviceLoader:
            """Epoch iterator with the DataLoader's contract for this loop: shuffled (sampler-sharded) index batches -> device tensors."""

            epoch = 0

            def __iter__(self):
                self.epoch += 1
                if sampler is not None:
                    order = list(iter(sampler))
                else:                                  # reproducible from --seed alone (not from the global numpy state)
                    order = list(np.random.default_rng(args.seed + self.epoch).permutation(len(dev_ds)))
                bs = args.batch_size
                stop_at = len(order) - (len(order) % bs if world > 1 else 0)          # drop_last under data parallel, as below
                for b in range(0, stop_at, bs):
                    img, lab = augmenter.batch(order[b:b + bs])
                    yield {"img": img, "label": lab}

        train_loader = _DeviceLoader()
    else:
        train_loader = DataLoader(train_ds, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler,
                                  num_workers=args.num_workers, pin_memory=True, drop_last=world > 1,
                                  worker_init_fn=data.seed_worker, persistent_workers=args.num_workers > 0)
    val_loader = DataLoader(val_ds, batch_size=1, shuffle=False, num_workers=1)

    model = ca.UNet(spatial_dims=2, in_channels=3, out_channels=args.num_class, channels=(16, 32, 64, 128, 256),
                    strides=(2, 2, 2, 2), num_res_units=2).to(device)
    model.precision = args.precision
    loss_function = ca.DiceCELoss(softmax=True)
    optimizer = ca.AdamW(model, args.initial_lr)      # torch.optim.AdamW's update rule and state_dict layout, one HIP launch per step
    train_step = None if args.eager else ca.TrainStep(model, loss_function, optimizer, capture=args.graph)

    best_metric, best_metric_epoch = -1, -1
    epoch_loss_values, metric_values = [], []
    checkpoint = {}
    stop = False
    for epoch in range(1, args.max_epochs):
        model.train()
        if sampler is not None:
            sampler.set_epoch(epoch)
        epoch_loss, step, t0 = torch.zeros((), dtype=torch.float64, device=device), 0, time.time()
        for step, batch in enumerate(train_loader, 1):
            inputs, labels = batch["img"].to(device, non_blocking=True), batch["label"].to(device, non_blocking=True)
            if train_step is not None:
                # the loop body below as one lean launch sequence on static buffers (forward, DiceCE, backward, all-reduce, AdamW, re-pack)
                loss = train_step(inputs, ca.one_hot(labels, args.num_class))
            else:
                optimizer.zero_grad()
                outputs = model(inputs)
                loss = loss_function(outputs, ca.one_hot(labels, args.num_class))
                loss.backward()
                if world > 1:
                    parallel.sync_gradients(model)
                optimizer.step()
            # `epoch_loss += loss.item()` (model_training_3class.py:263) would drain the GPU every step; the sum is kept on the device and
            # read once per epoch - the same average, without 0.25 ms of idle GPU in a 1.4 ms step
            epoch_loss += loss.detach().double()
        epoch_loss = float(epoch_loss.item()) / max(step, 1)
        epoch_loss_values.append(epoch_loss)
        if rank == 0:
            print(f"epoch {epoch} average loss: {epoch_loss:.4f} ({time.time() - t0:.1f}s)")
        checkpoint = {"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                      "loss": epoch_loss_values}
        if epoch > args.val_start_epoch and epoch % args.val_interval == 0 and rank == 0:
            model.eval()
            dices = []
            with torch.no_grad():
                for val in val_loader:
                    img, lab = val["img"].to(device), val["label"].to(device)
                    out = ca.sliding_window_inference(img, (256, 256), 4, model)
                    prob = torch.softmax(out, 1)[0]
                    pred = (prob >= 0.5)                                  # AsDiscrete(threshold=0.5)
                    gt = ca.one_hot(lab, args.num_class)[0]
                    dices.append(dice_excluding_background(pred, gt))
            metric = float(np.nanmean(dices)) if dices else float("nan")
            metric_values.append(metric)
            if metric > best_metric:
                best_metric, best_metric_epoch = metric, epoch + 1
                torch.save(checkpoint, join(model_path, "best_Dice_model.pth"))
                print("saved new best metric model")
            print(f"current epoch: {epoch + 1} current mean dice: {metric:.4f} best mean dice: {best_metric:.4f} at epoch {best_metric_epoch}")
            if (epoch - best_metric_epoch) > args.epoch_tolerance:
                print(f"validation metric does not improve for {args.epoch_tolerance} epochs! current {epoch=}, {best_metric_epoch=}")
                stop = True
        if world > 1:
            # validation runs on rank 0 only: the stop decision is made collective (one broadcast per epoch, which also keeps
            # the ranks in step) so that no rank enters the next epoch's all-reduce after rank 0 has left the loop
            flag = torch.tensor([1 if stop else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
            dist.broadcast(flag, src=0)
            stop = bool(flag.item())
        if stop:
            break
    if rank == 0:
        print(f"train completed, best_metric: {best_metric:.4f} at epoch: {best_metric_epoch}")
        torch.save(checkpoint, join(model_path, "final_model.pth"))
        np.savez_compressed(join(model_path, "train_log.npz"), val_dice=metric_values, epoch_loss=epoch_loss_values)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
