"""Training-set preparation CLI with the reference's flags and file layout (/root/reference/data/pre_process_3class.py:58-103).

    python -m cellseg_amd.pre_process_3class -i ./data/Train_Labeled -o ./data/Train_Pre_3class

``<input>/images/<stem>.*`` + ``<input>/labels/<stem>_label.tiff``  ->  ``<output>/images/<stem>.png`` (uint8, per-channel
1-99 percentile stretch) and ``<output>/labels/<stem>_label.png`` (0 background / 1 interior / 2 boundary).
The percentile stretch of integer images and the instance -> 3-class conversion run on the GPU
(cs_normalize_image, cs_interior_map); float images take the host normalisation.
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from .postprocess import create_interior_map
from .preprocess import normalize_image_gpu, prepare_image, read_image


def preprocess_pair(img_data: np.ndarray, gt_data: np.ndarray, device):
    if img_data.dtype in (np.uint8, np.uint16):
        t = torch.from_numpy(img_data.view(np.int16) if img_data.dtype == np.uint16 else img_data).to(device)
        _, pre = normalize_image_gpu(t, return_uint8=True)
        pre = pre.cpu().numpy()
    else:
        pre = prepare_image(img_data)
    inst = torch.from_numpy(np.ascontiguousarray(gt_data.astype(np.int16).astype(np.int32))).to(device)   # :97 astype(np.int16)
    return pre, create_interior_map(inst).cpu().numpy()


def main(argv=None):
    from PIL import Image
    ap = argparse.ArgumentParser("Preprocessing for microscopy image segmentation")
    ap.add_argument("-i", "--input_path", default="./data/Train_Labeled", type=str)
    ap.add_argument("-o", "--output_path", default="./data/Train_Pre_3class", type=str)
    args = ap.parse_args(argv)
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    img_path, gt_path = os.path.join(args.input_path, "images"), os.path.join(args.input_path, "labels")
    img_names = sorted(os.listdir(img_path))
    os.makedirs(os.path.join(args.output_path, "images"), exist_ok=True)
    os.makedirs(os.path.join(args.output_path, "labels"), exist_ok=True)
    for img_name in img_names:
        stem = img_name.split(".")[0]
        pre, interior = preprocess_pair(read_image(os.path.join(img_path, img_name)),
                                        read_image(os.path.join(gt_path, stem + "_label.tiff")), device)
        Image.fromarray(pre.astype(np.uint8)).save(os.path.join(args.output_path, "images", stem + ".png"))
        Image.fromarray(interior.astype(np.uint8)).save(os.path.join(args.output_path, "labels", stem + "_label.png"))
    return len(img_names)


if __name__ == "__main__":
    main()
