"""cellseg_amd - MI355X (gfx950) engine behind the NeurIPS-CellSeg 3-class U-Net inference/training path.

Host-side mirror of the reference's interfaces (same names, arguments and error behaviour):
  UNet                       <- monai.networks.nets.UNet            (baseline/predict.py:45-52)
  sliding_window_inference   <- monai.inferers.sliding_window_inference (baseline/predict.py:109)
  label / mask_to_instances  <- skimage.measure.label / morphology.remove_small_* (baseline/predict.py:113)
  create_interior_map        <- data/pre_process_3class.py:31-56
  DiceCELoss / AdamW         <- monai.losses.DiceCELoss(softmax=True) / torch.optim.AdamW (baseline/model_training_3class.py:234-237)
  TrainStep                  <- the loop body baseline/model_training_3class.py:250-262 as one captured launch sequence (hipGraph)
All arithmetic runs in libcellseg_hip.so; importing this package without the built library raises.
"""
from ._lib import CellSegHipError, LIB_PATH, lib  # noqa: F401
from .unet import UNet  # noqa: F401
from .inferers import (balanced_batch, make_plan, rank_rows, sliding_window_inference, share_partition, strip_partition, swi_infer_image, swi_infer_share, swi_infer_strip,  # noqa: F401
                       window_starts)
from .postprocess import StripLabeller, create_interior_map, label, mask_to_instances, mask_to_instances_strips, prob_to_instances  # noqa: F401
from .losses import DiceCELoss, one_hot  # noqa: F401
from .optim import AdamW  # noqa: F401
from .trainer import TrainStep  # noqa: F401
from . import metrics, parallel  # noqa: F401

__all__ = ["UNet", "sliding_window_inference", "swi_infer_image", "make_plan", "window_starts", "rank_rows", "strip_partition", "swi_infer_strip", "share_partition", "swi_infer_share",
           "label", "mask_to_instances", "mask_to_instances_strips", "StripLabeller", "prob_to_instances", "create_interior_map", "DiceCELoss", "one_hot", "AdamW", "TrainStep", "CellSegHipError", "lib", "LIB_PATH"]
