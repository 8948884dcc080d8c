#!/bin/bash
# End-to-end smoke of the two reference CLIs on the GPU box: train on synthetic cells, then predict with the checkpoint.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export PYTHONPATH=$R/neurips-cellseg_amd:$R
W=$(mktemp -d)
python -m cellseg_amd.model_training_3class --synthetic 40 --max_epochs 7 --val_start_epoch 1 --val_interval 2 --batch_size 8 --num_workers 2 --work_dir $W/work
python - <<PY
import numpy as np, os
from PIL import Image
from scipy.ndimage import gaussian_filter
os.makedirs("$W/in", exist_ok=True)
rng = np.random.default_rng(5)
for i, (h, w) in enumerate([(700, 900), (200, 180)]):
    g = gaussian_filter(rng.random((h, w)), 5); g = (g - g.min()) / (g.max() - g.min())
    Image.fromarray((np.stack([g, g, g], -1) * 255).astype(np.uint8)).save(f"$W/in/cell_{i:03d}.png")
Image.fromarray((gaussian_filter(rng.random((300, 300)), 4) * 60000).astype(np.uint16)).save("$W/in/gray16.tif")
PY
python -m cellseg_amd.predict -i $W/in -o $W/out --model_path $W/work/unet_3class --model_name unet --show_overlay
python -m cellseg_amd.predict -i $W/in -o $W/out16 --model_path $W/work/unet_3class --model_name unet --precision fp16x3
python - <<PY
import numpy as np, os
from PIL import Image
for f in sorted(os.listdir("$W/out16")):
    a, b = np.array(Image.open(os.path.join("$W/out16", f))), np.array(Image.open(os.path.join("$W/out", f)))
    print("fp16x3 vs fp32 labels", f, "identical" if np.array_equal(a, b) else "differ: %d px" % int((a != b).sum()))
PY
python - <<PY
import numpy as np, os
from PIL import Image
for f in sorted(os.listdir("$W/out")):
    a = np.array(Image.open(os.path.join("$W/out", f)))
    print(f, a.shape, a.dtype, int(a.max()))
PY
# score the predictions against themselves shifted by a pixel (compute_metric CLI: reference flags, CSV columns)
python - <<PY
import numpy as np, os
from PIL import Image
os.makedirs("$W/gt", exist_ok=True)
for f in sorted(os.listdir("$W/out")):
    if f.endswith("_label.tiff"):
        a = np.array(Image.open(os.path.join("$W/out", f)))
        Image.fromarray(np.roll(a, 1, 1).astype(np.int32)).save(os.path.join("$W/gt", f.replace("_label.tiff", ".tif")))
PY
python -m cellseg_amd.compute_metric -g $W/gt -s $W/out -o $W -n smoke -thre 0.5 0.7
cat $W/smoke-0.5.csv
# training-set preparation CLI (pre_process_3class): instance labels -> 3-class PNGs
python - <<PY
import numpy as np, os, shutil
from PIL import Image
os.makedirs("$W/raw/images", exist_ok=True); os.makedirs("$W/raw/labels", exist_ok=True)
for f in sorted(os.listdir("$W/in")):
    shutil.copy(os.path.join("$W/in", f), "$W/raw/images/")
    stem = f.split(".")[0]
    shutil.copy(os.path.join("$W/out", stem + "_label.tiff"), os.path.join("$W/raw/labels", stem + "_label.tiff"))
PY
python -m cellseg_amd.pre_process_3class -i $W/raw -o $W/pre
python - <<PY
import numpy as np, os
from PIL import Image
for d in ("images", "labels"):
    for f in sorted(os.listdir(os.path.join("$W/pre", d))):
        a = np.array(Image.open(os.path.join("$W/pre", d, f)))
        print(d, f, a.shape, a.dtype, np.unique(a)[:4])
PY
