"""Ad-hoc GPU diagnostics (not a test): prints max-abs logit error for a ladder of network configs."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import numpy as np, torch
import cellseg_amd as ca
from oracle.unet_ref import UNetRef

dev = torch.device("cuda:0")
print("device", torch.cuda.get_device_name(0), "devices", ca.lib.cs_device_count())
for channels, nru, shape in [((16, 32), 1, (1, 3, 32, 32)), ((16, 32), 2, (2, 3, 32, 48)), ((16, 32, 64), 2, (2, 3, 64, 64)),
                             ((16, 32, 64, 128, 256), 2, (2, 3, 64, 64)), ((16, 32, 64, 128, 256), 2, (4, 3, 256, 256)),
                             ((32, 64, 128, 256, 512), 2, (2, 3, 128, 128))]:
    try:
        torch.manual_seed(0)
        ref = UNetRef(channels=channels, strides=(2,) * (len(channels) - 1), num_res_units=nru).eval()
        hip = ca.UNet(channels=channels, strides=(2,) * (len(channels) - 1), num_res_units=nru)
        hip.load_state_dict(ref.state_dict()); hip = hip.to(dev).eval()
        x = torch.rand(*shape)
        with torch.no_grad():
            want = ref(x); got = hip(x.to(dev)); torch.cuda.synchronize(); got = got.cpu()
        d = (got - want).abs()
        print(channels, nru, shape, "max err %.3e mean %.3e ref absmax %.3f nan=%d" % (d.max(), d.mean(), want.abs().max(), torch.isnan(got).sum()))
    except Exception:
        traceback.print_exc()
