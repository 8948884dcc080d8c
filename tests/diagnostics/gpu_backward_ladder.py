"""Ad-hoc GPU diagnostics (not a test): per-parameter gradient error of the HIP backward vs torch-CPU autograd."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch
import cellseg_amd as ca
from oracle.unet_ref import UNetRef

dev = torch.device("cuda:0")
torch.set_num_threads(16)
for channels, nru, shape in [((16, 32), 1, (2, 3, 32, 32)), ((16, 32), 2, (2, 3, 32, 48)), ((16, 32, 64), 2, (2, 3, 64, 64)),
                             ((16, 32, 64, 128, 256), 2, (2, 3, 64, 64)), ((32, 64, 128, 256, 512), 2, (2, 3, 128, 128))]:
    try:
        torch.manual_seed(0)
        ref = UNetRef(channels=channels, strides=(2,) * (len(channels) - 1), num_res_units=nru)
        hip = ca.UNet(channels=channels, strides=(2,) * (len(channels) - 1), num_res_units=nru)
        hip.load_state_dict(ref.state_dict()); hip = hip.to(dev)
        x = torch.rand(*shape); wgt = torch.randn(*shape)
        ref.train(); hip.train()
        (ref(x) * wgt).sum().backward()
        out = hip(x.to(dev)); (out * wgt.to(dev)).sum().backward(); torch.cuda.synchronize()
        gref = dict((k, p.grad) for k, p in ref.named_parameters())
        scale = max(g.abs().max().item() for g in gref.values())
        worst = []
        for k, p in hip.named_parameters():
            g, r = p.grad.cpu(), gref[k]
            err = (g - r).abs().max().item()
            rel = err / max(r.abs().max().item(), 1e-3 * scale)
            worst.append((rel, err, r.abs().max().item(), k))
        worst.sort(reverse=True)
        print(channels, nru, shape, "fwd err %.2e" % (out.detach().cpu() - ref(x)).abs().max().item(), "grad scale %.3e" % scale)
        for rel, err, mag, k in worst[:6]:
            print("   rel %.2e abs %.2e |ref| %.2e  %s" % (rel, err, mag, k))
    except Exception:
        traceback.print_exc()
