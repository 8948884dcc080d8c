import sys, time, torch
sys.path[:0] = ["/root/repo", "/root/repo/neurips-cellseg_amd"]
import cellseg_amd as ca
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ca.UNet(channels=(32, 64, 128, 256, 512), strides=(2, 2, 2, 2), num_res_units=2).to(dev).train()
x = torch.rand(32, 3, 256, 256, device=dev); y = torch.randint(0, 3, (32, 1, 256, 256), device=dev)
onehot = ca.one_hot(y, 3); loss_fn = ca.DiceCELoss(softmax=True); opt = torch.optim.AdamW(model.parameters(), 6e-4)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
tot = [0.0] * 5
for it in range(25):
    opt.zero_grad(set_to_none=True)
    ev[0].record(); out = model(x); ev[1].record(); loss = loss_fn(out, onehot); ev[2].record(); loss.backward(); ev[3].record(); opt.step(); ev[4].record()
    torch.cuda.synchronize()
    if it >= 5:
        for k in range(4): tot[k] += ev[k].elapsed_time(ev[k + 1])
print("ms per step: forward %.3f  loss %.3f  backward(loss+net) %.3f  adamw %.3f" % tuple(t / 20 for t in tot[:4]))
