import sys, torch
sys.path[:0] = ["/root/repo", "/root/repo/neurips-cellseg_amd"]
import cellseg_amd as ca
dev = torch.device("cuda:0")
y = torch.randint(0, 3, (32, 1, 256, 256), device=dev); onehot = ca.one_hot(y, 3)
loss_fn = ca.DiceCELoss(softmax=True)
x = torch.randn(32, 3, 256, 256, device=dev, requires_grad=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for it in range(30):
    x.grad = None
    ev[0].record(); loss = loss_fn(x, onehot); ev[1].record(); loss.backward(); ev[2].record()
    torch.cuda.synchronize()
    if it >= 10: tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print("DiceCE forward %.3f ms  backward %.3f ms" % (tf / 20, tb / 20))
