import sys, json
ROOT='/root/repo'
for p in (ROOT, ROOT+'/neurips-cellseg_amd', ROOT+'/tests'): sys.path.insert(0,p)
import torch, cellseg_amd as ca
from oracle.unet_ref import UNetRef
from blank_inputs import blank_windows
dev=torch.device('cuda:0')
w=blank_windows()
x=torch.cat([w[k] for k in ('zero','const','patch22x13','sparse_cells_u8','noise','one_pixel','half_zero','corner9x17')],0)
for slope in (1.0, None):
    torch.manual_seed(0)
    cfg=dict(channels=(16,32,64,128,256), strides=(2,2,2,2), num_res_units=2)
    ref=UNetRef(**cfg).train()
    if slope is not None:
        for n,p in ref.named_parameters():
            if n.endswith('A.weight'): p.data.fill_(slope)
    ref64=UNetRef(**cfg).double().train(); ref64.load_state_dict({k:v.double() for k,v in ref.state_dict().items()})
    hip=ca.UNet(**cfg); hip.load_state_dict(ref.state_dict()); hip=hip.to(dev).train()
    g=torch.Generator().manual_seed(1)
    wgt=torch.randn(*x.shape, generator=g)
    (ref64(x.double())*wgt.double()).sum().backward()
    (ref(x)*wgt).sum().backward()
    (hip(x.to(dev))*wgt.to(dev)).sum().backward()
    g64={k:p.grad for k,p in ref64.named_parameters()}; g32={k:p.grad.double() for k,p in ref.named_parameters()}
    rows=[]
    for k,p in hip.named_parameters():
        gh=p.grad.cpu().double()
        rows.append((k,(gh-g64[k]).abs().max().item(),(g32[k]-g64[k]).abs().max().item(),g64[k].abs().max().item()))
    scale=max(r[3] for r in rows)
    wh=max(r[1]/max(r[3],1e-3*scale) for r in rows); wt=max(r[2]/max(r[3],1e-3*scale) for r in rows)
    print('slope',slope,'scale %.3e'%scale,'worst rel err hip %.3e torch32 %.3e'%(wh,wt))
    for r in sorted(rows,key=lambda r:-r[1]/max(r[3],1e-3*scale))[:5]: print('   ',r[0][-50:], 'hip %.2e t32 %.2e mag %.2e'%(r[1],r[2],r[3]))
