#!/usr/bin/env python3
"""Diagnostic: loss curves of the bench's training workload (32...512, batch 32 x 3 x 256^2, random labels, AdamW 6e-4) in the fp32,
fp16x3 and bf16 training modes, every 15th of 121 steps.  Round 3: the three curves agree to 2e-3 over the whole run.
    python tests/diagnostics/gpu_loss_curves.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips-cellseg_amd")]
import torch  # noqa: E402

import cellseg_amd as ca  # noqa: E402

dev=torch.device('cuda:0')
for prec in ('fp32','fp16x3','bf16'):
    torch.manual_seed(2022)
    model=ca.UNet(channels=(32,64,128,256,512),strides=(2,2,2,2),num_res_units=2).to(dev).train()
    model.precision=prec
    g=torch.Generator().manual_seed(2023)
    x=torch.rand(32,3,256,256,generator=g).to(dev); y=ca.one_hot(torch.randint(0,3,(32,1,256,256),generator=g).to(dev),3)
    step=ca.TrainStep(model, ca.DiceCELoss(softmax=True), ca.AdamW(model,6e-4))
    out=[]
    for it in range(121):
        l=step(x,y)
        if it%15==0: out.append(round(l.item(),4))
    print(prec,out)
