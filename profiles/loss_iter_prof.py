#!/usr/bin/env python3
"""One looping-loss iteration (forward + gradient to x) at 720p, ref-view configuration, 12 times: target for rocprofv3 --stats."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from videoloop3d_amd.utils_vid import Patch3DGPNNLowMemLoss
dev = torch.device("cuda:0")
x = torch.empty((1, 3, 52, 720, 1280), device=dev).uniform_().requires_grad_(True)
y = torch.empty((1, 3, 75, 720, 1280), device=dev).uniform_()
cfg = dict(macro_block=65, patch_size=11, stride=4, patcht_size=3, stridet=1, rou='-2', scaling=0.1, alpha=0)
lm = Patch3DGPNNLowMemLoss()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for it in range(12):
        loss = lm(x, y, **cfg)
        (gx,) = torch.autograd.grad(loss, x)
torch.cuda.synchronize()
