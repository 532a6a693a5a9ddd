#!/usr/bin/env python3
"""The bench's `loss` leg, one configuration, y prepared once (bench.py loss_bench): for rocprofv3 --kernel-trace + profiles/gap_analysis.py.
  python profiles/loss_ref_trace.py [ref|other] [iters]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoloop3d_amd import synth
from videoloop3d_amd.utils_vid import Patch3DGPNNLowMemLoss, PreparedClip
which = sys.argv[1] if len(sys.argv) > 1 else "ref"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
H, W, T, Ty = 720, 1280, 50, 75
x = synth.make_video(T + 2, H, W, seed=3, device=dev).requires_grad_(True)
y = synth.make_video(Ty, H, W, seed=4, device=dev)
yp = PreparedClip(y).crop(0, 0)
cfg = {"ref": dict(macro_block=65, patch_size=11, stride=4, patcht_size=3, stridet=1, rou='-2', scaling=0.1, alpha=0),
       "other": dict(macro_block=65, patch_size=3, stride=2, patcht_size=3, stridet=1, rou='-2', scaling=0.1, alpha=10000)}[which]
lm = Patch3DGPNNLowMemLoss()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for it in range(iters + 3):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = lm(x, y, y_prepared=yp, **cfg)
        (gx,) = torch.autograd.grad(loss, x)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"{which}: {dt * 1e3:.3f} ms per iteration = {1 / dt:.1f} it/s")
