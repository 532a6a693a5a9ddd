#!/usr/bin/env python3
"""The NN search alone on resident 720p clips, y prepared (as training does): for rocprofv3 --kernel-trace --stats.
  python profiles/nn_prof.py ref|other [variant] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoloop3d_amd import synth
import videoloop3d_amd.utils_vid as U
cfg = sys.argv[1] if len(sys.argv) > 1 else "ref"
U.KERNEL_VARIANT = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
x = synth.make_video(52, 719, 1279, seed=3, device=dev)
y = synth.make_video(75, 719, 1279, seed=4, device=dev)
yp = U.PreparedClip(y).crop(0, 0)
ps, s, al = {"ref": (11, 4, 0.0), "other": (3, 2, None)}[cfg]
for r in range(rounds):
    U.find_nn_indices(x, y, ps, 3, s, 1, al, y_prepared=yp)
torch.cuda.synchronize()
