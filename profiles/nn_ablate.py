#!/usr/bin/env python3
"""Where the NN search's time goes: patchnn5_k with parts switched off (timing-only ablations, WRONG results -- needs the measurement build:
  profiles/build_variant.sh abl -DVL3D_VARIANTS && VL3D_LIB_PATH=videoloop3d_amd/lib/ab/abl.so python profiles/nn_ablate.py
Ablation bits of desc->variant (bits 4-7): 1 no epilogue, 2 no MFMA loop, 4 no staging DMA.  Prepared y (as training does)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoloop3d_amd import synth
import videoloop3d_amd.utils_vid as U
dev = torch.device("cuda:0")
x = synth.make_video(52, 719, 1279, seed=3, device=dev)
y = synth.make_video(75, 719, 1279, seed=4, device=dev)
yp = U.PreparedClip(y).crop(0, 0)
orig = U._loss_desc
abl = [0]
def desc(*a):
    d = orig(*a)
    d.variant |= abl[0] << 4
    return d
U._loss_desc = desc
for name, (ps, s, al) in {"ref": (11, 4, 0.0), "other": (3, 2, None)}.items():
    res = {}
    for r in range(6):
        for a in (0, 1, 2, 4, 3, 6, 7):
            abl[0] = a
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            U.find_nn_indices(x, y, ps, 3, s, 1, al, y_prepared=yp)
            e1.record()
            torch.cuda.synchronize()
            if r:
                res.setdefault(a, []).append(e0.elapsed_time(e1))
    for a, v in res.items():
        what = " ".join(n for b, n in ((1, "-epilogue"), (2, "-mfma"), (4, "-dma")) if a & b) or "full"
        print(f"{name:6s} {what:24s} median {statistics.median(v):.3f} ms")
