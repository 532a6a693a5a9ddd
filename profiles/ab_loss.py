#!/usr/bin/env python3
"""In-process A/B of the patch-NN kernel variants (utils_vid.KERNEL_VARIANT bits) on resident 720p clips: NN search time per variant.
  python profiles/ab_loss.py 0,4 [rounds]"""
import os, statistics, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from videoloop3d_amd import synth
from videoloop3d_amd import utils_vid as UV
from videoloop3d_amd.utils_vid import find_nn_indices
variants = [int(v, 0) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,4").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
x = synth.make_video(52, 719, 1279, seed=3, device=dev)
y = synth.make_video(75, 719, 1279, seed=4, device=dev)
for name, (ps, s, al) in {"ref": (11, 4, 0.0), "other": (3, 2, None)}.items():
    res = {v: [] for v in variants}
    for r in range(rounds + 1):
        for v in variants:
            UV.KERNEL_VARIANT = v
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            nn = find_nn_indices(x, y, ps, 3, s, 1, al)[0]
            e1.record()
            torch.cuda.synchronize()
            if r:
                res[v].append(e0.elapsed_time(e1))
    ref = None
    for v in variants:
        print(f"{name:6s} variant {v:#5x}  NN median {statistics.median(res[v]):.3f} min {min(res[v]):.3f} ms")
