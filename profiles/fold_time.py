#!/usr/bin/env python3
"""kernel time of the fused fold + loss (vl3d_vote_fold_robust) on resident 720p clips, both loss configurations, with the library
VL3D_LIB_PATH names (measurement builds: profiles/build_variant.sh NAME -DVL3D_FOLD_ABLATE=1|2|3)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoloop3d_amd import synth, _lib as L
from videoloop3d_amd.utils_vid import find_nn_indices
dev = torch.device("cuda:0")
x = synth.make_video(52, 719, 1279, seed=3, device=dev)
y = synth.make_video(75, 719, 1279, seed=4, device=dev)
for name, (ps, s, al) in {"ref": (11, 4, 0.5), "other": (3, 2, None)}.items():
    nn, desc, xv, yv = find_nn_indices(x, y, ps, 3, s, 1, al)
    outs = [torch.empty((3, desc.Tx, desc.H, desc.W), device=dev) for _ in range(2)]
    w = torch.empty((desc.Tx, desc.H, desc.W), device=dev)
    acc = torch.empty((), dtype=torch.float64, device=dev)
    ts = []
    for r in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().vl3d_vote_fold_robust(desc, L.ptr(yv), L.ptr(nn), L.ptr(xv), L.RHO["barron"], -2.0, 0.1, L.ptr(outs[0]), L.ptr(w),
                                              L.ptr(outs[1]), L.ptr(acc), L.stream_ptr(dev)), "fold")
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1))
    print(f"{os.path.basename(L.LIB_PATH):24s} {name:6s} fold median {statistics.median(ts):.3f} min {min(ts):.3f} ms")
    # the TRAINING form: y2x / weight stay in registers (no sum / weight outputs), gradient through strides -- what _FoldRobustMean calls
    gx = torch.empty((3, desc.Tx, desc.H, desc.W), device=dev)
    ts = []
    for r in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().vl3d_vote_fold_robust_strided(desc, L.ptr(yv), L.ptr(nn), L.ptr(xv), L.RHO["barron"], -2.0, 0.1, None, None,
                                                      L.ptr(gx), gx.stride(0), gx.stride(1), gx.stride(2), L.ptr(acc), L.stream_ptr(dev)), "fold")
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1))
    print(f"{os.path.basename(L.LIB_PATH):24s} {name:6s} fold (training form) median {statistics.median(ts):.3f} min {min(ts):.3f} ms")
