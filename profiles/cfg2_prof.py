#!/usr/bin/env python3
"""cfg2 of BASELINE.json (stage-1 shape: ONE 720p frame, D = 32) render fwd + bwd, for rocprofv3 --kernel-trace --stats.  python profiles/cfg2_prof.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoloop3d_amd import synth
from videoloop3d_amd.render import RenderSpec, render_planes
from videoloop3d_amd.utils_mpi import compute_homography, make_depths
dev = torch.device("cuda:0")
D, H, W = 32, 720, 1280
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                           make_depths(D, 1.0, 100.0).flip(0)[None])[0].to(dev)
st = synth.make_plane_stack(D, 1, H, W, seed=2, device=dev).requires_grad_(True)
g = synth.hash_uniform((1, H, W, 3), seed=5, device=dev) - 0.5
spec = RenderSpec.mpv()
for it in range(45):
    if it == 5:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    r, _ = render_planes(st, homos, H, W, spec)
    (gs,) = torch.autograd.grad(r, st, g)
torch.cuda.synchronize()
print("cfg2 ms per step", (time.perf_counter() - t0) / 40 * 1e3)
