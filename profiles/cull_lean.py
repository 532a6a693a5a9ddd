#!/usr/bin/env python3
"""bench.py's `tile_culling` leg in process, interleaved rounds: cfg3 render fwd + bwd on the blob model (16.5 % of the quads kept) --
plain (no quad map) / culled (dense gradient: zeros written for culled texels) / lean (VL3D_GRAD_CULLED_UNWRITTEN: the default, 32-wide regions
at the plain kernel's register budget) / lean_reg32 (desc->variant 5: the instantiation with the regularisers' 128-register budget).
  python profiles/cull_lean.py [--T 50] [--rounds 5]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=50)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
import __graft_entry__ as ge  # noqa: E402
ge.build()
from videoloop3d_amd import synth, tiles  # noqa: E402
from videoloop3d_amd.render import RenderSpec, render_planes  # noqa: E402
from videoloop3d_amd.utils_mpi import compute_homography, make_depths  # noqa: E402

dev = torch.device("cuda:0")
D, T, H, W = 32, a.T, 720, 1280
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                           make_depths(D, 1.0, 100.0).flip(0)[None])[0].to(dev)
stack = synth.make_plane_stack(D, T, H, W, seed=2, device=dev).requires_grad_(True)
g = synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5
QH, QW = 35, 63
qy, qx = torch.meshgrid(torch.arange(QH, device=dev), torch.arange(QW, device=dev), indexing="ij")
keep = torch.zeros((D, QH, QW), dtype=torch.bool, device=dev)
for d in range(D):
    cy, cx = (7 * d + 3) % QH, (11 * d + 5) % QW
    keep[d] = ((qy - cy).abs() <= QH // 5) & ((qx - cx).abs() <= QW // 4)
with torch.no_grad():
    tiles.cull_stack_(stack, keep)
legs = {"plain": (None, False, 0), "culled": (keep, False, 0), "lean": (keep, True, 0), "lean_reg32": (keep, True, 5)}
ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
res = {k: ([], []) for k in legs}
for r in range(a.rounds + 1):
    for name, (qk, lean, v) in legs.items():
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        rgb, _ = render_planes(stack, homos, H, W, RenderSpec.mpv(variant=v), quad_keep=qk, grad_culled_unwritten=lean)
        e1.record()
        (gs,) = torch.autograd.grad(rgb, stack, g)
        e2.record()
        torch.cuda.synchronize()
        if r:
            res[name][0].append(e0.elapsed_time(e1)); res[name][1].append(e1.elapsed_time(e2))
        del rgb, gs
print(f"kept quads {float(keep.float().mean()):.3f}, T = {T}")
for name, (f, b) in res.items():
    print(f"{name:10s} fwd median {statistics.median(f):7.3f}  bwd median {statistics.median(b):7.3f} min {min(b):7.3f}  step {statistics.median(f) + statistics.median(b):7.3f} ms")
