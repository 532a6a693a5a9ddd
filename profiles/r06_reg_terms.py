#!/usr/bin/env python3
"""PMC / timing target for pricing the layer-regulariser terms: ONE forward + backward of the reference geometry (1.1x stack, D = 32, T = 50, 720p, smoothness
regularisers: render_fwd_reg_k + reg_slot_fwd_k<PATCH> + render_bwd_pair_k<REG>) and, for the comparison, of the same geometry WITHOUT regularisers
(render_fwd2x_k + render_bwd_tile_k / pair).  Run with the shipped library and with measurement builds (-DVL3D_REG_ABLATE=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoloop3d_amd import synth
from videoloop3d_amd.render import RenderSpec, render_planes, render_planes_with_smoothness
from videoloop3d_amd.utils_mpi import compute_homography, make_depths
dev = torch.device("cuda:0")
D, T, H, W = 32, 50, 720, 1280
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3), make_depths(D, 1.0, 100.0).flip(0)[None])[0]
Hs, Ws = int(H * 1.1), int(W * 1.1)
shift = torch.tensor([[1.0, 0, (Ws - W) // 2], [0, 1.0, (Hs - H) // 2], [0, 0, 1.0]])
hom = (shift @ homos).to(dev)
torch.manual_seed(0)
stack = torch.empty((D, T, Hs, Ws, 4), dtype=torch.float32, device=dev).uniform_(-2.0, 2.0)
stack[..., 3] -= 2.0
stack.requires_grad_(True)
g = torch.empty((T, H, W, 3), dtype=torch.float32, device=dev).uniform_(-0.5, 0.5)
spec = RenderSpec.mpv()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ev = lambda: torch.cuda.Event(enable_timing=True)
tf, tb, pf, pb = [], [], [], []
for r in range(reps + 1):
    e = [ev() for _ in range(5)]
    e[0].record()
    rgb, _, sums = render_planes_with_smoothness(stack, hom, H, W, spec)
    e[1].record()
    (gs,) = torch.autograd.grad((rgb * g).sum() + 1e-6 * sums.sum(), stack)
    e[2].record()
    del gs, rgb, sums
    rgb, _ = render_planes(stack, hom, H, W, spec)
    e[3].record()
    (gs,) = torch.autograd.grad((rgb * g).sum(), stack)
    e[4].record()
    del gs, rgb
    torch.cuda.synchronize()
    if r:
        tf.append(e[0].elapsed_time(e[1])); tb.append(e[1].elapsed_time(e[2])); pf.append(e[2].elapsed_time(e[3])); pb.append(e[3].elapsed_time(e[4]))
if reps > 1:
    m = lambda v: sorted(v)[len(v) // 2]
    from videoloop3d_amd import _lib
    print(f"{os.path.basename(_lib.LIB_PATH):22s} REG fwd {m(tf):6.2f} ms  bwd {m(tb):6.2f} ms (incl. autograd glue) | plain fwd {m(pf):6.2f}  bwd {m(pb):6.2f}")
