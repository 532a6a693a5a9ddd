import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import mpi_oracle as MO
from videoloop3d_amd import synth
from videoloop3d_amd.render import RenderSpec, render_planes
from videoloop3d_amd.utils_mpi import compute_homography, make_depths
import __graft_entry__ as g; g.build()
dev = torch.device("cuda:0")
D, T, Hs, Ws, H, W = 32, 2, 720, 1280, 720, 1280
stack = synth.make_plane_stack(D, T, Hs, Ws, seed=2, device=dev).requires_grad_(True)
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3), make_depths(D, 1.0, 100.0).flip(0)[None])[0]
gr = (synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5)
r0, c0, h, w = 340, 990, 30, 34
gwin = torch.zeros_like(gr); gwin[:, r0:r0+h, c0:c0+w] = gr[:, r0:r0+h, c0:c0+w]
res = {}
for v in (1, 3):
    rgb, _ = render_planes(stack, homos.to(dev), H, W, RenderSpec.mpv(variant=v))
    (gs,) = torch.autograd.grad(rgb, stack, gwin)
    res[v] = gs[:, 1].cpu()
s_cpu = stack.detach()[:, 1:2].cpu().requires_grad_(True)
shift = torch.tensor([[1.0, 0, c0], [0, 1.0, r0], [0, 0, 1.0]])
rgb_o, _, _ = MO.render_planes(s_cpu, homos @ shift, h, w, MO.RenderSpec(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post"))
(go,) = torch.autograd.grad(rgb_o, s_cpu, gr[1:2, r0:r0+h, c0:c0+w].cpu())
go = go[:, 0]
for v in (1, 3):
    d = (res[v] - go).abs()
    print("variant", v, "vs oracle max", float(d.max()), "at", torch.nonzero(d == d.max())[0].tolist())
d13 = (res[1] - res[3]).abs()
idx = torch.nonzero(d13 == d13.max())[0].tolist()
print("1 vs 3 max", float(d13.max()), idx, float(res[1][tuple(idx)]), float(res[3][tuple(idx)]), float(go[tuple(idx)]))
