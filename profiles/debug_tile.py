import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videoloop3d_amd import synth
from videoloop3d_amd.render import RenderSpec, render_planes
from videoloop3d_amd.utils_mpi import compute_homography, make_depths
import __graft_entry__ as g; g.build()
dev = torch.device("cuda:0")
D, T, Hs, Ws, H, W = 32, 2, 720, 1280, 720, 1280
stack = synth.make_plane_stack(D, T, Hs, Ws, seed=2, device=dev).requires_grad_(True)
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3), make_depths(D, 1.0, 100.0).flip(0)[None])[0].to(dev)
gr = (synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5)
out = {}
for v in (1, 2, 3, 1):
    rgb, _ = render_planes(stack, homos, H, W, RenderSpec.mpv(variant=v))
    (gs,) = torch.autograd.grad(rgb, stack, gr)
    out.setdefault(v, []).append(gs)
a1, a1b = out[1]
print("atomics vs atomics:", float((a1 - a1b).abs().max()))
for v in (2, 3):
    d = (out[v][0] - a1).abs()
    m = float(d.max()); idx = torch.nonzero(d == d.max())[0].tolist()
    print("variant", v, "max diff", m, "at (d,t,y,x,c)", idx, "values", float(out[v][0][tuple(idx)]), float(a1[tuple(idx)]), "count>2e-5:", int((d > 2e-5).sum()))
d = (out[3][0] - a1).abs()
idx = torch.nonzero(d > 2e-5)
print(idx.tolist())
print([float(d[tuple(i)]) for i in idx.tolist()])
