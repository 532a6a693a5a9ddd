"""fwd / bwd kernel-level times of the tile-culled cfg3 render (bench `tile_culling` leg) -- python profiles/cull_breakdown.py"""
import sys
import torch
sys.path.insert(0, ".")
from videoloop3d_amd import synth, tiles
from videoloop3d_amd.render import RenderSpec, render_planes
from videoloop3d_amd.utils_mpi import compute_homography, make_depths
dev = torch.device("cuda:0")
D, T, H, W = 32, 50, 720, 1280
spec = RenderSpec.mpv()
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                           make_depths(D, 1.0, 100.0).flip(0)[None])[0].to(dev)
stack = synth.make_plane_stack(D, T, H, W, seed=2, device=dev)
QH, QW = 35, 63
qy, qx = torch.meshgrid(torch.arange(QH, device=dev), torch.arange(QW, device=dev), indexing="ij")
for frac_h, frac_w in ((5, 4), (3, 3), (2, 2)):
    keep = torch.zeros((D, QH, QW), dtype=torch.bool, device=dev)
    for d in range(D):
        cy, cx = (7 * d + 3) % QH, (11 * d + 5) % QW
        keep[d] = ((qy - cy).abs() <= QH // frac_h) & ((qx - cx).abs() <= QW // frac_w)
    s = stack.clone()
    tiles.cull_stack_(s, keep)
    s.requires_grad_(True)
    g = synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5
    for name, qk in (("plain ", None), ("culled", keep)):
        fw, bw = [], []
        for it in range(5):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            rgb, alpha = render_planes(s, homos, H, W, spec, quad_keep=qk)
            e[1].record()
            (gs,) = torch.autograd.grad(rgb, s, g)
            e[2].record()
            torch.cuda.synchronize()
            if it >= 2:
                fw.append(e[0].elapsed_time(e[1])); bw.append(e[1].elapsed_time(e[2]))
        print(f"kept {float(keep.float().mean()):.3f}  {name}: fwd {sum(fw)/len(fw):6.3f} ms  bwd {sum(bw)/len(bw):6.3f} ms")
    del s, gs, rgb
