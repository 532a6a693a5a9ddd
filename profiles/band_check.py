"""One rank's share of the N-GPU bench on a single GPU (no process group): renders band r of N, checks that the backward took
the owner-computes tile path and prints kernel times.  python profiles/band_check.py [N] [r]"""
import sys
import torch
sys.path.insert(0, ".")
from videoloop3d_amd import render as R, synth
from videoloop3d_amd.dist import plan_bands, render_band
from videoloop3d_amd.render import RenderSpec
from videoloop3d_amd.utils_mpi import compute_homography, make_depths

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
r = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
D, T, H, W = 32, 50, 720, 1280
spec = RenderSpec.mpv()
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                           make_depths(D, 1.0, 100.0).flip(0)[None])[0]
band = plan_bands(homos, H, W, H, N, spec)[r]
rows = band.src1 - band.src0
stack = synth.make_plane_stack(D, T, rows, W, seed=2, device=dev).requires_grad_(True)
g = synth.hash_uniform((T, band.rows, W, 3), seed=5, device=dev) - 0.5
for it in range(3):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    rgb, alpha = render_band(stack, homos.to(dev), band, W, H, spec)
    e[1].record()
    (gs,) = torch.autograd.grad(rgb, stack, g)
    e[2].record()
    torch.cuda.synchronize()
    tile = int(R.LAST_BWD_SCRATCH[:1].view(torch.int32).item())
    print(f"band {r}/{N}: rows {band.rows} (stack rows {rows})  fwd {e[0].elapsed_time(e[1]):.3f} ms  bwd {e[1].elapsed_time(e[2]):.3f} ms  tile_path={tile}")
