"""Frame-pair backward (render_bwd_pair_k, default) vs the one-frame tile kernel (variant 3) in one process, cfg3 geometry,
stack = frame size and the reference's 1.1x, fp32 and fp16."""
import sys
import torch
sys.path.insert(0, ".")
from videoloop3d_amd import synth
from videoloop3d_amd.render import RenderSpec, render_planes
from videoloop3d_amd.utils_mpi import compute_homography, make_depths
dev = torch.device("cuda:0")
D, T, H, W = 32, 50, 720, 1280
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                           make_depths(D, 1.0, 100.0).flip(0)[None])[0].to(dev)
g = synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5
for dtype, scale in ((torch.float32, 1.0), (torch.float32, 1.05), (torch.float16, 1.0), (torch.float16, 1.1)):
    Hs, Ws = int(round(H * scale)), int(round(W * scale))
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=2, device=dev, dtype=dtype).requires_grad_(True)
    res = {}
    for rnd in range(3):
        for variant in (0, 3):
            spec = RenderSpec.mpv(scale=(scale, scale), variant=variant)
            tb = 0.0
            for it in range(6):
                rgb, _ = render_planes(stack, homos, H, W, spec)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); (gs,) = torch.autograd.grad(rgb, stack, g); e1.record(); torch.cuda.synchronize()
                if it >= 2: tb += e0.elapsed_time(e1) / 4
            res.setdefault(variant, []).append(tb)
            del gs, rgb
    print(f"{str(dtype):14s} stack {scale}x: bwd frame pairs {['%.2f' % v for v in res[0]]} ms   one frame per thread {['%.2f' % v for v in res[3]]} ms")
    del stack
