"""Frame-pair forward (render_fwd2x_k, default) vs the one-frame-per-thread kernel (forward variant 6) in one process, cfg3."""
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
for dtype in (torch.float32, torch.float16):
    stack = synth.make_plane_stack(D, T, H, W, seed=2, device=dev, dtype=dtype)
    res = {}
    for rnd in range(3):
        for variant in (0, 0x600):
            spec = RenderSpec.mpv(variant=variant)
            tf = 0.0
            for it in range(8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); rgb, _ = render_planes(stack, homos, H, W, spec); e1.record(); torch.cuda.synchronize()
                if it >= 3: tf += e0.elapsed_time(e1) / 5
            res.setdefault(variant, []).append(tf)
    print(f"{str(dtype):14s} fwd frame pairs {['%.3f' % v for v in res[0]]} ms   one frame per thread {['%.3f' % v for v in res[0x600]]} ms")
    del stack
