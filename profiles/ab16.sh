#!/bin/bash
# like ab.sh, for the fp16-stack leg (cfg5's storage format on the cfg3 geometry): fwd / bwd kernel times from HIP events
for r in 1 2 3; do
  for so in "$@"; do
    VL3D_LIB_PATH=$PWD/videoloop3d_amd/$so timeout 200 python - "$so" <<'PY'
import sys, torch
sys.path.insert(0, ".")
from videoloop3d_amd import synth
from videoloop3d_amd.render import RenderSpec, render_planes
from videoloop3d_amd.utils_mpi import compute_homography, make_depths
dev = torch.device("cuda:0")
D, T, H, W = 32, 50, 720, 1280
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                           make_depths(D, 1.0, 100.0).flip(0)[None])[0].to(dev)
st = synth.make_plane_stack(D, T, H, W, seed=2, device=dev, dtype=torch.float16).requires_grad_(True)
g = synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5
f = b = 0.0
for it in range(8):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); rgb, _ = render_planes(st, homos, H, W, RenderSpec.mpv()); e[1].record()
    (gs,) = torch.autograd.grad(rgb, st, g); e[2].record(); torch.cuda.synchronize()
    if it >= 3: f += e[0].elapsed_time(e[1]) / 5; b += e[1].elapsed_time(e[2]) / 5
print(f"{sys.argv[1]:24s} fp16 stack: fwd {f:.3f} ms  bwd {b:.3f} ms  -> {T*H*W/(f+b)/1e3:.0f} Mpix/s  checksum {float(rgb.double().sum()):.6f} {float(gs.double().abs().sum()):.4f}")
PY
  done
done
