"""One rank's shard of BASELINE.json's 8-GPU configurations at FULL per-GPU size on a single MI355X (no process group):
cfg4 = 1080p, D=64, T=80, fp32 stack;  cfg5 = 4K, D=96, T=120, fp16 stack (and fp16 gradient) resident in the 288 GB.
Renders band r of N, runs the backward, reports memory and kernel times, and checks the size-independent property of
tests/test_gpu_render.py::test_cfg3_full_size_frame_independence_and_linearity on it (a stack of identical frames renders
identical frames and receives identical gradient frames) -- i.e. no index of the > 100 GB shard wraps.
    python profiles/cfg45_shard.py cfg4|cfg5 [N=8] [r=3]"""
import sys
import torch
sys.path.insert(0, ".")
from videoloop3d_amd import render as R, synth
from videoloop3d_amd.dist import plan_bands, render_band
from videoloop3d_amd.render import RenderSpec
from videoloop3d_amd.utils_mpi import compute_homography, make_depths

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = int(sys.argv[3]) if len(sys.argv) > 3 else 3
D, T, H, W, dtype = {"cfg4": (64, 80, 1080, 1920, torch.float32), "cfg5": (96, 120, 2160, 3840, torch.float16)}[cfg]
dev = torch.device("cuda:0")
spec = RenderSpec.mpv()
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                           make_depths(D, 1.0, 100.0).flip(0)[None])[0]
band = plan_bands(homos, H, W, H, N, spec)[r]
rows = band.src1 - band.src0
one = synth.make_plane_stack(D, 1, rows, W, seed=2, device=dev, dtype=dtype)
stack = one.expand(D, T, rows, W, 4).contiguous().requires_grad_(True)
del one
g1 = synth.hash_uniform((1, band.rows, W, 3), seed=5, device=dev) - 0.5
g = g1.expand(T, band.rows, W, 3).contiguous()
gib = lambda b: b / 2**30
print(f"{cfg} band {r}/{N}: {band.rows} frame rows, {rows} stack rows; stack {gib(stack.numel() * stack.element_size()):.1f} GiB "
      f"({str(dtype).split('.')[-1]}), gradient the same")
for it in range(3):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    rgb, alpha = render_band(stack, homos.to(dev), band, W, H, spec)
    e[1].record()
    (gs,) = torch.autograd.grad(rgb, stack, g)
    e[2].record()
    torch.cuda.synchronize()
    tile = int(R.LAST_BWD_SCRATCH[:1].view(torch.int32).item())
    f, b = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
    print(f"  fwd {f:.2f} ms  bwd {b:.2f} ms  -> {T * band.rows * W / (f + b) / 1e3:.0f} Mpix/s on this shard, tile_path={tile}, "
          f"peak HBM {gib(torch.cuda.max_memory_allocated()):.1f} GiB")
    if it < 2:
        del gs, rgb, alpha
ok_rgb = all(torch.equal(rgb[t], rgb[0]) for t in range(1, T))
ok_g = all(torch.equal(gs[:, t], gs[:, 0]) for t in range(1, T))
print(f"  frames identical: rgb {ok_rgb}, gradient {ok_g}; finite {bool(torch.isfinite(gs[:, 0]).all())}; "
      f"|grad| max {float(gs[:, 0].float().abs().max()):.4f}")
assert ok_rgb and ok_g and tile == 1
