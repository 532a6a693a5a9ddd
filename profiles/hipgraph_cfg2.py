import sys, time, torch
sys.path.insert(0, ".")
from videoloop3d_amd import synth
from videoloop3d_amd.render import RenderSpec, render_planes
from videoloop3d_amd.utils_mpi import compute_homography, make_depths
dev = torch.device("cuda:0")
D, T, H, W = 32, 1, 720, 1280
spec = RenderSpec.mpv()
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3), make_depths(D, 1.0, 100.0).flip(0)[None])[0].to(dev)
stack = synth.make_plane_stack(D, T, H, W, seed=2, device=dev).requires_grad_(True)
g = synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5
def step():
    rgb, _ = render_planes(stack, homos, H, W, spec)
    (gs,) = torch.autograd.grad(rgb, stack, g)
    return gs
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); print("eager  %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = step()
ref = step().clone()
gr.replay(); torch.cuda.synchronize()
print("graph == eager:", bool(torch.equal(out, ref)))
t0 = time.perf_counter()
for _ in range(50): gr.replay()
torch.cuda.synchronize(); print("graph  %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
