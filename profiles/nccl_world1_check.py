import os, torch, torch.distributed as dist, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from videoloop3d_amd.dist import all_gather_frame, plan_bands
from videoloop3d_amd import synth
from videoloop3d_amd.render import RenderSpec
from videoloop3d_amd.utils_mpi import compute_homography, make_depths
H, W, D = 720, 1280, 4
ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3), make_depths(D, 1.0, 100.0).flip(0)[None])[0]
bands = plan_bands(homos, H, W, H, 1, RenderSpec.mpv())
x = torch.rand(3, bands[0].rows, W, 3, device="cuda")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    f = all_gather_frame(x, bands)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("nccl all_gather ok", f.shape, bool(torch.equal(f, x)))
dist.barrier(); dist.destroy_process_group()
