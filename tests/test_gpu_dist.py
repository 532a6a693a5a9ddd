"""The N-GPU path on the one GPU a test box has: RCCL (backend 'nccl') process group of world size 1, one rank's row band rendered
from its band-local stack rows, the composited band all-gathered on a side stream.  (World sizes 2 and 3 of the same code run on
CPU/gloo in test_dist_cpu.py; the 8-GPU run is the driver's.)"""
import os
import socket

import pytest
import torch

from videoloop3d_amd import synth

pytestmark = pytest.mark.gpu


def test_rccl_world1_band_render_and_all_gather():
    import torch.distributed as dist
    import __graft_entry__ as g
    g.build()
    from videoloop3d_amd.dist import all_gather_frame, plan_bands, render_band
    from videoloop3d_amd.render import RenderSpec, render_planes
    from videoloop3d_amd.utils_mpi import compute_homography, make_depths
    dev = torch.device("cuda:0")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        D, T, H, W = 4, 2, 96, 160
        spec = RenderSpec.mpv()
        ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
        homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                                   make_depths(D, 1.0, 100.0).flip(0)[None])[0]
        stack = synth.make_plane_stack(D, T, H, W, seed=2, device=dev)
        full, _ = render_planes(stack, homos.to(dev), H, W, spec)
        # this rank's band of a 3-way split, rendered from the band-local rows only, then gathered (world 1: itself)
        bands3 = plan_bands(homos, H, W, H, 3, spec)
        b = bands3[1]
        band_rgb, _ = render_band(stack[:, :, b.src0:b.src1].contiguous(), homos.to(dev), b, W, H, spec)
        assert float((band_rgb - full[:, b.row0:b.row0 + b.rows]).abs().max()) <= 2e-5
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            frame = all_gather_frame(band_rgb, [b])
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(frame, band_rgb)
        dist.barrier()
    finally:
        dist.destroy_process_group()
