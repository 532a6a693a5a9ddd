"""The offline renderer's frame loop (videoloop3d_amd/render_video.py; scripts/script_render_video.py:129-139) on the MI355X: frames equal the
model's own eval forward frame by frame, whether a fixed view's loop goes through the renderer as one call or one call per frame."""
import types

import numpy as np
import pytest
import torch

from videoloop3d_amd import render_video as RV
from videoloop3d_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import __graft_entry__ as ge
    ge.build()
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_render_frames_equal_per_frame_eval(dev, golden):
    from videoloop3d_amd.MPV import MPMeshVid
    g = golden("g18_render_poses.npz")
    T = 6
    # a small frame with the poses of G18 (view "a", setting s0: 360 x 640 at factor 2 -> scaled down again for the test)
    poses, intrins, bds, rposes, rintr = RV.load_llff_poses(g["a_poses_bounds"], factor=2, recenter=True, bd_factor=(0.9, 1.1), render_frm=12, render_scaling=1.0)
    H, W = 72, 128
    sc = np.diag([W / (2 * intrins[0, 0, 2]), H / (2 * intrins[0, 1, 2]), 1.0]).astype(np.float32)
    intrins, rintr = sc @ intrins, sc @ rintr
    ext, K, near, far = RV.reference_camera(poses, intrins, bds)
    args = types.SimpleNamespace(mpv_frm_num=T, mpv_isloop=True, mpi_h_scale=1.1, mpi_w_scale=1.1, mpi_d=6, atlas_grid_h=2, init_std=0.5,
                                 rgb_mlp_type="direct", rgb_activate="sigmoid", alpha_activate="sigmoid", bg_color="", scale_invariant=True,
                                 fp16=False, swd_patch_size=3, swd_patcht_size=3, swd_stride=2, swd_stridet=1, sparsity_loss_weight=0.0,
                                 rgb_smooth_loss_weight=0.0, a_smooth_loss_weight=0.0, density_loss_weight=0.0, d_smooth_loss_weight=0.0,
                                 optimizer="adam", lrate=0.1, lrate_decay=30)
    model = MPMeshVid(args, H, W, ext, K.astype(np.float64), near, far).to(dev)
    with torch.no_grad():
        model.stack.copy_(synth.make_plane_stack(*model.stack.shape[:4], seed=5, device=dev) * 0.8)
    for v, t in (("", ""), ("r3", ""), ("1", "0:6,5:2"), ("", "0,5,11")):
        vp, vi, rt = RV.select_views_times(rposes, rintr, poses, intrins, T, v, t)
        ve = RV.pose2extrin_np(vp)
        frames = RV.render_frames(model, H, W, ve, vi, rt)
        assert frames.shape == (len(rt), H, W, 3) and frames.dtype == torch.uint8 and frames.device.type == "cuda"
        model.eval()
        with torch.no_grad():
            for i in range(len(rt)):
                rgb, _ = model(H, W, torch.tensor(ve[i:i + 1], dtype=torch.float32), torch.tensor(vi[i:i + 1]), torch.tensor(rt[i:i + 1]))
                want = (255 * rgb.permute(0, 2, 3, 1).clamp(0, 1)).to(torch.uint8)[0]
                assert torch.equal(frames[i], want), (v, t, i)
        one = RV.render_frames(model, H, W, ve, vi, rt, max_batch=1)          # one launch per frame, like the reference's loop
        assert torch.equal(one, frames)
    assert float(frames.float().std()) > 1.0
