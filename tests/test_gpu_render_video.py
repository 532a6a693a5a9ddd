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


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_frame_run_in_place_equals_the_gathered_frames(dev, dtype):
    """render.render_frame_run (vl3d_render_fwd_frames: a run of frames of a longer clip read where it lies) == render_planes on the gathered
    stack[:, ts], bit for bit: single frames, even and odd runs (the frame-pair kernel's tail), first / last frames; and it refuses runs that
    leave the clip."""
    from videoloop3d_amd.render import RenderSpec, render_frame_run, render_planes
    D, T, Hs, Ws, H, W = 5, 7, 50, 70, 44, 60
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=3, device=dev).to(dtype)
    homos = torch.eye(3).repeat(D, 1, 1)
    for d in range(D):
        homos[d, 0, 2], homos[d, 1, 2], homos[d, 0, 0] = 3.0 + 0.7 * d, 2.0 - 0.3 * d, 1.0 + 0.01 * d
    homos = homos.to(dev)
    spec = RenderSpec.mpv()
    for t0, n in ((0, 1), (3, 1), (6, 1), (2, 2), (1, 3), (0, 7), (4, 3)):
        rgb, alpha = render_frame_run(stack, t0, n, homos, H, W, spec)
        rgb_g, alpha_g = render_planes(stack[:, t0:t0 + n].contiguous(), homos, H, W, spec)
        assert torch.equal(rgb, rgb_g) and torch.equal(alpha, alpha_g), (t0, n)
    buf = (torch.empty((2, H, W, 3), device=dev), torch.empty((2, H, W), device=dev))
    r2, a2 = render_frame_run(stack, 5, 2, homos, H, W, spec, out=buf)
    assert r2.data_ptr() == buf[0].data_ptr() and torch.equal(r2, render_planes(stack[:, 5:7].contiguous(), homos, H, W, spec)[0])
    with pytest.raises(RuntimeError):
        render_frame_run(stack, 6, 2, homos, H, W, spec)


def test_render_frames_in_place_equals_the_module_loop(dev, golden):
    """render_video.render_frames: the in-place path (homographies of the path uploaded once, frames read where they lie, one uint8 conversion
    per chunk) gives the frames of the loop over the module's eval forward -- spiral (a camera per frame), fixed view, wrap-around of the
    clip, a background colour, chunks smaller than a run."""
    from videoloop3d_amd.MPV import MPMeshVid
    g = golden("g18_render_poses.npz")
    T = 6
    poses, intrins, bds, rposes, rintr = RV.load_llff_poses(g["a_poses_bounds"], factor=2, recenter=True, bd_factor=(0.9, 1.1), render_frm=14, render_scaling=1.0)
    H, W = 72, 128
    sc = np.diag([W / (2 * intrins[0, 0, 2]), H / (2 * intrins[0, 1, 2]), 1.0]).astype(np.float32)
    intrins, rintr = sc @ intrins, sc @ rintr
    ext, K, near, far = RV.reference_camera(poses, intrins, bds)
    for bg, sparse in (("", False), ("0.1#0.5#0.9", False), ("", True)):
        args = types.SimpleNamespace(mpv_frm_num=T, mpv_isloop=True, mpi_h_scale=1.1, mpi_w_scale=1.1, mpi_d=6, atlas_grid_h=2, init_std=0.5,
                                     rgb_mlp_type="direct", rgb_activate="sigmoid", alpha_activate="sigmoid", bg_color=bg, scale_invariant=True,
                                     fp16=False, swd_patch_size=3, swd_patcht_size=3, swd_stride=2, swd_stridet=1, sparsity_loss_weight=0.0,
                                     rgb_smooth_loss_weight=0.0, a_smooth_loss_weight=0.0, density_loss_weight=0.0, d_smooth_loss_weight=0.0,
                                     optimizer="adam", lrate=0.1, lrate_decay=30)
        model = MPMeshVid(args, H, W, ext, K.astype(np.float64), near, far).to(dev)
        with torch.no_grad():
            model.stack.copy_(synth.make_plane_stack(*model.stack.shape[:4], seed=5, device=dev) * 0.8)
        if sparse:      # a tile-culled model: samples inside culled quads are not covered (the culled forward, on frame runs too)
            model.quad_keep = synth.hash_uniform((6, 5, 7), seed=12, device=dev) > 0.45
            model.quad_dyn = model.quad_keep.clone()
            model.is_sparse = model.has_dyn = True
        for v, t in (("", ""), ("r3", ""), ("1", "0:6,5:2"), ("", "0,5,11")):
            vp, vi, rt = RV.select_views_times(rposes, rintr, poses, intrins, T, v, t)
            ve = RV.pose2extrin_np(vp)
            loop = RV.render_frames(model, H, W, ve, vi, rt, in_place=False)
            if sparse:
                model.is_sparse = False
                assert not torch.equal(RV.render_frames(model, H, W, ve, vi, rt, in_place=False), loop)      # (the quad map matters)
                model.is_sparse = True
            for chunk in (64, 4, 1):
                assert torch.equal(RV.render_frames(model, H, W, ve, vi, rt, max_batch=chunk), loop), (bg, sparse, v, t, chunk)
