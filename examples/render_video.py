#!/usr/bin/env python3
"""The offline renderer (scripts/script_render_video.py) on the device: frames per second of `videoloop3d_amd.render_video.render_frames`
at 720p, D = 32, a 50-frame loop -- the spiral (a new camera per frame: one launch each, like the reference's loop) and a fixed view (`--v r0`:
the loop's frames in batched calls).  Synthetic poses and weights."""
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def run(H=720, W=1280, planes=32, frames=50, views=8, dev="cuda:0"):
    from videoloop3d_amd.MPV import MPMeshVid
    from videoloop3d_amd import render_video as RV
    dev = torch.device(dev)
    rng = np.random.RandomState(3)
    rows = []
    for v in range(views):      # an LLFF poses_bounds array: small rotations, positions on an arc, (H, W, f), (near, far)
        a, b = np.radians(rng.uniform(-2, 2, 2))
        R = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]]) @ np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        t = np.array([0.3 * np.cos(2 * np.pi * v / views), 0.2 * np.sin(2 * np.pi * v / views), 0.0])
        rows.append(np.concatenate([np.concatenate([R, t[:, None], [[H], [W], [0.9 * W]]], 1).reshape(-1), [2.0, 50.0]]))
    poses, intrins, bds, rposes, rintr = RV.load_llff_poses(np.stack(rows), factor=1, render_frm=RV.default_render_frames(frames))
    ext, K, near, far = RV.reference_camera(poses, intrins, bds)
    args = types.SimpleNamespace(mpv_frm_num=frames, mpv_isloop=True, mpi_h_scale=1.1, mpi_w_scale=1.1, mpi_d=planes, atlas_grid_h=4, init_std=0.5,
                                 rgb_mlp_type="direct", rgb_activate="sigmoid", alpha_activate="sigmoid", bg_color="", scale_invariant=True, fp16=False,
                                 swd_patch_size=3, swd_patcht_size=3, swd_stride=2, swd_stridet=1, sparsity_loss_weight=0.0, rgb_smooth_loss_weight=0.0,
                                 a_smooth_loss_weight=0.0, density_loss_weight=0.0, d_smooth_loss_weight=0.0, optimizer="adam", lrate=0.1, lrate_decay=30)
    model = MPMeshVid(args, H, W, ext, K.astype(np.float64), near, far, device=dev).to(dev)
    out = {}
    for name, v in (("spiral", ""), ("fixed_view", "r0")):
        vp, vi, rt = RV.select_views_times(rposes, rintr, poses, intrins, frames, v, "")
        ve = RV.pose2extrin_np(vp)
        RV.render_frames(model, H, W, ve[:4], vi[:4], rt[:4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fr = RV.render_frames(model, H, W, ve, vi, rt)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"frames": len(rt), "frames_per_s": len(rt) / dt, "ms_per_frame": dt / len(rt) * 1e3}
    out["shape"] = f"{H}x{W}, D={planes}, T={frames}, planes {tuple(model.stack.shape[2:4])}, uint8 frames on the device"
    return out


if __name__ == "__main__":
    import __graft_entry__ as g
    g.build()
    print(json.dumps(run()))
