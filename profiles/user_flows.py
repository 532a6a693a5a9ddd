#!/usr/bin/env python3
"""Wall time of the things a user does AROUND the hot path at the shipped stage-2 shape (D = 32, T = 50, 360 x 640 frames on 1.1x planes, 8 views of 75
frames): model construction, the pyramid's lod switches, the crop dataset per level, packing, checkpoints, the NN-error metric.  python profiles/user_flows.py"""
import os, sys, time, types, warnings, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import numpy as np
import torch
import __graft_entry__ as g
g.build()
from stage2_schedule import make_views
from videoloop3d_amd.MPV import MPMeshVid
from videoloop3d_amd.train_3dvid import MVVidPatchDataset
from videoloop3d_amd import evaluations, synth

dev = torch.device("cuda:0")
H, W, V, D, T, F = 360, 640, 8, 32, 50, 75
args = types.SimpleNamespace(mpv_frm_num=T, mpv_isloop=True, mpi_h_scale=1.1, mpi_w_scale=1.1, mpi_d=D, atlas_grid_h=4, init_std=0.02, rgb_mlp_type="direct",
                             rgb_activate="sigmoid", alpha_activate="sigmoid", bg_color="", scale_invariant=True, add_uv_noise=False, fp16=False, swd_patch_size=3,
                             swd_patcht_size=3, swd_stride=2, swd_stridet=1, sparsity_loss_weight=0.0, rgb_smooth_loss_weight=0.2, a_smooth_loss_weight=0.2,
                             density_loss_weight=0.0, d_smooth_loss_weight=0.0, optimizer="adam", lrate=0.5, lrate_decay=100, mpi_h_verts=36, mpi_w_verts=64)
out = {}


def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize(); out[name] = round(time.perf_counter() - t0, 4)
    return r


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    poses, intrins, vids = make_views(V, H, W, F, dev)
    K = intrins[0].numpy().astype(np.float64)
    timed("warmup_model", lambda: MPMeshVid(args, H, W, np.eye(4), K, 1.0, 100.0, device=dev).to(dev))
    m = timed("construct_on_device", lambda: MPMeshVid(args, H, W, np.eye(4), K, 1.0, 100.0, device=dev).to(dev))
    timed("lod_0.25", lambda: m.lod(0.25))
    timed("lod_0.5", lambda: m.lod(0.5))
    timed("lod_1.0", lambda: m.lod(1.0))
    timed("get_optimizer", lambda: m.get_optimizer(step=0))
    other = dict(loss_name="gpnn_lm", patch_size=3, patcht_size=3, stride=2, stridet=1, alpha=10000.0, rou="-2", scaling=0.1, dist_fn="mse", macro_block=65, factor=1)
    for hw in ((90, 160), (180, 320), (360, 640)):
        timed(f"dataset_{hw[0]}x{hw[1]}", lambda: MVVidPatchDataset(hw, vids, (180, 320), (90, 160), poses, intrins, loss_configs=[other] * V))
    sd = timed("state_dict", lambda: m.state_dict())
    timed("init_from_mpi_dense", lambda: m.init_from_mpi(sd))
    x = synth.make_video(T, H, W, seed=1, device=dev); y = synth.make_video(F, H, W, seed=2, device=dev)
    timed("compute_nnerr_first", lambda: evaluations.compute_nnerr(x, y))
    timed("compute_nnerr", lambda: evaluations.compute_nnerr(x, y))
print(json.dumps(out))
