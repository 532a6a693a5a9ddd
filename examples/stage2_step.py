#!/usr/bin/env python3
"""End-to-end stage-2 training iterations on the drop-in module (SURVEY §8f-3 'next' row; reference loop:
train_3dvid.py:214-255 run_iter): MPMeshVid.forward (render crop -> looping loss + regularisers) -> backward -> Adam.

Shapes follow configs/mpv_base.txt: 720p data at factor 2 (360x640), mpi_{h,w}_scale 1.1, mpi_d 32, 50 frames,
180x320 crops with a shifted principal point, other-view and ref-view loss configurations.  Synthetic scene/video.
The reference authors' run is ~0.4-0.9 it/s on an RTX 3090 (derived)."""
import argparse
import json
import os
import sys
import time
import types
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


# untimed iterations in front of every timed loop: one per distinct crop of the six the loops cycle through (the optimiser's window buffers and the
# allocator have seen every size before the clock starts: a buffer growing inside the timed loop is a multi-GB hipMalloc, docs/measurement_log.md round 6)
WARM = 6

def run(iters=10, frames=50, planes=32, smooth=0.2, dev="cuda:0", crop=(180, 320), full=(360, 640), baseline=0.03):
    from videoloop3d_amd import synth
    from videoloop3d_amd.MPV import MPMeshVid
    dev = torch.device(dev)
    H, W = full
    h, w = crop
    args = types.SimpleNamespace(
        mpv_frm_num=frames, mpv_isloop=True, mpi_h_scale=1.1, mpi_w_scale=1.1, mpi_d=planes, atlas_grid_h=4, init_std=0.02,
        rgb_mlp_type="direct", rgb_activate="sigmoid", alpha_activate="sigmoid", bg_color="", scale_invariant=True,
        add_uv_noise=False, fp16=False, swd_patch_size=3, swd_patcht_size=3, swd_stride=2, swd_stridet=1,
        sparsity_loss_weight=0.0, rgb_smooth_loss_weight=smooth, a_smooth_loss_weight=smooth, density_loss_weight=0.0,
        d_smooth_loss_weight=0.0)
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], np.float64)
    model = MPMeshVid(args, H, W, np.eye(4), K, 1.0, 100.0, device=dev).to(dev).train()
    args.optimizer, args.lrate, args.lrate_decay = "adam", 0.5 * 0.01, 30
    opt = model.get_optimizer(0)        # MPV.py:199-214 (Adam, betas (0.9, 0.999), eps 6e-8): the crop-aware WindowAdam on a dense model
    if hasattr(opt, "acknowledge_fused_backward"):
        opt.acknowledge_fused_backward()      # this loop steps once per backward
    a = np.radians(0.5)
    tar = np.eye(4)
    tar[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
    tar[:3, 3] = [baseline, baseline / 3, 0.0]       # camera offset in units of the nearest plane's depth: the parallax between the planes
    tar_e = torch.tensor(tar)[None]          # poses stay on the host, as the DataLoader yields them (train_3dvid.py:214-216): the module
                                             # turns them into homographies there and uploads 1 KiB -- no device round trip per iteration
    res = synth.hash_uniform((1, 75, 3, h, w), seed=8, device=dev)
    cfgs = {
        "other": dict(loss_name=["gpnn_lm"], loss_gain=torch.tensor([1.0]), macro_block=torch.tensor([65]), patch_size=torch.tensor([3]),
                      stride=torch.tensor([2]), patcht_size=torch.tensor([3]), stridet=torch.tensor([1]), alpha=torch.tensor([10000.0]),
                      dist_fn=["mse"], rou=["-2"], scaling=torch.tensor([0.1])),
        "ref": dict(loss_name=["gpnn_lm"], loss_gain=torch.tensor([3.5]), macro_block=torch.tensor([65]), patch_size=torch.tensor([11]),
                    stride=torch.tensor([4]), patcht_size=torch.tensor([3]), stridet=torch.tensor([1]), alpha=torch.tensor([0.0]),
                    dist_fn=["mse"], rou=["-2"], scaling=torch.tensor([0.1])),
    }
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, cfg in cfgs.items():
            for it in range(iters + WARM):
                if it == WARM:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                Kc = K.copy()
                Kc[0, 2] -= 90 + (it % 3) * 40                                   # crop offset (utils.py:196-200)
                Kc[1, 2] -= 45 + (it % 2) * 60
                tar_k = torch.tensor(Kc)[None]
                opt.zero_grad(set_to_none=True)
                _, extra = model(h, w, tar_e, tar_k, res=res, losscfg=cfg)
                loss = extra["swd"].sum()
                for k in ("rgb_smooth", "a_smooth"):
                    if k in extra:
                        loss = loss + smooth * extra[k].sum()
                loss.backward()
                opt.step()
            torch.cuda.synchronize()
            out[name] = {"iters_per_s": iters / (time.perf_counter() - t0), "loss": float(loss.detach())}
        # the same iteration on a tile-culled model (what stage 2 sees after sparsify_faces, MPI.py:288-442): ~20 % of the quads kept,
        # half of them dynamic; culled kernels + static tiles tied across frames (videoloop3d_amd/tiles.py)
        from videoloop3d_amd import tiles
        QH, QW = 35, 63
        qy, qx = torch.meshgrid(torch.arange(QH, device=dev), torch.arange(QW, device=dev), indexing="ij")
        keep = torch.zeros((planes, QH, QW), dtype=torch.bool, device=dev)
        for d in range(planes):
            cy, cx = (7 * d + 3) % QH, (11 * d + 5) % QW
            keep[d] = ((qy - cy).abs() <= QH // 5) & ((qx - cx).abs() <= QW // 4)
        model.register_buffer("quad_keep", keep)
        model.register_buffer("quad_dyn", keep & ((qy + qx) % 2 == 0)[None])
        model.is_sparse = model.has_dyn = True
        with torch.no_grad():
            tiles.cull_stack_(model.stack.data, keep)
        model._install_tie_hook()
        opt = model.get_optimizer(0)                       # the same Adam update on the kept texels only
        if hasattr(opt, "acknowledge_fused_backward"):
            opt.acknowledge_fused_backward()      # this loop steps once per backward
        cfg = cfgs["other"]
        for it in range(iters + WARM):
            if it == WARM:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            Kc = K.copy()
            Kc[0, 2] -= 90 + (it % 3) * 40
            Kc[1, 2] -= 45 + (it % 2) * 60
            opt.zero_grad(set_to_none=True)
            _, extra = model(h, w, tar_e, torch.tensor(Kc)[None], res=res, losscfg=cfg)
            loss = extra["swd"].sum()
            for k in ("rgb_smooth", "a_smooth"):
                if k in extra:
                    loss = loss + smooth * extra[k].sum()
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        out["other_tile_culled"] = {"iters_per_s": iters / (time.perf_counter() - t0), "loss": float(loss.detach()),
                                    "kept_quads": float(keep.float().mean()),
                                    "texture_bytes": int(model.stack.numel() * 4), "texture_and_adam_state_bytes": int(model.stack.numel() * 12)}
        # ... and on the PACKED model (videoloop3d_amd/packed.py: static blocks once, dynamic blocks per frame, culled blocks not stored --
        # the reference's static / dynamic atlases, MPI.py:364-436): same kernels on the hot path, same bits, a fraction of the memory
        model.pack_()
        torch.cuda.empty_cache()
        opt = model.get_optimizer(0)
        if hasattr(opt, "acknowledge_fused_backward"):
            opt.acknowledge_fused_backward()      # this loop steps once per backward
        for it in range(iters + WARM):
            if it == WARM:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            Kc = K.copy()
            Kc[0, 2] -= 90 + (it % 3) * 40
            Kc[1, 2] -= 45 + (it % 2) * 60
            opt.zero_grad(set_to_none=True)
            _, extra = model(h, w, tar_e, torch.tensor(Kc)[None], res=res, losscfg=cfg)
            loss = extra["swd"].sum()
            for k in ("rgb_smooth", "a_smooth"):
                if k in extra:
                    loss = loss + smooth * extra[k].sum()
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        lay = model.packed
        out["other_tile_culled_packed"] = {"iters_per_s": iters / (time.perf_counter() - t0), "loss": float(loss.detach()),
                                           "texture_bytes": int(lay.pool_bytes), "texture_and_adam_state_bytes": int(3 * lay.pool_bytes),
                                           "fraction_of_dense": lay.pool_bytes / lay.dense_bytes,
                                           "blocks": {"static": lay.n_static, "dynamic": lay.n_dynamic, "of": int(lay.blocks.numel())}}
    out["shape"] = (f"D={planes} T={frames} stack {tuple(model.stack_dims()) + (4,)} crop {h}x{w} of {H}x{W}, Ty=75, "
                    f"smooth weights {smooth}, Adam")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--smooth", type=float, default=0.2)
    ap.add_argument("--baseline", type=float, default=0.03, help="camera offset / nearest plane depth (0.03: ~17 texels of parallax; 0.15: ~85)")
    a = ap.parse_args()
    import __graft_entry__ as g
    g.build()
    print(json.dumps(run(a.iters, smooth=a.smooth, baseline=a.baseline)))
