"""Host logic of the stage-2 driver loop (videoloop3d_amd/train_3dvid.py, mirrors train_3dvid.py:22-66,103-119,160-189,
262-290 and utils.py:115-134) and the driver hooks of MPMeshVid (MPV.py:140-229).  CPU only, no kernels."""
import types

import numpy as np
import pytest
import torch

from videoloop3d_amd import train_3dvid as drv


def test_generate_patchinfo_covers_frame_in_reference_order():
    wh, pad = drv.generate_patchinfo(360, 640, (180, 320), (90, 160))
    # utils.py:120-134: h starts 0..180 step 90, w starts 0..320 step 160; (w,h) pairs with w slowest
    assert pad == [0, 0, 0, 0]
    assert wh.tolist()[:4] == [[0, 0], [0, 90], [0, 180], [160, 0]] and len(wh) == 9
    wh, pad = drv.generate_patchinfo(100, 130, (64, 64), (48, 48))
    hs, ws = sorted(set(wh[:, 1].tolist())), sorted(set(wh[:, 0].tolist()))
    assert hs == [0, 48] and ws == [0, 48, 96]
    assert pad == [0, 96 + 64 - 130, 0, 48 + 64 - 100]
    assert all(0 <= p < 48 for p in (pad[1], pad[3]))


def _args(**kw):
    a = dict(pyr_minimal_dim=-1, pyr_stage="20,50", N_iters=100, pyr_factor=0.5, pyr_num_epoch=7,
             lrate=0.1, lrate_decay=30, optimize_verts_gain=1, optimizer="adam", lrate_adaptive=True)
    a.update(kw)
    return types.SimpleNamespace(**a)


def test_pyramid_schedule_both_modes():
    f, hw, ep = drv.pyramid_schedule(_args(), 360, 640)
    assert f == [0.25, 0.5, 1.0] and hw == [(90, 160), (180, 320), (360, 640)] and ep == [20, 30, 50]
    f, hw, ep = drv.pyramid_schedule(_args(pyr_minimal_dim=60), 360, 640)
    # int(log(60/360)/log(0.5)) + 1 = 3 levels
    assert f == [0.25, 0.5, 1.0] and ep == [7, 7, 7]


def test_loss_configs_pick_reference_views():
    a = types.SimpleNamespace(loss_name="gpnn_lm", swd_patch_size=3, swd_patcht_size=3, swd_stride=2, swd_stridet=1,
                              swd_alpha=10000, swd_rou="-2", swd_scaling=0.1, swd_dist_fn="mse", swd_macro_block=65,
                              swd_factor=1, loss_name_ref="gpnn_lm", swd_loss_gain_ref=3.5, swd_patch_size_ref=11,
                              swd_patcht_size_ref=3, swd_stride_ref=4, swd_stridet_ref=1, swd_alpha_ref=0, swd_rou_ref="-2",
                              swd_scaling_ref=0.1, swd_dist_fn_ref="mse", swd_factor_ref=1, loss_ref_idx="1,3")
    cfgs = drv.loss_configs(a, 5)
    assert [c["patch_size"] for c in cfgs] == [3, 11, 3, 11, 3]
    assert "loss_gain" in cfgs[1] and "loss_gain" not in cfgs[0]
    assert [c["patch_size"] for c in drv.loss_configs(a, 5, train_view=[3, 4])] == [11, 3]


def test_dataset_crops_and_intrinsics():
    vids = [torch.rand(5, 3, 40, 64) for _ in range(2)]
    poses = torch.eye(4)[None, :3].repeat(2, 1, 1)
    K = torch.tensor([[50., 0, 32], [0, 50., 20], [0, 0, 1]])[None].repeat(2, 1, 1)
    ds = drv.MVVidPatchDataset((20, 32), vids, (16, 16), (8, 16), poses, K, loss_configs=[{"a": 1}, {"a": 2}])
    # h starts 0,8 (pad 4), w starts 0,16 -> 4 crops per view
    assert len(ds) == 8
    w0, h0, pose, intrin, crop, cfg = ds[3]
    assert (w0, h0) == (16, 8) and crop.shape == (5, 3, 16, 16) and cfg == {"a": 1}
    # resized intrinsics (x0.5) then principal point shifted by the crop origin (utils.py:196-200)
    assert torch.allclose(intrin, torch.tensor([[25., 0, 16 - 16], [0, 25., 10 - 8], [0, 0, 1]]))
    assert ds[7][5] == {"a": 2}
    # a frame smaller than the crop: one full-frame item per view
    ds2 = drv.MVVidPatchDataset((10, 12), vids, (16, 16), (8, 16), poses, K, loss_configs=[{}, {}])
    assert len(ds2) == 2 and ds2[0][4].shape == (5, 3, 10, 12)


def test_pose2extrin():
    p = torch.tensor([[[0., -1, 0, 1], [1, 0, 0, 2], [0, 0, 1, 3]]])
    e = drv.pose2extrin_torch(p)
    full = torch.cat([p, torch.tensor([[[0., 0, 0, 1]]])], 1)
    assert torch.allclose(e @ full, torch.eye(4)[None], atol=1e-6)


def _mpv_args(**kw):
    a = dict(mpv_frm_num=3, mpv_isloop=True, mpi_h_scale=1.0, mpi_w_scale=1.0, mpi_d=2, atlas_grid_h=1, init_std=0.5,
             rgb_mlp_type="direct", rgb_activate="sigmoid", alpha_activate="sigmoid", bg_color="", scale_invariant=True,
             fp16=False, swd_patch_size=3, swd_patcht_size=3, swd_stride=2, swd_stridet=1, lrate=0.1, lrate_decay=30,
             optimize_verts_gain=2, optimizer="adam", optimize_geo_start=5)
    a.update(kw)
    return types.SimpleNamespace(**a)


def test_mpmeshvid_driver_hooks():
    from videoloop3d_amd.MPV import MPMeshVid
    m = MPMeshVid(_mpv_args(), 20, 32, np.eye(4), np.array([[30., 0, 16], [0, 30., 10], [0, 0, 1]]), 1.0, 100.0)
    # MPV.py:216-225: lr * 0.1 ** (step / (lrate_decay * 1000))
    (n0, lr), (n1, vlr) = m.get_lrate(15000)
    assert (n0, n1) == ("lr", "vertlr") and lr == pytest.approx(0.1 * 0.1 ** 0.5) and vlr == pytest.approx(2 * lr)
    opt = m.get_optimizer(0)
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["eps"] == 6e-8 and opt.param_groups[0]["lr"] == pytest.approx(0.1)
    assert isinstance(MPMeshVid(_mpv_args(optimizer="sgd"), 20, 32, np.eye(4), np.eye(3), 1.0, 100.0).get_optimizer(0), torch.optim.SGD)
    with pytest.raises(RuntimeError):
        MPMeshVid(_mpv_args(optimizer="lbfgs"), 20, 32, np.eye(4), np.eye(3), 1.0, 100.0).get_optimizer(0)
    m.update_step(4)
    assert not m.optimize_geometry
    m.update_step(5)
    assert m.optimize_geometry
    # lod: planes resized, texel scale follows, parameter re-registered (the optimiser is re-created by the driver)
    full = m.stack.detach().clone()
    aa = bool(getattr(m.args, "lod_antialias", False))      # (torchvision 0.11's Resize of a tensor: no antialiasing -- the release the reference pins)
    m.lod(0.5)
    assert m.stack.shape == (2, 3, 10, 16, 4) and m.stack.requires_grad
    assert m.spec.scale == pytest.approx((15 / 31, 9 / 19))
    ref = torch.nn.functional.interpolate(full.permute(0, 1, 4, 2, 3).reshape(6, 4, 20, 32), size=(10, 16), mode="bilinear",
                                          align_corners=False, antialias=aa).reshape(2, 3, 4, 10, 16).permute(0, 1, 3, 4, 2)
    assert torch.equal(m.stack.detach(), ref)
    m.lod(1.0)
    assert m.stack.shape == (2, 3, 20, 32, 4) and m.spec.scale == pytest.approx((1.0, 1.0))
    # args.lod_antialias = True: torchvision >= 0.17's Resize of a tensor (differs from the pinned release's on a down-sampling call only)
    m2 = MPMeshVid(_mpv_args(lod_antialias=True), 20, 32, np.eye(4), np.array([[30., 0, 16], [0, 30., 10], [0, 0, 1]]), 1.0, 100.0)
    full2 = m2.stack.detach().clone()
    m2.lod(0.5)
    planes = full2.permute(0, 1, 4, 2, 3).reshape(6, 4, 20, 32)
    want = {aa_: torch.nn.functional.interpolate(planes, size=(10, 16), mode="bilinear", align_corners=False, antialias=aa_)
            .reshape(2, 3, 4, 10, 16).permute(0, 1, 3, 4, 2) for aa_ in (True, False)}
    assert torch.equal(m2.stack.detach(), want[True]) and not torch.equal(want[True], want[False])


# ---- product helpers against the reference goldens (plain torch: run on the CPU) ------------------------------------------------
def test_product_geometry_helpers_match_the_reference_goldens(golden):
    """videoloop3d_amd.utils_mpi.make_depths / compute_homography against G1 and gen_mpi_vertices against G12 -- the PRODUCT
    functions directly (the oracle's copies are checked in test_oracle_golden.py)."""
    import numpy as np
    import torch
    from videoloop3d_amd import utils_mpi as U
    T = lambda a: torch.from_numpy(np.asarray(a))
    g = golden("g1_homography.npz")
    assert float((U.make_depths(8, 1.0, 100.0) - T(g["depths"])).abs().max()) == 0
    for ci in range(2):
        h = U.compute_homography(T(g[f"c{ci}_src_ext"]), T(g[f"c{ci}_src_K"]), T(g[f"c{ci}_tar_ext"]), T(g[f"c{ci}_tar_K"]),
                                 T(g[f"c{ci}_normal"]), T(g[f"c{ci}_dist"]))
        assert float((h - T(g[f"c{ci}_homo"])).abs().max()) <= 1e-6
    g = golden("g12_helpers.npz")
    v = U.gen_mpi_vertices(40, 60, T(g["K"]), 5, 7, T(g["planedepth"]))
    assert v.shape == tuple(g["verts"].shape) and float((v - T(g["verts"])).abs().max()) <= 1e-6


def test_patch3d_mse_and_avg_match_the_reference_goldens(golden):
    """utils_vid.py:437-445 (the two loss_name alternatives that stay in PyTorch)."""
    import numpy as np
    import torch
    from videoloop3d_amd.utils_vid import Patch3DAvg, Patch3DMSE
    g = golden("g12_helpers.npz")
    xa, ya = torch.from_numpy(g["xa"]), torch.from_numpy(g["ya"])
    assert abs(float(Patch3DMSE(xa, ya)) - float(g["mse"])) <= 1e-7
    assert abs(float(Patch3DMSE(ya, xa)) - float(g["mse_rev"])) <= 1e-7
    assert abs(float(Patch3DAvg(xa, ya)) - float(g["avg"])) <= 1e-7


def test_unit_grad_is_cached_per_module_and_follows_the_loss():
    """train_3dvid.unit_grad: d loss / d loss = 1 from a tensor cached on the module (no fill launch per iteration); a loss of another dtype / shape gets
    a new one."""
    from videoloop3d_amd.train_3dvid import unit_grad
    mod = types.SimpleNamespace()
    l32 = torch.tensor(2.0, requires_grad=True) * 3
    one = unit_grad(mod, l32)
    assert one.shape == l32.shape and one.dtype == l32.dtype and float(one) == 1.0 and not one.requires_grad
    assert unit_grad(mod, l32 * 2) is one
    l64 = torch.tensor(2.0, dtype=torch.float64, requires_grad=True) * 3
    assert unit_grad(mod, l64).dtype == torch.float64
    l1 = torch.ones(1, requires_grad=True) * 3
    assert unit_grad(mod, l1).shape == (1,)
    w = torch.tensor(2.0, requires_grad=True)
    (w * 3).backward(unit_grad(mod, w * 3))
    assert float(w.grad) == 3.0
