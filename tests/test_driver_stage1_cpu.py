"""Host logic of the stage-1 driver loop (videoloop3d_amd/train_3d.py, mirrors train_3d.py:20-95, 262-318).  CPU only, no kernels: the
iteration itself (run_iter) is replaced by a recorder."""
import os
import types

import numpy as np
import torch

from videoloop3d_amd import train_3d as drv


def test_vid2img_modes():
    vid = torch.rand(6, 3, 10, 12)
    assert torch.equal(drv.vid2img(vid, "first"), vid[0])
    assert torch.allclose(drv.vid2img(vid, "average"), vid.mean(0))
    import numpy as np
    assert np.allclose(drv.vid2img(vid, "median").numpy(), np.median(vid.numpy(), axis=0), atol=1e-7)      # train_3d.py:58: np.median (even clip: mean of the middle pair)
    assert not torch.equal(drv.vid2img(vid, "median"), vid.median(0).values)                                # ... not torch's lower median
    # 'dynamic' (configs/mpi_base.txt:10), k = 1: frames weighted by their colour distance from the temporal mean (train_3d.py:66-73)
    wgt = (vid - vid.mean(0, keepdim=True)).norm(dim=1, keepdim=True).clamp(1e-10, 999999)
    assert torch.allclose(drv.vid2img(vid, "dynamic"), (vid * wgt).sum(0) / wgt.sum(0), atol=1e-6)
    w2 = (0.5 * (vid - vid.mean(0, keepdim=True)).norm(dim=1, keepdim=True) + 0.5).clamp(1e-10, 999999)
    assert torch.allclose(drv.vid2img(vid, "dynamic0.5"), (vid * w2).sum(0) / w2.sum(0), atol=1e-6)
    assert drv.vid2img(vid, "blur5").shape == (3, 10, 12)
    static = vid[:1].repeat(6, 1, 1, 1)
    assert torch.allclose(drv.vid2img(static, "blur5")[:, 3:-3, 3:-3], drv._gaussian_blur(static[:1], 5)[0, :, 3:-3, 3:-3], atol=1e-6)


def test_loopable_mask_semantics():
    # utils.py:337-364: a pixel that rises AND falls by more than eps is loopable; one that only rises is not; a constant one is not
    F, h, w = 8, 16, 24
    vid = torch.full((F, 3, h, w), 0.5)
    t = torch.arange(F).float()
    vid[:, :, :, :8] = (0.5 + 0.3 * torch.sin(t * 2 * np.pi / F)).view(F, 1, 1, 1)        # oscillates: loopable
    vid[:, :, :, 8:16] = (0.2 + 0.08 * t).view(F, 1, 1, 1)                               # monotone rise: unloopable
    m = drv.compute_loopable_mask(vid)
    assert m.shape == (h, w) and m.dtype == torch.bool
    assert m[:, 1:6].all() and not m[:, 10:14].any() and not m[:, 18:].any()


def test_dataset_items():
    vids = [torch.rand(4, 3, 40, 64) for _ in range(2)]
    poses = torch.eye(4)[None, :3].repeat(2, 1, 1)
    K = torch.tensor([[50., 0, 32], [0, 50., 20], [0, 0, 1]])[None].repeat(2, 1, 1)
    ds = drv.MVPatchDataset((20, 32), vids, (16, 16), (4, 16), poses, K, mode="average")
    # h starts 0,4 ; w starts 0,16 -> 4 crops per view, (w, h) pairs with w slowest (utils.py:120-134)
    assert len(ds) == 8
    w0, h0, pose, intrin, crop, mask = ds[3]
    assert (w0, h0) == (16, 4) and crop.shape == (3, 16, 16) and mask.shape == (16, 16)
    assert torch.allclose(intrin, torch.tensor([[25., 0, 16 - 16], [0, 25., 10 - 4], [0, 0, 1]]))
    img = torch.nn.functional.interpolate(vids[0], size=(20, 32), mode="bilinear", align_corners=False).mean(0)
    assert torch.allclose(crop, img[:, 4:20, 16:32], atol=1e-6)
    # the caller's own stills / masks are taken as they are
    own_i, own_m = [torch.rand(3, 20, 32) for _ in range(2)], [torch.rand(20, 32) > 0.5 for _ in range(2)]
    ds2 = drv.MVPatchDataset((20, 32), vids, (16, 16), (4, 16), poses, K, images=own_i, dynmasks=own_m)
    assert torch.equal(ds2[5][4], own_i[1][:, 4:20, 0:16]) and torch.equal(ds2[5][5], own_m[1][4:20, 0:16].float())
    # a frame smaller than the crop: one full-frame item per view
    ds3 = drv.MVPatchDataset((10, 12), vids, (16, 16), (8, 16), poses, K)
    assert len(ds3) == 2 and ds3[0][4].shape == (3, 10, 12)


class _FakeMesh:
    def __init__(self, a):
        self.args, self.sparsified, self.opts = a, 0, 0

    def get_optimizer(self):
        self.opts += 1
        return types.SimpleNamespace(param_groups=[{"lr": None}], tag=self.opts)

    def get_lrate(self, step):
        return [("lr", 0.1 * 0.5 ** step)]

    def update_step(self, step):
        self.last_update = step

    def sparsify_faces(self, erode_num, alpha_thresh):
        self.sparsified += 1
        self.sparsify_args = (erode_num, alpha_thresh)

    def state_dict(self):
        return {"w": torch.zeros(1), "self.is_sparse": bool(self.sparsified)}


def test_epoch_loop_schedule(monkeypatch, tmp_path):
    a = types.SimpleNamespace(N_iters=5, sparsify_epoch=3, sparsify_erode=2, sparsify_alpha_thresh=0.05, density_loss_epoch=2,
                              density_loss_weight=0.02, patch_h_size=16, patch_w_size=16, patch_h_stride=4, patch_w_stride=16,
                              vid2img_mode="first", i_weights=2)
    vids = [torch.rand(3, 3, 20, 32) for _ in range(2)]
    poses = torch.eye(4)[None, :3].repeat(2, 1, 1)
    K = torch.tensor([[50., 0, 16], [0, 50., 10], [0, 0, 1]])[None].repeat(2, 1, 1)
    rec = []

    def fake_iter(nerf, opt, item, args, device):
        rec.append((opt.tag, opt.param_groups[0]["lr"], args.density_loss_weight, (item[0], item[1])))
        return (torch.zeros(()),) * 3 + ({},)
    monkeypatch.setattr(drv, "run_iter", fake_iter)
    mesh = _FakeMesh(a)
    steps = []
    out = drv.train(mesh, a, vids, poses, K, 20, 32, device="cpu", generator=torch.Generator().manual_seed(1), save_dir=str(tmp_path),
                    on_step=lambda e, i, *r: steps.append((e, i)))
    n = 8                                         # crops per epoch: 2 views x (2 x 2)
    assert out == {"iters": 5 * n, "epochs": 5, "sparsified_at": 3} and len(rec) == 5 * n
    assert steps[0] == (0, 0) and steps[-1] == (4, 5 * n - 1)
    # a new optimiser after sparsify_faces at epoch 3 (train_3d.py:282-285), called with the config's arguments
    assert [r[0] for r in rec[:3 * n]] == [1] * (3 * n) and [r[0] for r in rec[3 * n:]] == [2] * (2 * n)
    assert mesh.sparsified == 1 and mesh.sparsify_args == (2, 0.05)
    # the learning rate follows the ITERATION count (train_3d.py:303-306), the density weight ramps quadratically per epoch (:292-293)
    assert rec[0][1] == 0.1 and abs(rec[9][1] - 0.1 * 0.5 ** 9) < 1e-12
    dens = [rec[e * n][2] for e in range(5)]
    assert np.allclose(dens, [0.0, 0.02 / 9, 0.02 * 4 / 9, 0.02, 0.02]) and a.density_loss_weight == 0.02
    # every epoch visits every crop once, in a shuffled order
    for e in range(5):
        assert sorted(r[3] for r in rec[e * n:(e + 1) * n]) == sorted([(w, h) for w in (0, 16) for h in (0, 4)] * 2)
    assert rec[0:n] != rec[n:2 * n]
    # checkpoints with the reference's keys every i_weights epochs (:311-318)
    files = sorted(os.listdir(tmp_path))
    assert files == ["epoch_0001.tar", "epoch_0003.tar"]
    ck = torch.load(os.path.join(tmp_path, files[1]), weights_only=False)
    assert ck["epoch_i"] == 3 and ck["network_state_dict"]["self.is_sparse"] is True
    # resume skips the epochs before `start_epoch` (:278-279)
    rec.clear()
    out = drv.train(_FakeMesh(a), a, vids, poses, K, 20, 32, device="cpu", start_epoch=4)
    assert out["iters"] == n and len(rec) == n


def test_crop_aware_optimiser_rule():
    """args.crop_aware_adam = "auto": the four measured shapes of profiles/s1_crop_aware.py fall on the faster side."""
    from videoloop3d_amd.MPI import crop_aware_pays
    assert not crop_aware_pays((32, 1, 576, 1024, 4), 180, 320)        # the reference's native shape: 302 MB, host-bound
    assert not crop_aware_pays((32, 1, 792, 1408, 4), 720, 1280)       # 720p on 1.1x planes: the view is most of a plane
    assert not crop_aware_pays((32, 1, 1152, 2048, 4), 720, 1280)      # 720p on 1.6x planes: 1.2 GB, the view 39 % of a plane -- the zero-skip Adam took this one back (376 | 366 it/s)
    assert crop_aware_pays((32, 1, 1152, 2048, 4), 360, 640)
