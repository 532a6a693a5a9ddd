"""The stage-1 driver loop (videoloop3d_amd/train_3d.py; reference train_3d.py:189-318) on the MI355X against the SAME loop written on the
CPU oracle (oracle/mpv_oracle.mpi_forward + the reference's torch loss chain + torch.optim.Adam): the loss trajectory of a tiny scene."""
import types

import numpy as np
import pytest
import torch

from oracle import mpv_oracle
from videoloop3d_amd import synth
from videoloop3d_amd import train_3d as drv
from videoloop3d_amd.train_3dvid import pose2extrin_torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import __graft_entry__ as ge
    ge.build()
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _args(**kw):
    a = dict(mpi_h_scale=1.3, mpi_w_scale=1.3, mpi_d=6, atlas_grid_h=2, rgb_mlp_type="direct", rgb_activate="sigmoid",
             alpha_activate="sigmoid", bg_color="", learn_loop_mask=True, upsample_stage="", scale_invariant=True,
             sparsity_loss_weight=0.004, rgb_smooth_loss_weight=0.2, a_smooth_loss_weight=0.5, density_loss_weight=0.02,
             d_smooth_loss_weight=0.0, l_smooth_loss_weight=0.0, optimizer="adam", lrate=0.05, lrate_decay=100,
             N_iters=2, sparsify_epoch=-1, density_loss_epoch=0, patch_h_size=24, patch_w_size=32, patch_h_stride=20, patch_w_stride=28,
             vid2img_mode="average", add_intrin_noise=False, i_weights=1000, mpi_h_verts=5, mpi_w_verts=7)
    a.update(kw)
    return types.SimpleNamespace(**a)


def _scene(dev):
    H, W, V = 44, 60, 2
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], np.float64)
    poses = []
    for v in range(V):
        a = np.radians(0.8 * (v + 1))
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        poses.append(np.concatenate([R, np.array([[0.03 * (v + 1)], [-0.01], [0.0]])], 1))
    vids = [synth.hash_uniform((3, 3, H, W), seed=30 + v, device=dev) for v in range(V)]
    return H, W, K, torch.tensor(np.stack(poses), dtype=torch.float32), torch.tensor(K, dtype=torch.float32)[None].repeat(V, 1, 1), vids


def test_stage1_loop_follows_the_oracle_loop(dev):
    from videoloop3d_amd.MPI import MPMesh
    H, W, K, poses, intrins, vids = _scene(dev)
    args = _args()
    model = MPMesh(args, H, W, np.eye(4), K, 1.0, 100.0).to(dev).train()
    with torch.no_grad():
        model.stack.copy_(synth.make_plane_stack(*model.stack.shape[:4], seed=5) * 0.7)
        model.stack_mask.copy_(synth.hash_uniform(tuple(model.stack_mask.shape), seed=6) * 3 - 2)
    stack_cpu = model.stack.detach().cpu().clone().requires_grad_(True)
    mask_cpu = model.stack_mask.detach().cpu().clone().requires_grad_(True)
    losses = []
    out = drv.train(model, args, vids, poses, intrins, H, W, device=dev, generator=torch.Generator().manual_seed(7),
                    on_step=lambda e, i, loss, img, loop, extra: losses.append(loss))
    assert out["iters"] == 16 and len(losses) == 16
    got = torch.stack(losses).cpu().numpy()

    # the same loop on the oracle: reference loss chain (train_3d.py:196-231) and torch's Adam (MPI.py:122-141)
    ds = drv.MVPatchDataset((H, W), [v.cpu() for v in vids], (24, 32), (20, 28), poses, intrins, "average")
    opt = torch.optim.Adam([stack_cpu, mask_cpu], lr=args.lrate, betas=(0.9, 0.999))
    gen = torch.Generator().manual_seed(7)
    want, step = [], 0
    a2 = _args()
    for epoch in range(2):
        pct = float(np.clip(epoch / 1, 0, 1))
        a2.density_loss_weight = pct * pct * 0.02
        for i in torch.randperm(len(ds), generator=gen).tolist():
            for g in opt.param_groups:
                g["lr"] = args.lrate * 0.1 ** (step / (args.lrate_decay * 1000))
            _, _, pose, intrin, crop, mask = ds[i]
            rgbl, extra = mpv_oracle.mpi_forward(stack_cpu, mask_cpu, a2, H, W, np.eye(4), K, 1.0, 100.0, 24, 32,
                                                 pose2extrin_torch(pose[None]).double(), intrin[None].double())
            lm = torch.clamp(rgbl[:, -1], 0.001, 1 - 0.001)
            loop_loss = -(mask[None] * torch.log(lm) + (1 - mask[None]) * torch.log(1 - lm)).mean()
            rgb = rgbl[:, :3]
            sc = torch.exp(torch.log((crop[None] + 0.01) / (rgb.detach() + 0.01)).mean())
            rgb = rgb * ((sc + 3) / 4)
            loss = ((rgb - crop[None]) ** 2).mean() + loop_loss
            for k, v in extra.items():
                if getattr(a2, f"{k}_loss_weight") > 0:
                    loss = loss + v.mean() * getattr(a2, f"{k}_loss_weight")
            opt.zero_grad()
            loss.backward()
            opt.step()
            want.append(float(loss))
            step += 1
    want = np.array(want)
    assert np.all(np.abs(got - want) <= 2e-3 * np.maximum(1.0, np.abs(want))), np.abs(got - want).max()
    assert want[-4:].mean() < want[:4].mean()                      # (and it trains)
    # parameters after 16 steps
    assert float((model.stack.detach().cpu() - stack_cpu.detach()).abs().max()) <= 5e-3


def test_stage1_loop_sparsifies_and_keeps_training(dev, tmp_path):
    from videoloop3d_amd.MPI import MPMesh
    H, W, K, poses, intrins, vids = _scene(dev)
    args = _args(N_iters=3, sparsify_epoch=1, sparsify_erode=1, sparsify_alpha_thresh=0.05, i_weights=1)
    model = MPMesh(args, H, W, np.eye(4), K, 1.0, 100.0).to(dev).train()
    D, _, Hs, Ws, _ = model.stack.shape
    with torch.no_grad():
        st = synth.make_plane_stack(D, 1, Hs, Ws, seed=5) * 0.7
        yy, xx = torch.meshgrid(torch.arange(Hs).float(), torch.arange(Ws).float(), indexing="ij")
        for d in range(D):
            st[d, 0, :, :, 3] = 6.0 * torch.exp(-(((yy - Hs * (0.3 + 0.1 * d)) / 9) ** 2 + ((xx - Ws * (0.2 + 0.12 * d)) / 12) ** 2)) - 4.5
        model.stack.copy_(st)
    losses = []
    out = drv.train(model, args, vids, poses, intrins, H, W, device=dev, generator=torch.Generator().manual_seed(3), save_dir=str(tmp_path),
                    on_step=lambda e, i, loss, *r: losses.append(loss))
    assert out == {"iters": 24, "epochs": 3, "sparsified_at": 1}
    assert model.is_sparse and 0.05 < float(model.quad_keep.float().mean()) < 0.9
    assert all(bool(torch.isfinite(l)) for l in losses)
    ck = torch.load(str(tmp_path / "epoch_0002.tar"), weights_only=False)
    assert ck["epoch_i"] == 2 and ck["network_state_dict"]["self.is_sparse"] is True


@pytest.mark.parametrize("case", ["mask", "no_mask", "sparse", "few_terms", "no_gain"])
def test_fused_objective_equals_the_generic_spelling(dev, case):
    """MPMesh.objective (vl3d_stage1_objective: the whole scalar head in one sweep) against forward + image_and_loop_loss + weighted_total:
    total, every part, and the gradients w.r.t. the stack and the mask texture -- also under an upstream gradient that is not 1."""
    from videoloop3d_amd.MPI import MPMesh, image_and_loop_loss
    from videoloop3d_amd.train_3dvid import weighted_total
    H, W, K, poses, intrins, vids = _scene(dev)
    kw = {}
    if case in ("no_mask", "sparse"):
        kw["learn_loop_mask"] = False
    if case == "few_terms":
        kw.update(sparsity_loss_weight=0.0, a_smooth_loss_weight=0.0, density_loss_weight=0.0)
    if case == "no_gain":
        kw["scale_invariant"] = False
    args = _args(**kw)
    model = MPMesh(args, H, W, np.eye(4), K, 1.0, 100.0).to(dev).train()
    D, _, Hs, Ws, _ = model.stack.shape
    with torch.no_grad():
        st = synth.make_plane_stack(D, 1, Hs, Ws, seed=5) * 0.7
        if case == "sparse":
            yy, xx = torch.meshgrid(torch.arange(Hs).float(), torch.arange(Ws).float(), indexing="ij")
            for d in range(D):
                st[d, 0, :, :, 3] = 6.0 * torch.exp(-(((yy - Hs * (0.3 + 0.1 * d)) / 9) ** 2 + ((xx - Ws * (0.2 + 0.12 * d)) / 12) ** 2)) - 4.5
        model.stack.copy_(st)
        if model.learn_loop_mask:
            model.stack_mask.copy_(synth.hash_uniform(tuple(model.stack_mask.shape), seed=6) * 3 - 2)
    if case == "sparse":
        model.sparsify_faces(erode_num=1, alpha_thresh=0.05)
    h, w = 24, 32
    ext = pose2extrin_torch(poses[1:2])
    K2 = intrins[1:2].clone()
    K2[:, 0, 2] -= 11
    K2[:, 1, 2] -= 7
    target = synth.hash_uniform((1, 3, h, w), seed=40, device=dev)
    tmask = (synth.hash_uniform((1, h, w), seed=41, device=dev) > 0.5).float()
    names = ("sparsity", "rgb_smooth", "a_smooth", "density")
    wt = lambda k: getattr(args, f"{k}_loss_weight", 0)  # noqa: E731
    params = [model.stack] + ([model.stack_mask] if model.learn_loop_mask else [])

    def generic():
        rgbl, extra = model(h, w, ext, K2)
        img, loop = image_and_loop_loss(rgbl, target, tmask if model.learn_loop_mask else None, scale_invariant=args.scale_invariant)
        loss, _, ex = weighted_total([img] + ([loop] if torch.is_tensor(loop) else []), extra, wt)
        return loss, img, loop, ex

    for up in (1.0, 2.5):
        la, ia, pa, ea = model.objective(h, w, ext, K2, target, tmask if model.learn_loop_mask else None, scale_invariant=args.scale_invariant)
        ga = torch.autograd.grad(la * up, params)
        lb, ib, pb, eb = generic()
        gb = torch.autograd.grad(lb * up, params)
        assert abs(float(la) - float(lb)) <= 2e-6 * max(1.0, abs(float(lb)))
        assert abs(float(ia) - float(ib)) <= 2e-6 * max(1.0, abs(float(ib)))
        if model.learn_loop_mask:
            assert abs(float(pa) - float(pb)) <= 2e-6 * max(1.0, abs(float(pb)))
        else:
            assert pa == 0
        assert sorted(ea) == sorted(eb) == sorted(k for k in names if wt(k) > 0)
        for k in ea:
            assert abs(float(ea[k]) - float(eb[k])) <= 2e-6 * max(1e-3, abs(float(eb[k]))), k
        for x, y in zip(ga, gb):
            assert float((x - y).abs().max()) <= 2e-6 * max(1e-6, float(y.abs().max()))
        assert not la.requires_grad or la.grad_fn is not None
    # the node scales its gradient buffer in place: a second backward through the same forward is refused, not silently wrong
    lc, _, _, _ = model.objective(h, w, ext, K2, target, tmask if model.learn_loop_mask else None, scale_invariant=args.scale_invariant)
    torch.autograd.grad(lc, params, retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        torch.autograd.grad(lc, params)
