"""Crop-aware Adam (videoloop3d_amd/optim.py, csrc/vl3d_optim.hip): deferring and replaying the zero-gradient updates of texels
outside the training crop's window gives torch.optim.Adam's parameters (MPV.py:199-214: betas (0.9, 0.999), eps 6e-8)."""
import types

import numpy as np
import pytest
import torch

from videoloop3d_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import __graft_entry__ as g
    g.build()
    return torch.device("cuda:0")


def test_window_adam_equals_torch_adam_over_a_shuffled_window_schedule(dev):
    """the optimiser alone, on identical gradients: torch.optim.Adam sees the dense gradient (zero outside the step's window),
    WindowAdam the compact one; windows jump around, overlap, leave tiles untouched for many steps, the learning rate changes
    every step (train_3dvid.py:263-277), and one step has no window at all (dense fallback).  After flush(): equal to 2e-6."""
    from videoloop3d_amd.optim import WindowAdam, align_window
    D, T, Hs, Ws = 3, 2, 75, 101
    g = torch.Generator().manual_seed(11)
    p0 = (torch.rand((D, T, Hs, Ws, 4), generator=g) - 0.5).to(dev)
    pa = torch.nn.Parameter(p0.clone())
    pb = torch.nn.Parameter(p0.clone())
    oa = torch.optim.Adam([pa], lr=5e-3, betas=(0.9, 0.999), eps=6e-8)
    ob = WindowAdam([pb], lr=5e-3, betas=(0.9, 0.999), eps=6e-8)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for step in range(40):
        lr = 5e-3 * 0.97 ** step
        for o in (oa, ob):
            o.param_groups[0]["lr"] = lr
        if step == 17:                       # a dense step in the middle (no window leaf: p.grad filled directly)
            G = (torch.rand(pa.shape, generator=g) - 0.5).to(dev)
            pa.grad, pb.grad = G.clone(), G.clone()
            oa.step(); ob.step()
            pa.grad = pb.grad = None
            continue
        if step % 9 == 8:                    # no gradient at all: torch's Adam skips the parameter, so must we
            oa.step(); ob.step()
            continue
        y0, x0 = r(0, Hs - 20), r(0, Ws - 20)
        win = align_window(y0, y0 + r(10, 40), x0, x0 + r(10, 50), Hs, Ws)
        wy, wx, wh, ww = win
        leaf = ob.window_leaf(win)
        assert torch.equal(leaf.detach(), pb.detach()[:, :, wy:wy + wh, wx:wx + ww])          # the leaf holds CURRENT parameters
        gc = (torch.rand((D, T, wh, ww, 4), generator=g) - 0.5).to(dev)
        gc[:, :, :3] = 0                     # texels with a zero gradient inside the window as well
        leaf.grad = gc
        G = torch.zeros_like(pa)
        G[:, :, wy:wy + wh, wx:wx + ww] = gc
        pa.grad = G
        oa.step(); ob.step()
        pa.grad = None
    assert float((pa.detach() - pb.detach()).abs().max()) > 1e-3          # deferred updates are really outstanding ...
    ob.flush()
    assert float((pa.detach() - pb.detach()).abs().max()) <= 2e-6         # ... and replayed exactly
    sa, sb = oa.state[pa], ob.state[pb]
    assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 1e-7 and float((sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max()) <= 1e-7
    assert int(sb["last_step"].min()) == ob.t == int(sa["step"])
    ob.flush()                                                                    # idempotent
    assert float((pa.detach() - pb.detach()).abs().max()) <= 2e-6


def _args(**kw):
    a = dict(mpv_frm_num=4, mpv_isloop=True, mpi_h_scale=1.1, mpi_w_scale=1.1, mpi_d=5, atlas_grid_h=1, init_std=0.3,
             rgb_mlp_type="direct", rgb_activate="sigmoid", alpha_activate="sigmoid", bg_color="", scale_invariant=True, fp16=False,
             swd_patch_size=3, swd_patcht_size=3, swd_stride=2, swd_stridet=1, sparsity_loss_weight=0.0, rgb_smooth_loss_weight=0.2,
             a_smooth_loss_weight=0.2, density_loss_weight=0.0, d_smooth_loss_weight=0.0, optimizer="adam", lrate=5e-3, lrate_decay=30)
    a.update(kw)
    return types.SimpleNamespace(**a)


def test_crop_aware_training_equals_dense_training(dev):
    """the stage-2 iteration of examples/stage2_step.py on two copies of one model: A = dense render + torch.optim.Adam on the dense
    gradient, B = MPMeshVid's crop-aware path (window leaf -> compact gradient -> WindowAdam), crops at shuffled offsets with the
    fused smoothness regularisers on.  Same losses along the way, same parameters at the end (after the flush state_dict() does)."""
    import warnings
    from videoloop3d_amd.MPV import MPMeshVid
    from videoloop3d_amd.optim import WindowAdam
    H, W, h, w = 96, 128, 48, 64
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], np.float64)
    torch.manual_seed(5)
    A = MPMeshVid(_args(), H, W, np.eye(4), K, 1.0, 100.0).to(dev).train()
    B = MPMeshVid(_args(), H, W, np.eye(4), K, 1.0, 100.0).to(dev).train()
    B.load_state_dict({k: v for k, v in A.state_dict().items() if not k.startswith("self.")})
    oa = torch.optim.Adam([A.stack], lr=5e-3, betas=(0.9, 0.999), eps=6e-8)
    ob = B.get_optimizer(0)
    assert isinstance(ob, WindowAdam) and A._window_opt is None
    tar = np.eye(4)
    tar[:3, 3] = [0.03, 0.01, 0.0]
    res = synth.hash_uniform((1, 9, 3, h, w), seed=8, device=dev)
    cfg = dict(loss_name=["gpnn_lm"], loss_gain=torch.tensor([1.0]), macro_block=torch.tensor([65]), patch_size=torch.tensor([3]),
               stride=torch.tensor([2]), patcht_size=torch.tensor([3]), stridet=torch.tensor([1]), alpha=torch.tensor([10000.0]),
               dist_fn=["mse"], rou=["-2"], scaling=torch.tensor([0.1]))
    offs = [(0, 0), (40, 60), (10, 30), (48, 64), (0, 64), (40, 0), (20, 20), (0, 0)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it, (oy, ox) in enumerate(offs):
            Kc = K.copy()
            Kc[0, 2] -= ox
            Kc[1, 2] -= oy
            losses = []
            for model, opt, pose_dev in ((A, oa, dev), (B, ob, "cpu")):      # B takes its poses on the host (no device round trip)
                for grp in opt.param_groups:
                    grp["lr"] = 5e-3 * 0.9 ** it
                opt.zero_grad(set_to_none=True)
                _, extra = model(h, w, torch.tensor(tar, device=pose_dev)[None], torch.tensor(Kc, device=pose_dev)[None], res=res, losscfg=dict(cfg))
                loss = extra["swd"].sum() + 0.2 * extra["rgb_smooth"].sum() + 0.2 * extra["a_smooth"].sum()
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
            assert abs(losses[0] - losses[1]) <= 2e-5 * max(1.0, abs(losses[0])), (it, losses)
    assert B.stack.grad is None                          # the dense gradient was never materialised
    sd = B.state_dict()                                   # flushes the deferred updates
    # The two models see gradients that differ in the last bits (B's texel coordinates are relative to its window), and with
    # eps = 6e-8 Adam's step lr * m^/(sqrt(v^) + eps) of a texel whose gradient is ~1e-8 (a tap at the edge of the crop's footprint)
    # turns those bits into ~1e-4: a handful of texels may differ by that much, the rest agrees tightly.  (The optimiser itself is
    # pinned to 2e-6 on identical gradients by the test above.)
    diff = (sd["stack"] - A.stack.detach()).abs()
    assert float((diff > 2e-5).float().mean()) <= 1e-3 and float(diff.max()) <= 2e-3
    assert float((sd["stack"] - A.stack.detach()).abs().mean()) <= 1e-6
    # evaluation renders read the whole (current) stack
    B.eval()
    A.eval()
    ra, _ = A(H, W, torch.tensor(tar, device=dev)[None], torch.tensor(K, device=dev)[None])
    rb, _ = B(H, W, torch.tensor(tar, device=dev)[None], torch.tensor(K, device=dev)[None])
    assert float((ra - rb).abs().max()) <= 2e-4
