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


@pytest.mark.parametrize("with_boxes,lean", [(False, False), (True, False), (True, True), (False, True)])
def test_window_adam_equals_torch_adam_over_a_shuffled_window_schedule(dev, with_boxes, lean):
    """the optimiser alone, on identical gradients: torch.optim.Adam sees the dense gradient (zero outside the step's window),
    WindowAdam the compact one; windows jump around, overlap, leave tiles untouched for many steps, the learning rate changes
    every step (train_3dvid.py:263-277), and one step has no window at all (dense fallback).  After flush(): equal to 2e-6.
    with_boxes: every plane has its own box inside the window (what a crop's parallax leaves of the union window for one plane): the
    gradient is zero outside it, the leaf shows zeros there, and those texels' updates stay deferred like the rest of the plane.
    lean: the window copy is a view of a persistent buffer and slots outside the boxes are not written (the default)."""
    from videoloop3d_amd.optim import WindowAdam, align_window, tile_side
    D, T, Hs, Ws = 3, 2, 75, 101
    g = torch.Generator().manual_seed(11)
    p0 = (torch.rand((D, T, Hs, Ws, 4), generator=g) - 0.5).to(dev)
    pa = torch.nn.Parameter(p0.clone())
    pb = torch.nn.Parameter(p0.clone())
    oa = torch.optim.Adam([pa], lr=5e-3, betas=(0.9, 0.999), eps=6e-8)
    ob = WindowAdam([pb], lr=5e-3, betas=(0.9, 0.999), eps=6e-8, lean_window=lean)
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
        inside = torch.ones((D, 1, wh, ww, 1), dtype=torch.bool)
        boxes = None
        if with_boxes:
            ts, boxes = tile_side(), []
            for d in range(D):               # a tile-aligned sub-box of the window per plane (one plane gets an empty one now and then)
                by0, by1 = sorted((wy + r(0, wh // ts) * ts, wy + r(0, wh // ts) * ts))
                bx0, bx1 = sorted((wx + r(0, ww // ts) * ts, wx + r(0, ww // ts) * ts))
                by1, bx1 = min(by1, wy + wh), min(bx1, wx + ww)
                boxes.append((by0, by1, bx0, bx1))
                inside[d] = False
                inside[d, :, by0 - wy:by1 - wy, bx0 - wx:bx1 - wx] = True
        inside = inside.to(dev)
        leaf = ob.window_leaf(win, boxes)
        # the leaf holds the CURRENT parameters (what torch's Adam has there now), the stack itself is not written before the step
        cur = pa.detach()[:, :, wy:wy + wh, wx:wx + ww]
        assert float(((leaf.detach() - cur) * inside).abs().max()) <= 2e-6
        if ob.lean_window:      # slots outside a plane's box are left alone (finite leftovers of earlier windows in the persistent buffer)
            assert bool(torch.isfinite(leaf.detach()).all())
        else:
            assert float((leaf.detach() * ~inside).abs().max()) == 0
        gc = (torch.rand((D, T, wh, ww, 4), generator=g) - 0.5).to(dev) * inside
        gc[:, :, :3] = 0                     # texels with a zero gradient inside the window as well
        leaf.grad = gc
        G = torch.zeros_like(pa)
        G[:, :, wy:wy + wh, wx:wx + ww] = gc
        pa.grad = G
        oa.step(); ob.step()
        pa.grad = None
    assert float((pa.detach() - pb.detach()).abs().max()) > 1e-3          # deferred updates are really outstanding ...
    # ... but bounded: the sweep that runs one step in eight leaves no tile more than max_defer steps behind (a dense step at 17 made
    # everything current; 22 more steps followed, so without the bound tiles would sit at step 18)
    ob8 = WindowAdam([torch.nn.Parameter(p0.clone())], lr=5e-3, betas=(0.9, 0.999), eps=6e-8, max_defer=8, lean_window=lean)
    deepest = 0
    for step in range(30):
        leaf = ob8.window_leaf(align_window(0, 16, 0, 16, Hs, Ws))
        leaf.grad = torch.ones_like(leaf)
        ob8.step()
        depth = int((ob8.t - ob8.state[ob8.p]["last_step"]).max())
        assert depth <= 8
        deepest = max(deepest, depth)
    assert deepest >= 4                                                        # (and really deferred in between)
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


def test_window_adam_with_quad_maps_stores_static_texels_once(dev):
    """tile-culled model: culled texels are no parameters, a texel only static quads read is ONE parameter (frame 0; gradient summed
    over the frames inside the step), dynamic texels one per frame -- equal to torch.optim.Adam on the TIED dense gradient
    (tiles.tie_static_grad, the definition), windows shuffled, after flush() also in the mirrored frames."""
    from videoloop3d_amd import tiles
    from videoloop3d_amd.optim import WindowAdam, align_window
    D, T, Hs, Ws, QH, QW = 3, 3, 75, 101, 6, 8
    g = torch.Generator().manual_seed(21)
    keep = torch.rand((D, QH, QW), generator=g) < 0.6
    dyn = keep & (torch.rand((D, QH, QW), generator=g) < 0.5)
    keep_t = tiles.quad_to_texel_mask(keep, Hs, Ws)
    static_t = keep_t & ~tiles.quad_to_texel_mask(dyn, Hs, Ws)
    p0 = (torch.rand((D, T, Hs, Ws, 4), generator=g) - 0.5)
    p0 = torch.where(static_t[:, None, :, :, None], p0[:, :1], p0)               # static texels start as one texture
    tiles.cull_stack_(p0, keep)
    pa = torch.nn.Parameter(p0.clone().to(dev))
    pb = torch.nn.Parameter(p0.clone().to(dev))
    oa = torch.optim.Adam([pa], lr=5e-3, betas=(0.9, 0.999), eps=6e-8)
    ob = WindowAdam([pb], lr=5e-3, betas=(0.9, 0.999), eps=6e-8, quad_keep=keep.to(dev), quad_dyn=dyn.to(dev), culled_alpha=tiles.CULLED_ALPHA,
                    lean_window=False)
    # the default, lean window copy (a persistent buffer; culled texels are left alone) of a second optimiser over the same parameters:
    # what the render can read must be identical, what it cannot must be finite
    ol = WindowAdam([torch.nn.Parameter(p0.clone().to(dev))], lr=5e-3, betas=(0.9, 0.999), eps=6e-8, quad_keep=keep.to(dev), quad_dyn=dyn.to(dev),
                    culled_alpha=tiles.CULLED_ALPHA)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for step in range(25):
        for o in (oa, ob):
            o.param_groups[0]["lr"] = 5e-3 * 0.97 ** step
        y0, x0 = r(0, Hs - 20), r(0, Ws - 20)
        wy, wx, wh, ww = align_window(y0, y0 + r(10, 40), x0, x0 + r(10, 50), Hs, Ws)
        leaf = ob.window_leaf((wy, wx, wh, ww))
        kt = keep_t[:, wy:wy + wh, wx:wx + ww].to(dev)
        stt = static_t[:, wy:wy + wh, wx:wx + ww].to(dev)
        assert bool((leaf.detach()[..., 3][~kt[:, None].expand(D, T, wh, ww)] == tiles.CULLED_ALPHA).all())      # culled: transparent
        assert torch.equal(leaf.detach()[:, 1][stt], leaf.detach()[:, 0][stt])                                      # static: one texture
        gc = (torch.rand((D, T, wh, ww, 4), generator=g) - 0.5).to(dev) * kt[:, None, :, :, None]                  # the culled render leaves 0 there
        leaf.grad = gc
        ol.param_groups[0]["lr"] = 5e-3 * 0.97 ** step
        lean = ol.window_leaf((wy, wx, wh, ww))
        ktx = kt[:, None, :, :, None].expand(D, T, wh, ww, 4)
        assert torch.equal(lean.detach()[ktx], leaf.detach()[ktx]) and bool(torch.isfinite(lean.detach()).all())
        lean.grad = gc.clone()
        ol.step()
        G = torch.zeros_like(pa)
        G[:, :, wy:wy + wh, wx:wx + ww] = gc
        pa.grad = tiles.tie_static_grad(G, keep.to(dev), dyn.to(dev))
        oa.step(); ob.step()
        pa.grad = None
    ob.flush()
    kept = keep_t[:, None, :, :, None].expand_as(pa).to(dev)
    assert float((pa.detach() - pb.detach())[kept].abs().max()) <= 2e-6
    assert torch.equal(pb.detach()[~kept], p0.to(dev)[~kept])                    # culled texels were never written
    ol.flush()
    assert torch.equal(ol.p.detach(), pb.detach())                                # lean and full window copies: the same training


def test_culled_render_from_a_window_of_the_stack(dev):
    """vl3d_render_*_culled with desc->cull_* : the stack is a texel window of the plane the quad grid lies over -- same image and
    gradient (on the window) as the culled render of the whole stack."""
    from videoloop3d_amd.render import RenderSpec, render_planes_with_regularisers
    import dataclasses
    D, T, Hs, Ws, H, W, QH, QW = 5, 2, 150, 200, 60, 90, 7, 9
    torch.manual_seed(4)
    keep = (torch.rand(D, QH, QW) < 0.5).to(dev)
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=13, device=dev)
    from test_gpu_render import bench_homos
    homos = (torch.tensor([[1.0, 0, 50.0], [0, 1.0, 40.0], [0, 0, 1.0]]) @ bench_homos(D, H, W, scale=1.5)).to(dev)
    spec = RenderSpec.mpv()
    full = stack.clone().requires_grad_(True)
    out_f = render_planes_with_regularisers(full, homos, H, W, spec, quad_keep=keep)
    g = synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5
    obj = lambda o: (o[0] * g).sum() + 1e-4 * o[2].sum() + 1e-3 * o[3].sum()
    (g_f,) = torch.autograd.grad(obj(out_f), full)
    y0, x0, wh, ww = 32, 32, 96, 144
    assert float(g_f.abs().sum()) == pytest.approx(float(g_f[:, :, y0:y0 + wh, x0:x0 + ww].abs().sum()))      # the window holds the whole footprint
    win = stack[:, :, y0:y0 + wh, x0:x0 + ww].contiguous().requires_grad_(True)
    out_w = render_planes_with_regularisers(win, homos, H, W, dataclasses.replace(spec, offset=(-float(x0), -float(y0))), quad_keep=keep,
                                            cull_window=(y0, x0, Hs, Ws))
    (g_w,) = torch.autograd.grad(obj(out_w), win)
    assert float((out_w[0] - out_f[0]).abs().max()) <= 1e-5 and float((out_w[1] - out_f[1]).abs().max()) <= 1e-5
    assert float(((out_w[2] - out_f[2]).abs() / out_f[2].abs().clamp_min(1.0)).max()) <= 1e-5
    assert float((g_w - g_f[:, :, y0:y0 + wh, x0:x0 + ww]).abs().max()) <= 1e-5 * max(1.0, float(g_f.abs().max()))
    assert float(out_f[0].abs().max()) > 0.01
    # VL3D_GRAD_CULLED_UNWRITTEN (what the window path of a sparsified model asks for): texels no kept quad can read are left undefined,
    # every texel that is a parameter gets the same gradient bit for bit
    from videoloop3d_amd import tiles
    win2 = win.detach().clone().requires_grad_(True)
    out_u = render_planes_with_regularisers(win2, homos, H, W, dataclasses.replace(spec, offset=(-float(x0), -float(y0))), quad_keep=keep,
                                            cull_window=(y0, x0, Hs, Ws), grad_culled_unwritten=True)
    (g_u,) = torch.autograd.grad(obj(out_u), win2)
    kt = tiles.quad_to_texel_mask(keep.cpu(), Hs, Ws)[:, y0:y0 + wh, x0:x0 + ww].to(dev)[:, None, :, :, None].expand_as(g_w)
    assert torch.equal(g_u[kt], g_w[kt]) and float(g_w[~kt].abs().max()) == 0.0


def test_unwritten_culled_gradient_on_a_texel_aligned_view(dev):
    """VL3D_GRAD_CULLED_UNWRITTEN rests on the kernels' float box test (a tile / plane pair is skipped when its texel box touches no kept
    quad) agreeing with the integer texel classification the optimiser uses (tiles.quad_to_texel_mask == vl3d texel_class) -- hardest where
    samples sit exactly ON texel centres and quad borders: a view that is an integer texel shift of every plane, quad borders on integer
    texels.  Every texel a kept quad can read must be written, with the zero-filling kernel's bits."""
    import dataclasses
    from videoloop3d_amd import tiles
    from videoloop3d_amd.render import RenderSpec, render_planes_with_regularisers
    D, T, QH, QW, H, W = 6, 2, 7, 9, 64, 96
    Hs, Ws = 20 * QH + 1, 22 * QW + 1                       # quad borders on texels 0, 20, 40, ... / 0, 22, 44, ...
    y0, x0, wh, ww = 32, 40, 96, 136
    g = synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5
    for seed in range(4):
        torch.manual_seed(seed)
        keep = (torch.rand(D, QH, QW) < (0.15 if seed == 3 else 0.5)).to(dev)
        win = synth.make_plane_stack(D, T, wh, ww, seed=13 + seed, device=dev)
        # plane d: texel = pixel + (41 + 3d, 33 + 2d) exactly (pixel centre 0.5 folded into the offset), inside the window
        homos = torch.eye(3)[None].repeat(D, 1, 1)
        homos[:, 0, 2] = 41.0 + 3 * torch.arange(D)
        homos[:, 1, 2] = 33.0 + 2 * torch.arange(D)
        spec = dataclasses.replace(RenderSpec.mpv(), offset=(-0.5 - x0, -0.5 - y0))
        obj = lambda o: (o[0] * g).sum() + 1e-4 * o[2].sum() + 1e-3 * o[3].sum()      # noqa: E731
        grads = []
        for lean in (False, True):
            leaf = win.clone().requires_grad_(True)
            out = render_planes_with_regularisers(leaf, homos.to(dev), H, W, spec, quad_keep=keep, cull_window=(y0, x0, Hs, Ws),
                                                  grad_culled_unwritten=lean)
            grads.append(torch.autograd.grad(obj(out), leaf)[0])
        kt = tiles.quad_to_texel_mask(keep.cpu(), Hs, Ws)[:, y0:y0 + wh, x0:x0 + ww].to(dev)[:, None, :, :, None].expand_as(grads[0])
        assert float(grads[0][kt].abs().max()) > 0 and float(grads[0][~kt].abs().max()) == 0.0
        assert torch.equal(grads[1][kt], grads[0][kt]), seed


@pytest.mark.parametrize("variant", [0, 3, 5])
def test_unwritten_culled_gradient_without_regularisers(dev, variant):
    """VL3D_GRAD_CULLED_UNWRITTEN on the plain culled backward (no layer regularisers): the call takes the instantiation that SKIPS the planes a
    tile cannot see (64-wide regions; desc->variant 5: 32-wide) -- every texel a kept quad can read has the zero-filling kernel's bits."""
    from videoloop3d_amd import tiles
    from videoloop3d_amd.render import RenderSpec, render_planes
    from test_gpu_render import bench_homos
    D, T, Hs, Ws, H, W, QH, QW = 7, 3, 170, 230, 150, 200, 9, 12
    torch.manual_seed(11)
    keep = (torch.rand(D, QH, QW) < 0.3).to(dev)
    keep[2] = False
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=17, device=dev)
    with torch.no_grad():
        tiles.cull_stack_(stack, keep)
    homos = (torch.tensor([[1.0, 0, 14.0], [0, 1.0, 9.0], [0, 0, 1.0]]) @ bench_homos(D, H, W)).to(dev)
    g = synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5
    grads = []
    for lean, v in ((False, 0), (True, variant)):
        leaf = stack.clone().requires_grad_(True)
        rgb, _ = render_planes(leaf, homos, H, W, RenderSpec.mpv(variant=v), quad_keep=keep, grad_culled_unwritten=lean)
        grads.append(torch.autograd.grad(rgb, leaf, g)[0])
    kt = tiles.quad_to_texel_mask(keep.cpu(), Hs, Ws).to(dev)[:, None, :, :, None].expand_as(grads[0])
    assert float(grads[0][kt].abs().max()) > 1e-4 and float(grads[0][~kt].abs().max()) == 0.0
    assert torch.equal(grads[1][kt], grads[0][kt])


def test_sparsified_model_trains_through_the_window_path(dev):
    """MPMeshVid of a sparsified MPI: the crop-aware path (WindowAdam with the quad maps, static texels stored once) against the
    round-1 path (TileAdam over the whole stack + tie hook, args.tile_adam): same losses, same kept texels after the flush."""
    import warnings
    from videoloop3d_amd import tiles
    from videoloop3d_amd.MPV import MPMeshVid
    from videoloop3d_amd.optim import WindowAdam
    H, W, h, w = 96, 128, 48, 64
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], np.float64)
    torch.manual_seed(6)
    keep = torch.rand(5, 8, 11) < 0.6
    dyn = keep & (torch.rand(5, 8, 11) < 0.5)
    models = []
    for tile_adam in (True, False):
        torch.manual_seed(7)
        m = MPMeshVid(_args(mpi_h_verts=9, mpi_w_verts=12, tile_adam=tile_adam), H, W, np.eye(4), K, 1.0, 100.0)
        with torch.no_grad():
            st = tiles.quad_to_texel_mask(keep, m.mpi_h, m.mpi_w) & ~tiles.quad_to_texel_mask(dyn, m.mpi_h, m.mpi_w)
            m.stack.data = torch.where(st[:, None, :, :, None], m.stack.data[:, :1], m.stack.data)
            tiles.cull_stack_(m.stack.data, keep)
        m.register_buffer("quad_keep", keep.clone())
        m.register_buffer("quad_dyn", dyn.clone())
        m.is_sparse = m.has_dyn = True
        m = m.to(dev).train()
        m._install_tie_hook()
        models.append((m, m.get_optimizer(0)))
    assert isinstance(models[1][1], WindowAdam) and not isinstance(models[0][1], WindowAdam)
    tar = np.eye(4)
    tar[:3, 3] = [0.03, 0.01, 0.0]
    res = synth.hash_uniform((1, 9, 3, h, w), seed=8, device=dev)
    cfg = dict(loss_name=["gpnn_lm"], loss_gain=torch.tensor([1.0]), macro_block=torch.tensor([65]), patch_size=torch.tensor([3]),
               stride=torch.tensor([2]), patcht_size=torch.tensor([3]), stridet=torch.tensor([1]), alpha=torch.tensor([10000.0]),
               dist_fn=["mse"], rou=["-2"], scaling=torch.tensor([0.1]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it, (oy, ox) in enumerate([(0, 0), (40, 60), (10, 30), (48, 64), (0, 64), (40, 0)]):
            Kc = K.copy()
            Kc[0, 2] -= ox
            Kc[1, 2] -= oy
            losses = []
            for model, opt in models:
                opt.zero_grad(set_to_none=True)
                _, extra = model(h, w, torch.tensor(tar)[None], torch.tensor(Kc)[None], res=res, losscfg=dict(cfg))
                loss = extra["swd"].sum() + 0.2 * extra["rgb_smooth"].sum() + 0.2 * extra["a_smooth"].sum()
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
            assert abs(losses[0] - losses[1]) <= 2e-5 * max(1.0, abs(losses[0])), (it, losses)
    sa, sb = models[0][0].state_dict()["stack"], models[1][0].state_dict()["stack"]
    kept = tiles.quad_to_texel_mask(keep, *sa.shape[2:4]).to(dev)[:, None, :, :, None].expand_as(sa)
    diff = (sa - sb)[kept].abs()
    assert float((diff > 2e-5).float().mean()) <= 1e-3 and float(diff.max()) <= 2e-3 and float(diff.mean()) <= 1e-6


def test_window_adam_state_dict_round_trip(dev):
    """state_dict() flushes and saves torch.optim.Adam's layout (exp_avg, exp_avg_sq, step); a fresh WindowAdam that loads it continues
    exactly like the original (the deferral bookkeeping restarts at `step`)."""
    from videoloop3d_amd.optim import WindowAdam, align_window
    torch.manual_seed(3)
    D, T, Hs, Ws = 3, 2, 40, 56
    pa = torch.nn.Parameter(torch.randn(D, T, Hs, Ws, 4, device=dev) * 0.1)
    oa = WindowAdam([pa], lr=1e-2, betas=(0.9, 0.999), eps=6e-8)
    wins = [align_window(0, 17, 8, 30, Hs, Ws), align_window(16, 40, 24, 56, Hs, Ws), align_window(8, 24, 0, 24, Hs, Ws)]

    def steps(opt, p, k0, k1):
        for k in range(k0, k1):
            w = wins[k % len(wins)]
            leaf = opt.window_leaf(w)
            g = torch.Generator(device=dev).manual_seed(100 + k)
            leaf.grad = torch.randn(leaf.shape, device=dev, generator=g)
            opt.step()
    steps(oa, pa, 0, 5)
    import io
    buf = io.BytesIO()
    torch.save(oa.state_dict(), buf)          # through a file image, like a checkpoint (load_state_dict aliases same-device tensors)
    buf.seek(0)
    sd = torch.load(buf)
    assert set(next(iter(sd["state"].values())).keys()) == {"exp_avg", "exp_avg_sq", "step"}
    pb = torch.nn.Parameter(pa.detach().clone())
    ob = WindowAdam([pb], lr=1e-2, betas=(0.9, 0.999), eps=6e-8)
    ob.load_state_dict(sd)
    assert ob.t == 5
    steps(oa, pa, 5, 9)
    steps(ob, pb, 5, 9)
    oa.flush(); ob.flush()
    assert torch.equal(pa.detach(), pb.detach())
    assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"])


@pytest.mark.parametrize("smooth,T,scale,rot,variant", [(0.2, 4, 1.1, 0.0, 0), (0.0, 5, 1.6, 0.0, 0), (0.2, 5, 1.25, 0.0, 0), (0.2, 4, 1.1, 40.0, 0),
                                                        (0.0, 4, 1.6, 0.0, 0)])
def test_step_fused_into_the_backward_equals_backward_then_step(dev, smooth, T, scale, rot, variant):
    """WindowAdam(fused_backward=True) -- vl3d_render_bwd_adam: the owner-computes backward applies the optimiser's step where it would have
    stored a texel's gradient -- against the two-kernel path (vl3d_render_bwd, then vl3d_adam_window_step_boxes) on two copies of one
    model over the same shuffled crops: the SAME BITS in the parameters, both moments and the step table after every iteration (texels
    owned by a tile, texels only the pre-pass reaches, texels outside their plane's box, deferred tiles coming back), with and without the
    layer regularisers, even and odd frame counts, stacks above the frame's resolution (windows larger than a workgroup), and a view the
    device-side plan refuses (rot 40 degrees about the optical axis: the atomics kernel + the step kernel run instead, decided on the
    device) -- there the gradient comes from atomics, so only the loss is compared exactly and the parameters to the atomics' noise."""
    import warnings
    import videoloop3d_amd.render as R
    from videoloop3d_amd.MPV import MPMeshVid
    H, W, h, w = 96, 128, 48, 64
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], np.float64)
    kw = dict(mpv_frm_num=T, mpi_h_scale=scale, mpi_w_scale=scale, rgb_smooth_loss_weight=smooth, a_smooth_loss_weight=smooth,
              sparsity_loss_weight=0.01 if smooth else 0.0)
    torch.manual_seed(5)
    A = MPMeshVid(_args(fused_adam_backward=False, **kw), H, W, np.eye(4), K, 1.0, 100.0).to(dev).train()
    B = MPMeshVid(_args(**kw), H, W, np.eye(4), K, 1.0, 100.0).to(dev).train()      # (the default for dense models)
    B.load_state_dict({k: v for k, v in A.state_dict().items() if not k.startswith("self.")})
    oa, ob = A.get_optimizer(0), B.get_optimizer(0)
    assert ob.fused_backward and not oa.fused_backward
    tar = np.eye(4)
    c, s_ = np.cos(np.radians(rot)), np.sin(np.radians(rot))
    tar[:3, :3] = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])
    tar[:3, 3] = [0.03, 0.01, 0.0]
    res = synth.hash_uniform((1, 2 * T + 1, 3, h, w), seed=8, device=dev)
    cfg = dict(loss_name=["gpnn_lm"], loss_gain=torch.tensor([1.0]), macro_block=torch.tensor([65]), patch_size=torch.tensor([3]),
               stride=torch.tensor([2]), patcht_size=torch.tensor([3]), stridet=torch.tensor([1]), alpha=torch.tensor([10000.0]),
               dist_fn=["mse"], rou=["-2"], scaling=torch.tensor([0.1]))
    offs = [(0, 0), (40, 60), (10, 30), (48, 64), (0, 64), (40, 0), (20, 20), (0, 0), (48, 64), (10, 30)]
    if (smooth, T, scale, rot, variant) == (0.2, 4, 1.1, 0.0, 0):
        offs = offs * 4 + offs[:2]      # 42 steps: past the deferral bound (the sweeps of vl3d_adam_flush_older at steps 24, 32, 40 run between fused steps)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it, (oy, ox) in enumerate(offs):
            Kc = K.copy()
            Kc[0, 2] -= ox
            Kc[1, 2] -= oy
            losses, feasible = [], []
            for model, opt in ((A, oa), (B, ob)):
                for grp in opt.param_groups:
                    grp["lr"] = 5e-3 * 0.9 ** it
                opt.zero_grad(set_to_none=True)
                _, extra = model(h, w, torch.tensor(tar)[None], torch.tensor(Kc)[None], res=res, losscfg=dict(cfg))
                loss = extra["swd"].sum()
                if smooth:
                    loss = loss + 0.2 * extra["rgb_smooth"].sum() + 0.2 * extra["a_smooth"].sum() + 0.01 * extra["sparsity"].sum()
                loss.backward()
                feasible.append(int(R.LAST_BWD_SCRATCH[:1].view(torch.int32)))
                opt.step()
                losses.append(float(loss.detach()))
            assert feasible[0] == feasible[1] == (0 if rot else 1)
            sa, sb = oa.state[oa.p], ob.state[ob.p]
            assert torch.equal(sa["last_step"], sb["last_step"]) and oa.t == ob.t == it + 1
            if rot:
                assert abs(losses[0] - losses[1]) <= 1e-5 * max(1.0, abs(losses[0]))
                continue
            assert losses[0] == losses[1], (it, losses)
            for name, x, y in (("p", A.stack.data, B.stack.data), ("m", sa["exp_avg"], sb["exp_avg"]), ("v", sa["exp_avg_sq"], sb["exp_avg_sq"])):
                assert torch.equal(x, y), (it, name, float((x - y).abs().max()), int((x != y).sum()))
    assert ob.fused_steps == len(offs) and oa.fused_steps == 0
    assert B.stack.grad is None
    sda, sdb = A.state_dict(), B.state_dict()           # flushes the deferred updates
    if rot:
        assert float((sda["stack"] - sdb["stack"]).abs().mean()) <= 1e-6
    else:
        assert torch.equal(sda["stack"], sdb["stack"])


@pytest.mark.parametrize("variant", [0, 3, 5])      # 0: the default region shape, 3: 64-wide regions, 5: 32-wide regions (two workgroups per CU)
@pytest.mark.parametrize("smooth,T,scale,rot,packed", [(0.2, 4, 1.1, 0.0, False), (0.0, 5, 1.6, 0.0, False), (0.2, 3, 1.25, 40.0, False),
                                                       (0.2, 4, 1.1, 0.0, True), (0.0, 5, 1.25, 0.0, True), (0.2, 3, 1.1, 40.0, True)])
def test_tile_culled_step_fused_into_the_backward(dev, smooth, T, scale, rot, packed, variant):
    """the same for a TILE-CULLED model (vl3d_render_bwd_adam with quad maps): dynamic texels are stepped in the owner's store, a static
    texel's gradient is stored and summed over the frames by the step kernel behind the backward (static texels only), culled texels are
    no parameters -- against vl3d_render_bwd_culled + vl3d_adam_window_step_boxes over all classes: the same bits in p, m, v and the step
    table after every iteration (kept texels; culled slots hold whatever they held).  rot: the infeasible view (atomics + full step kernel).
    packed: parameters and moments in the pools of 8 x 8-texel blocks (block table addressing in the owner's store)."""
    import warnings
    import videoloop3d_amd.render as R
    from videoloop3d_amd import tiles
    from videoloop3d_amd.MPV import MPMeshVid
    H, W, h, w = 96, 128, 48, 64
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], np.float64)
    torch.manual_seed(6)
    keep = torch.rand(5, 8, 11) < 0.6
    keep[3] = False                                # a plane without a single kept quad
    dyn = keep & (torch.rand(5, 8, 11) < 0.5)
    kw = dict(mpv_frm_num=T, mpi_h_scale=scale, mpi_w_scale=scale, rgb_smooth_loss_weight=smooth, a_smooth_loss_weight=smooth,
              sparsity_loss_weight=0.01 if smooth else 0.0, mpi_h_verts=9, mpi_w_verts=12)
    models = []
    for fused in (False, True):
        torch.manual_seed(7)
        m = MPMeshVid(_args(fused_adam_backward=fused, **kw), H, W, np.eye(4), K, 1.0, 100.0)
        with torch.no_grad():
            st = tiles.quad_to_texel_mask(keep, m.mpi_h, m.mpi_w) & ~tiles.quad_to_texel_mask(dyn, m.mpi_h, m.mpi_w)
            m.stack.data = torch.where(st[:, None, :, :, None], m.stack.data[:, :1], m.stack.data)
            tiles.cull_stack_(m.stack.data, keep)
        m.register_buffer("quad_keep", keep.clone())
        m.register_buffer("quad_dyn", dyn.clone())
        m.is_sparse = m.has_dyn = True
        m = m.to(dev).train()
        m._install_tie_hook()
        if packed:
            m.pack_()
        if fused and variant:
            import dataclasses
            m.spec = dataclasses.replace(m.spec, variant=variant)
        models.append((m, m.get_optimizer(0)))
    (A, oa), (B, ob) = models
    assert ob.fused_backward and not oa.fused_backward and ob.quad_keep is not None and (ob.layout is not None) == packed
    tar = np.eye(4)
    c, s_ = np.cos(np.radians(rot)), np.sin(np.radians(rot))
    tar[:3, :3] = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])
    tar[:3, 3] = [0.03, 0.01, 0.0]
    res = synth.hash_uniform((1, 2 * T + 1, 3, h, w), seed=8, device=dev)
    cfg = dict(loss_name=["gpnn_lm"], loss_gain=torch.tensor([1.0]), macro_block=torch.tensor([65]), patch_size=torch.tensor([3]),
               stride=torch.tensor([2]), patcht_size=torch.tensor([3]), stridet=torch.tensor([1]), alpha=torch.tensor([10000.0]),
               dist_fn=["mse"], rou=["-2"], scaling=torch.tensor([0.1]))
    Dd, Td, Hs_, Ws_ = A.stack_dims()
    kept = tiles.quad_to_texel_mask(keep, Hs_, Ws_).to(dev)[:, None, :, :, None].expand(Dd, Td, Hs_, Ws_, 4)
    static = (tiles.quad_to_texel_mask(keep, Hs_, Ws_) & ~tiles.quad_to_texel_mask(dyn, Hs_, Ws_)).to(dev)
    offs = [(0, 0), (40, 60), (10, 30), (48, 64), (0, 64), (40, 0), (20, 20), (0, 0), (48, 64), (10, 30)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it, (oy, ox) in enumerate(offs):
            Kc = K.copy()
            Kc[0, 2] -= ox
            Kc[1, 2] -= oy
            losses, feasible = [], []
            for model, opt in models:
                for grp in opt.param_groups:
                    grp["lr"] = 5e-3 * 0.9 ** it
                opt.zero_grad(set_to_none=True)
                _, extra = model(h, w, torch.tensor(tar)[None], torch.tensor(Kc)[None], res=res, losscfg=dict(cfg))
                loss = extra["swd"].sum()
                if smooth:
                    loss = loss + 0.2 * extra["rgb_smooth"].sum() + 0.2 * extra["a_smooth"].sum() + 0.01 * extra["sparsity"].sum()
                loss.backward()
                feasible.append(int(R.LAST_BWD_SCRATCH[:1].view(torch.int32)))
                opt.step()
                losses.append(float(loss.detach()))
            assert feasible[0] == feasible[1] == (0 if rot else 1)
            sa, sb = oa.state[oa.p], ob.state[ob.p]
            assert torch.equal(sa["last_step"], sb["last_step"]) and oa.t == ob.t == it + 1
            if rot:
                assert abs(losses[0] - losses[1]) <= 1e-5 * max(1.0, abs(losses[0]))
                continue
            assert losses[0] == losses[1], (it, losses)
            for name, x, y in (("p", oa.p.data, ob.p.data), ("m", sa["exp_avg"], sb["exp_avg"]), ("v", sa["exp_avg_sq"], sb["exp_avg_sq"])):
                if packed:       # the pools: every slot belongs to a kept block
                    assert torch.equal(x, y), (it, name, float((x - y).abs().max()), int((x != y).sum()))
                    continue
                # a static texel is ONE parameter living in frame 0 (the other frames' slots are refreshed by flush())
                x = torch.where(static[:, None, :, :, None], x[:, :1].expand_as(x), x)[kept]
                y = torch.where(static[:, None, :, :, None], y[:, :1].expand_as(y), y)[kept]
                assert torch.equal(x, y), (it, name, float((x - y).abs().max()), int((x != y).sum()))
    assert ob.fused_steps == len(offs) and oa.fused_steps == 0
    if packed:
        oa.flush(); ob.flush()
        sda, sdb, kept = oa.p.data, ob.p.data, slice(None)
    else:
        sda, sdb = A.state_dict()["stack"], B.state_dict()["stack"]           # flushes the deferred updates
    if rot:
        assert float((sda - sdb)[kept].abs().mean()) <= 1e-6
    else:
        assert torch.equal(sda[kept], sdb[kept])


def test_render_bwd_adam_refuses_what_it_is_not_built_for(dev):
    """vl3d_render_bwd_adam launches nothing for descriptors outside its scope (include/vl3d.h): other conventions / activations / fp16 stacks,
    a dense model with one frame, a window that is not aligned to the bookkeeping tiles, quad maps without the class scratch, a packed
    layout without quad maps -- VL3D_EUNSUPPORTED / VL3D_EINVAL with a message, parameters untouched."""
    import ctypes as C
    from videoloop3d_amd import _lib as L
    from videoloop3d_amd.render import RenderSpec, _desc
    D, T, Hs, Ws, wh, ww, H, W = 2, 2, 32, 40, 16, 24, 8, 12
    z = lambda *sh, dt=torch.float32: torch.zeros(sh, dtype=dt, device=dev)      # noqa: E731
    stack, homos = z(D, T, wh, ww, 4), torch.eye(3, device=dev).repeat(D, 1, 1).contiguous()
    p, m, v = torch.ones((D, T, Hs, Ws, 4), device=dev), z(D, T, Hs, Ws, 4), z(D, T, Hs, Ws, 4)
    last, hist = z(D, 4, 5, dt=torch.int32), z(16, 2)
    rgb, alpha, g_rgb, g = z(T, H, W, 3), z(T, H, W), z(T, H, W, 3), z(D, T, wh, ww, 4)

    def call(spec=RenderSpec.mpv(), stack=stack, y0=8, **over):
        desc = _desc(stack, H, W, spec, 0, 0)
        aw = L.AdamWindow()
        aw.Hs, aw.Ws, aw.y0, aw.x0 = Hs, Ws, y0, 8
        aw.param, aw.exp_avg, aw.exp_avg_sq, aw.last_step, aw.hist = (t.data_ptr() for t in (p, m, v, last, hist))
        aw.lr, aw.beta1, aw.beta2, aw.eps, aw.step = 1e-3, 0.9, 0.999, 1e-8, 1
        for k, val in over.items():
            setattr(aw, k, val)
        n = int(L.lib().vl3d_render_bwd_scratch_bytes(desc))
        scratch = z((n + 3) // 4)
        rc = L.lib().vl3d_render_bwd_adam(desc, L.ptr(stack), L.ptr(homos), L.ptr(rgb), L.ptr(alpha), L.ptr(g_rgb), None, None, None, None, L.ptr(g),
                                          L.ptr(scratch), n, C.byref(aw), L.stream_ptr(dev))
        return rc, (L.lib().vl3d_last_error() or b"").decode()
    qk = torch.ones((D, 3, 4), dtype=torch.uint8, device=dev)
    blocks = z(D, 4, 5, dt=torch.int32)
    cases = [dict(spec=RenderSpec()), dict(spec=RenderSpec.mpv(rgb_act="none")), dict(stack=stack.half()), dict(stack=stack[:, :1].contiguous()),
             dict(spec=RenderSpec.mpv(variant=1)), dict(y0=4), dict(quad_keep=qk.data_ptr(), QH=3, QW=4), dict(blocks=blocks.data_ptr())]
    for kw in cases:
        rc, msg = call(**kw)
        assert rc != 0 and msg, (kw, rc, msg)
    torch.cuda.synchronize()
    assert bool((p == 1).all()) and bool((m == 0).all()) and int(last.abs().max()) == 0
    rc, msg = call()                                  # ... and the plain call goes through (zero upstream gradient: a zero-gradient step of the window)
    assert rc == 0, msg
    torch.cuda.synchronize()
    assert int(last[:, 1:3, 1:4].min()) == 1 and int(last.sum()) == D * 2 * 3


@pytest.mark.parametrize("sparse", [False, True])
def test_fused_step_at_the_schedule_s_shapes(dev, sparse):
    """the same bit-for-bit property at the shapes examples/stage2_schedule.py trains (configs/mpv_base.txt: 360 x 640 frames, 180 x 320 crops,
    D = 32 planes at 1.1x, 36 x 64 vertices; 6 frames here to bound the memory), two pyramid levels (lod 0.75, then 1.0: new optimisers), three
    poses, per-plane boxes, regularisers on: fused step == backward + step kernel in p, m, v and the step table after every iteration."""
    import warnings
    from videoloop3d_amd import tiles
    from videoloop3d_amd.MPV import MPMeshVid
    H, W, h, w, T, D = 360, 640, 180, 320, 6, 32
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], np.float64)
    kw = dict(mpv_frm_num=T, mpi_d=D, atlas_grid_h=4, init_std=0.02, mpi_h_verts=36, mpi_w_verts=64)
    models = []
    for fused in (False, True):
        torch.manual_seed(7)
        m = MPMeshVid(_args(fused_adam_backward=fused, **kw), H, W, np.eye(4), K, 1.0, 100.0).to(dev).train()
        if sparse:      # the quad maps of examples/stage2_schedule.py: a blob per plane, every other kept quad dynamic
            QH, QW = 35, 63
            qy, qx = torch.meshgrid(torch.arange(QH, device=dev), torch.arange(QW, device=dev), indexing="ij")
            keep = torch.zeros((D, QH, QW), dtype=torch.bool, device=dev)
            for d in range(D):
                cy, cx = (7 * d + 3) % QH, (11 * d + 5) % QW
                keep[d] = ((qy - cy).abs() <= QH // 5) & ((qx - cx).abs() <= QW // 4)
            m.register_buffer("quad_keep", keep)
            m.register_buffer("quad_dyn", keep & ((qy + qx) % 2 == 0)[None])
            m.is_sparse = m.has_dyn = True
            with torch.no_grad():
                tiles.cull_stack_(m.stack.data, keep)
            m._install_tie_hook()
        models.append(m)
    A, B = models
    res = synth.hash_uniform((1, 2 * T + 1, 3, h, w), seed=8, device=dev)
    cfg = dict(loss_name=["gpnn_lm"], loss_gain=torch.tensor([1.0]), macro_block=torch.tensor([65]), patch_size=torch.tensor([3]),
               stride=torch.tensor([2]), patcht_size=torch.tensor([3]), stridet=torch.tensor([1]), alpha=torch.tensor([10000.0]),
               dist_fn=["mse"], rou=["-2"], scaling=torch.tensor([0.1]))
    poses = []
    for v in range(3):
        a_, b_ = np.radians(1.2 * np.cos(2 * np.pi * v / 3)), np.radians(0.8 * np.sin(2 * np.pi * v / 3))
        Ry = np.array([[np.cos(a_), 0, np.sin(a_)], [0, 1, 0], [-np.sin(a_), 0, np.cos(a_)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(b_), -np.sin(b_)], [0, np.sin(b_), np.cos(b_)]])
        E = np.eye(4)
        E[:3, :3], E[:3, 3] = Ry @ Rx, [0.06 * np.cos(2 * np.pi * v / 3), 0.04 * np.sin(2 * np.pi * v / 3), 0.01 * (v - 1)]
        poses.append(E)
    it = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for factor in (0.75, 1.0):
            for m in models:
                m.lod(factor)
            opts = [m.get_optimizer(0) for m in models]
            assert opts[1].fused_backward and not opts[0].fused_backward
            fh, fw = int(H * factor), int(W * factor)
            Kl = K.copy()
            Kl[:2] *= factor
            for (oy, ox) in [(0, 0), (fh - h, fw - w), (45, 80), (0, fw - w), (fh - h, 0), (30, 100)]:
                Kc = Kl.copy()
                Kc[0, 2] -= ox
                Kc[1, 2] -= oy
                E = poses[it % 3]
                losses = []
                for m, opt in zip(models, opts):
                    for grp in opt.param_groups:
                        grp["lr"] = 5e-3 * 0.95 ** it
                    opt.zero_grad(set_to_none=True)
                    _, extra = m(h, w, torch.tensor(E)[None], torch.tensor(Kc)[None], res=res, losscfg=dict(cfg))
                    loss = extra["swd"].sum() + 0.2 * extra["rgb_smooth"].sum() + 0.2 * extra["a_smooth"].sum()
                    loss.backward()
                    opt.step()
                    losses.append(float(loss.detach()))
                assert losses[0] == losses[1], (it, losses)
                sa, sb = opts[0].state[opts[0].p], opts[1].state[opts[1].p]
                assert torch.equal(sa["last_step"], sb["last_step"])
                for name, x, y in (("p", A.stack.data, B.stack.data), ("m", sa["exp_avg"], sb["exp_avg"]), ("v", sa["exp_avg_sq"], sb["exp_avg_sq"])):
                    if sparse:      # frames 1.. of a static texel are scratch until the flush; culled slots are nobody's
                        x, y = x[:, :1], y[:, :1]
                        kept = tiles.quad_to_texel_mask(A.quad_keep.cpu(), *A.stack.shape[2:4]).to(dev)[:, None, :, :, None].expand_as(x)
                        x, y = x[kept], y[kept]
                    assert torch.equal(x, y), (it, name, int((x != y).sum()))
                it += 1
            assert opts[1].fused_steps == 6
    sda, sdb = A.state_dict()["stack"], B.state_dict()["stack"]
    if sparse:
        kept = tiles.quad_to_texel_mask(A.quad_keep.cpu(), *A.stack.shape[2:4]).to(dev)[:, None, :, :, None].expand_as(sda)
        assert torch.equal(sda[kept], sdb[kept])
    else:
        assert torch.equal(sda, sdb)


def test_fused_backward_announces_itself_and_refuses_stray_gradients(dev):
    """The fused step changes what `loss.backward()` means (the update happens there; `step()` is housekeeping): the optimiser says so ONCE
    unless the training loop has acknowledged it, says so once more when a second windowed forward arrives before `step()` (every backward
    is its own Adam step: nothing accumulates), and RAISES when a gradient reached the window leaf through another autograd path -- it would
    have been dropped silently."""
    import warnings
    from videoloop3d_amd.MPV import MPMeshVid
    H, W, h, w, T = 96, 128, 48, 64, 4
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], np.float64)
    tar = np.eye(4)
    tar[:3, 3] = [0.03, 0.01, 0.0]
    res = synth.hash_uniform((1, 2 * T + 1, 3, h, w), seed=8, device=dev)
    cfg = dict(loss_name=["gpnn_lm"], loss_gain=torch.tensor([1.0]), macro_block=torch.tensor([65]), patch_size=torch.tensor([3]),
               stride=torch.tensor([2]), patcht_size=torch.tensor([3]), stridet=torch.tensor([1]), alpha=torch.tensor([10000.0]),
               dist_fn=["mse"], rou=["-2"], scaling=torch.tensor([0.1]))
    model = MPMeshVid(_args(mpv_frm_num=T), H, W, np.eye(4), K, 1.0, 100.0).to(dev).train()
    opt = model.get_optimizer(0)
    assert opt.fused_backward

    def fwd():
        _, extra = model(h, w, torch.tensor(tar)[None], torch.tensor(K)[None], res=res, losscfg=dict(cfg))
        return extra["swd"].sum()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        before = model.stack.detach().clone()
        fwd().backward()
        assert not torch.equal(model.stack.detach(), before)                 # the backward moved the parameters ...
        assert sum("APPLIES the Adam update" in str(w_.message) for w_ in rec) == 1      # ... and said so
        opt.step()
        fwd().backward()
        opt.step()
        assert sum("APPLIES the Adam update" in str(w_.message) for w_ in rec) == 1      # once
        # two forward / backward pairs without step(): legal, announced once
        fwd().backward()
        fwd().backward()
        opt.step()
        assert sum("second windowed forward" in str(w_.message) for w_ in rec) == 1 and opt.t == 4
    # a loop that knows (train_3dvid.run_iter / train_3d.run_iter call this) hears nothing
    m2 = MPMeshVid(_args(mpv_frm_num=T), H, W, np.eye(4), K, 1.0, 100.0).to(dev).train()
    o2 = m2.get_optimizer(0)
    o2.acknowledge_fused_backward()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        _, extra = m2(h, w, torch.tensor(tar)[None], torch.tensor(K)[None], res=res, losscfg=dict(cfg))
        extra["swd"].sum().backward()
        o2.step()
        assert not any("APPLIES" in str(w_.message) for w_ in rec)
    # a gradient that reaches the window leaf around the render: step() refuses instead of dropping it
    _, extra = m2(h, w, torch.tensor(tar)[None], torch.tensor(K)[None], res=res, losscfg=dict(cfg))
    leaf = o2.pending[1]
    (extra["swd"].sum() + 1e-3 * (leaf ** 2).sum()).backward()
    with pytest.raises(RuntimeError, match="outside the render's fused backward"):
        o2.step()


@pytest.mark.gpu
def test_reserve_windows_takes_the_buffer_growth_out_of_the_epoch(dev):
    """MPMeshVid.reserve_windows (train_3dvid.train calls it per pyramid level): the crop-aware optimiser's persistent window buffers are sized for the
    largest crop window of a set of views BEFORE the epoch, so no iteration grows them -- a growth is a multi-GB hipMalloc in the middle of training
    (docs/measurement_log.md, round 6).  Views in increasing window order, which without the reservation grows the buffers again and again."""
    import warnings
    from videoloop3d_amd.MPV import MPMeshVid
    H, W, T, D = 96, 160, 4, 6
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], np.float64)
    res_cfg = dict(loss_name=["gpnn_lm"], loss_gain=torch.tensor([1.0]), macro_block=torch.tensor([65]), patch_size=torch.tensor([3]),
                   stride=torch.tensor([2]), patcht_size=torch.tensor([3]), stridet=torch.tensor([1]), alpha=torch.tensor([10000.0]),
                   dist_fn=["mse"], rou=["-2"], scaling=torch.tensor([0.1]))
    views = []
    for (h, w, ox, oy, tx) in [(32, 48, 10, 8, 0.0), (40, 64, 30, 20, 0.02), (48, 96, 50, 30, -0.03), (64, 128, 16, 16, 0.05)]:
        Kc = K.copy()
        Kc[0, 2] -= ox
        Kc[1, 2] -= oy
        E = np.eye(4)
        E[0, 3] = tx
        views.append((h, w, torch.tensor(E)[None], torch.tensor(Kc)[None]))
    grown = {}
    for reserve in (False, True):
        torch.manual_seed(3)
        m = MPMeshVid(_args(mpv_frm_num=T, mpi_d=D, atlas_grid_h=2, init_std=0.02), H, W, np.eye(4), K, 1.0, 100.0).to(dev).train()
        opt = m.get_optimizer(0)
        if reserve:
            assert m.reserve_windows(views) > 0
        ptrs = set()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for (h, w, E, Kc) in views:
                resv = synth.hash_uniform((1, 2 * T + 1, 3, h, w), seed=8, device=dev)
                opt.zero_grad(set_to_none=True)
                _, extra = m(h, w, E, Kc, res=resv, losscfg=dict(res_cfg))
                extra["swd"].sum().backward()
                opt.step()
                ptrs.add((opt._gfb.data_ptr() if opt._gfb is not None else 0, opt._compact_buf.data_ptr() if opt._compact_buf is not None else 0))
        grown[reserve] = len(ptrs)
    assert grown[True] == 1, grown      # reserved: the same buffers from the first iteration to the last
    assert grown[False] > 1, grown      # (and the test's views do grow them otherwise)
