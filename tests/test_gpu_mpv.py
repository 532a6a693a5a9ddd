"""MPMeshVid (module-level drop-in, MPV.py:26-556) on the HIP kernels vs the CPU oracle of MPV.forward."""
import types

import numpy as np
import pytest
import torch

from oracle import mpv_oracle
from videoloop3d_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    g.build()
    return torch.device("cuda:0")


def make_args(**kw):
    a = dict(mpv_frm_num=6, mpv_isloop=True, mpi_h_scale=1.1, mpi_w_scale=1.1, mpi_d=6, atlas_grid_h=2, init_std=0.5,
             rgb_mlp_type="direct", rgb_activate="sigmoid", alpha_activate="sigmoid", bg_color="", scale_invariant=True,
             add_uv_noise=False, fp16=False, normalize_verts=False, atlas_cnl=4,
             swd_patch_size=3, swd_patcht_size=3, swd_stride=2, swd_stridet=1,
             sparsity_loss_weight=0.0, rgb_smooth_loss_weight=0.2, a_smooth_loss_weight=0.2, density_loss_weight=0.0,
             d_smooth_loss_weight=0.0)
    a.update(kw)
    return types.SimpleNamespace(**a)


def scene(H=44, W=60):
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], np.float64)
    ref_extrin = np.eye(4)
    ref_extrin[:3, 3] = [0.01, -0.02, 0.03]
    a = np.radians(1.0)
    tar = np.eye(4)
    tar[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
    tar[:3, 3] = [0.04, 0.015, 0.0]
    return K, ref_extrin, tar


def collate(cfg):
    return {k: ([v] if isinstance(v, str) else torch.tensor([v])) for k, v in cfg.items()}


@pytest.mark.parametrize("which", ["other", "ref"])
@pytest.mark.parametrize("bg", ["", "0.2#0.4#0.6"])
def test_forward_train_matches_oracle(dev, which, bg):
    from videoloop3d_amd.MPV import MPMeshVid
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    args = make_args(bg_color=bg, sparsity_loss_weight=0.1 if bg else 0.0, density_loss_weight=0.05 if bg else 0.0)
    torch.manual_seed(0)
    model = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev)
    with torch.no_grad():
        model.stack.copy_(synth.make_plane_stack(*model.stack.shape[:4], seed=5) * 0.7)
    stack_cpu = model.stack.detach().cpu().clone().requires_grad_(True)
    h, w = 33, 47   # a training crop: shifted principal point (train_3dvid.py:60-66)
    Kc = K.copy(); Kc[0, 2] -= 6; Kc[1, 2] -= 5
    tar_e = torch.tensor(tar)[None]
    tar_k = torch.tensor(Kc)[None]
    res = synth.hash_uniform((1, 9, 3, h, w), seed=8)
    if which == "ref":
        cfg = dict(loss_name="gpnn_lm", loss_gain=3.5, macro_block=21, patch_size=11, stride=4, patcht_size=3, stridet=1,
                   alpha=0.5, dist_fn="mse", rou="-2", scaling=0.1)
    else:
        cfg = dict(loss_name="gpnn_lm", loss_gain=1.0, macro_block=21, patch_size=3, stride=2, patcht_size=3, stridet=1,
                   alpha=10000, dist_fn="mse", rou="-2", scaling=0.1)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.train()
        none, extra = model(h, w, tar_e.to(dev), tar_k.to(dev), res=res.to(dev), losscfg=collate(cfg))
        _, extra_o = mpv_oracle.mpv_forward(stack_cpu, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, tar_k, res=res,
                                            losscfg=collate(cfg), training=True)
    assert none is None
    assert set(extra) == set(extra_o)
    weights = {"swd": 1.0, "rgb_smooth": args.rgb_smooth_loss_weight, "a_smooth": args.a_smooth_loss_weight,
               "sparsity": args.sparsity_loss_weight, "density": args.density_loss_weight}
    tot = sum(extra[k].sum() * weights[k] for k in extra)
    tot_o = sum(extra_o[k].sum() * weights[k] for k in extra_o)
    for k in extra:
        assert extra[k].shape == (1, 1)
        assert abs(extra[k].item() - extra_o[k].item()) <= 2e-5 * max(1.0, abs(extra_o[k].item())), k
    (g,) = torch.autograd.grad(tot, model.stack)
    (g_o,) = torch.autograd.grad(tot_o, stack_cpu)
    assert float((g.cpu() - g_o).abs().max()) <= 1e-4 * max(1.0, float(g_o.abs().max()))
    assert float(g_o.abs().max()) > 0


@pytest.mark.parametrize("which", ["other", "ref", "one_term", "no_terms", "generic_fallback"])
def test_stage2_objective_equals_forward_plus_weighted_total(dev, which):
    """MPMeshVid.objective (the weighted total in one launch each way, vl3d_linear_head_*) against forward + train_3dvid.weighted_total: the
    total, the swd term, every weighted regulariser and the gradient w.r.t. the stack -- also under an upstream gradient that is not 1, and
    through the generic spelling it falls back to when a per-pixel term is on."""
    from videoloop3d_amd.MPV import MPMeshVid
    from videoloop3d_amd.train_3dvid import weighted_total
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    kw = {}
    if which == "one_term":
        kw["a_smooth_loss_weight"] = 0.0
    if which == "no_terms":
        kw.update(a_smooth_loss_weight=0.0, rgb_smooth_loss_weight=0.0)
    if which == "generic_fallback":
        kw.update(sparsity_loss_weight=0.1, density_loss_weight=0.05)
    args = make_args(**kw)
    model = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev).train()
    with torch.no_grad():
        model.stack.copy_(synth.make_plane_stack(*model.stack.shape[:4], seed=5) * 0.7)
    h, w = 33, 47
    Kc = K.copy(); Kc[0, 2] -= 6; Kc[1, 2] -= 5
    tar_e, tar_k = torch.tensor(tar)[None], torch.tensor(Kc)[None]
    res = synth.hash_uniform((1, 9, 3, h, w), seed=8).to(dev)
    if which == "ref":
        cfg = dict(loss_name="gpnn_lm", loss_gain=3.5, macro_block=21, patch_size=11, stride=4, patcht_size=3, stridet=1,
                   alpha=0.5, dist_fn="mse", rou="-2", scaling=0.1)
    else:
        cfg = dict(loss_name="gpnn_lm", loss_gain=1.0, macro_block=21, patch_size=3, stride=2, patcht_size=3, stridet=1,
                   alpha=10000, dist_fn="mse", rou="-2", scaling=0.1)
    wt = lambda k: getattr(args, f"{k}_loss_weight", 0)  # noqa: E731
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for up in (1.0, 0.4):
            la, sa, ea = model.objective(h, w, tar_e, tar_k, res, collate(cfg))
            (ga,) = torch.autograd.grad(la * up, model.stack)
            _, extra = model(h, w, tar_e, tar_k, res=res, losscfg=collate(cfg))
            swd = extra.pop("swd")
            lb, (sb,), eb = weighted_total([swd], extra, wt)
            (gb,) = torch.autograd.grad(lb * up, model.stack)
            assert abs(float(la) - float(lb)) <= 2e-6 * max(1.0, abs(float(lb)))
            assert abs(float(sa) - float(sb)) <= 2e-6 * max(1.0, abs(float(sb)))
            assert sorted(ea) == sorted(eb)
            for k in ea:
                assert abs(float(ea[k]) - float(eb[k])) <= 2e-6 * max(1e-3, abs(float(eb[k]))), k
            assert float((ga - gb).abs().max()) <= 2e-6 * max(1e-6, float(gb.abs().max()))
            assert float(gb.abs().max()) > 0


def test_forward_eval_and_frame_subset(dev):
    from videoloop3d_amd.MPV import MPMeshVid
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    args = make_args(rgb_smooth_loss_weight=0.0, a_smooth_loss_weight=0.0)
    model = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev).eval()
    with torch.no_grad():
        model.stack.copy_(synth.make_plane_stack(*model.stack.shape[:4], seed=5))
        tar_e, tar_k = torch.tensor(tar)[None], torch.tensor(K)[None]
        for ts in (None, torch.tensor([4]), torch.tensor([5, 0, 2])):
            rgb, extra = model(H, W, tar_e.to(dev), tar_k.to(dev), ts=None if ts is None else ts.to(dev))
            rgb_o, _ = mpv_oracle.mpv_forward(model.stack.detach().cpu(), args, H, W, ref_extrin, K, 1.0, 100.0, H, W, tar_e, tar_k,
                                              ts=ts, training=False)
            assert extra == {} and rgb.shape == rgb_o.shape
            assert float((rgb.cpu() - rgb_o).abs().max()) <= 1e-4


def test_d_smooth_regulariser_slow_path(dev):
    """d_smooth (MPV.py:463-466, 539-551; off in every shipped configuration): finite differences of the blend-weighted inverse view depth.
    The only term that reads materialised layers: value and stack gradient against the oracle, with the crop-aware optimiser's window path
    active (the layers are then sampled from the compact window copy)."""
    from videoloop3d_amd.MPV import MPMeshVid
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    args = make_args(d_smooth_loss_weight=0.5, rgb_smooth_loss_weight=0.0, a_smooth_loss_weight=0.0, lrate=0.05, lrate_decay=30, optimizer="adam")
    model = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev).train()
    with torch.no_grad():
        model.stack.copy_(synth.make_plane_stack(*model.stack.shape[:4], seed=5) * 0.7)
    res = synth.make_video(9, 24, 32, seed=31)[0].permute(1, 0, 2, 3)[None].contiguous()
    cfg = {"loss_name": "gpnn_lm", "patch_size": 3, "patcht_size": 3, "stride": 2, "stridet": 1, "alpha": 10000.0, "rou": "-2", "scaling": 0.1,
           "macro_block": 65}
    Kc = K.copy(); Kc[0, 2] -= 11; Kc[1, 2] -= 7
    tar_e, tar_k = torch.tensor(tar)[None], torch.tensor(Kc)[None]
    s_cpu = model.stack.detach().cpu().clone().requires_grad_(True)
    _, ex_o = mpv_oracle.mpv_forward(s_cpu, args, H, W, ref_extrin, K, 1.0, 100.0, 24, 32, tar_e, tar_k, res=res, losscfg=collate(cfg))
    (g_o,) = torch.autograd.grad(ex_o["swd"].sum() + 3.0 * ex_o["d_smooth"].sum(), s_cpu)
    opt = model.get_optimizer(0)                                  # window path: the render reads the compact copy of the crop's window
    _, ex = model(24, 32, tar_e, tar_k, res=res.to(dev), losscfg=collate(cfg))
    assert abs(ex["d_smooth"].item() - ex_o["d_smooth"].item()) <= 1e-5 * max(1.0, abs(ex_o["d_smooth"].item()))
    (ex["swd"].sum() + 3.0 * ex["d_smooth"].sum()).backward()
    win, leaf = opt.pending[0], opt.pending[1]
    y0, x0, wh, ww = win
    g = torch.zeros_like(model.stack)
    g[:, :, y0:y0 + wh, x0:x0 + ww] = leaf.grad
    d = (g.cpu() - g_o).abs()
    scale = max(1e-6, float(g_o.abs().max()))
    assert float(d.max()) <= 5e-3 * scale and float((d > 1e-4 * scale).float().mean()) <= 1e-3


def test_render_variables_on_request(dev):
    """render(..., need_layers=True): `mpi` in the reference's hit-slot order (MPV.py:441-449) and `blend_weight` (MPV.py:451-453), whose
    composite is the fused kernel's image and whose sum is `alpha` (MPV.py:454); off-centre camera so plane edges cross the view."""
    from oracle import mpi_oracle as MO
    from videoloop3d_amd.MPV import MPMeshVid
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    tar = tar.copy()
    tar[:3, 3] = [0.35, 0.2, 0.0]
    model = MPMeshVid(make_args(), H, W, ref_extrin, K, 1.0, 100.0).to(dev).eval()
    with torch.no_grad():
        model.stack.copy_(synth.make_plane_stack(*model.stack.shape[:4], seed=5))
        extr = torch.tensor(tar)[None].to(dev) @ torch.tensor(ref_extrin)[None].inverse().to(dev)
        rgb, v = model.render(H, W, extr, torch.tensor(K)[None].to(dev), torch.arange(model.frm_num), need_layers=True)
        homos = model.plane_homographies(extr, torch.tensor(K)[None].to(dev)).cpu().float()
    spec = MO.RenderSpec(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post")
    rgb_o, alpha_o, bw_o, slots = MO.render_planes(model.stack.detach().cpu(), homos, H, W, spec, return_layers=True)
    assert v["mpi"].shape == slots.shape and float((v["mpi"].cpu() - slots).abs().max()) <= 1e-4
    bw = v["blend_weight"]
    assert float(((bw[..., None] * v["mpi"][..., :3]).sum(-2) - rgb).abs().max()) <= 1e-4
    assert float((bw.sum(-1) - v["alpha"]).abs().max()) <= 1e-4
    assert float((slots[..., 3] == 0).float().mean()) > 0.02      # some slots are empty: plane edges are inside the view


def test_product_library_refuses_timing_only_variants(dev):
    """desc->variant bits 4-7 (ablations that skip parts of a kernel: wrong results) exist only in a -DVL3D_VARIANTS measurement
    build; the product returns VL3D_EINVAL for them on every entry point that takes a descriptor."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    stack = synth.make_plane_stack(2, 1, 8, 8, seed=1).to(dev)
    with pytest.raises(RuntimeError, match="VL3D_VARIANTS"):
        render_planes(stack, torch.eye(3).repeat(2, 1, 1).to(dev), 4, 4, RenderSpec.mpv(variant=0x10))
    from videoloop3d_amd import _lib as L
    d = L.LossDesc()
    d.Tx, d.Ty, d.H, d.W, d.ps, d.pt, d.stride, d.stridet, d.variant = 4, 4, 8, 8, 3, 3, 1, 1, 0x40
    assert L.lib().vl3d_patchnn(d, None, None, None, None, L.stream_ptr(dev)) == 1


def test_atlas_roundtrip():
    from videoloop3d_amd.MPV import atlas_to_stack, stack_to_atlas
    atlas = torch.arange(3 * 4 * 10 * 28, dtype=torch.float32).reshape(3, 4, 10, 28)   # T=3, grid 2x4 of 5x7 cells
    st = atlas_to_stack(atlas, 8, 2)
    assert st.shape == (8, 3, 5, 7, 4)
    assert torch.equal(st[5, 1, 2, 3], atlas[1, :, 1 * 5 + 2, 1 * 7 + 3])             # plane 5 = cell (1, 1)
    assert torch.equal(stack_to_atlas(st, 2), atlas)


# ---- stage 1: MPMesh (MPI.py) -----------------------------------------------------------------------------------------
def make_args_mpi(**kw):
    a = dict(mpi_h_scale=1.3, mpi_w_scale=1.3, mpi_d=6, atlas_grid_h=2, rgb_mlp_type="direct", rgb_activate="sigmoid",
             alpha_activate="sigmoid", bg_color="", learn_loop_mask=True, upsample_stage="",
             sparsity_loss_weight=0.004, rgb_smooth_loss_weight=0.2, a_smooth_loss_weight=0.5, density_loss_weight=0.02,
             d_smooth_loss_weight=0.0, l_smooth_loss_weight=0.0)
    a.update(kw)
    return types.SimpleNamespace(**a)


@pytest.mark.parametrize("loop_mask", [True, False])
def test_mpmesh_forward_train_matches_oracle(dev, loop_mask):
    """stage-1 module (cfg2 caller): rgb + loop-mask channel + the four regularisers of configs/mpi_base.txt, value and grads,
    two views in the batch."""
    from videoloop3d_amd.MPI import MPMesh
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    args = make_args_mpi(learn_loop_mask=loop_mask)
    model = MPMesh(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev).train()
    with torch.no_grad():
        model.stack.copy_(synth.make_plane_stack(*model.stack.shape[:4], seed=5) * 0.7)
        if loop_mask:
            model.stack_mask.copy_(synth.hash_uniform(tuple(model.stack_mask.shape), seed=6) * 3 - 2)
    stack_cpu = model.stack.detach().cpu().clone().requires_grad_(True)
    mask_cpu = model.stack_mask.detach().cpu().clone().requires_grad_(True) if loop_mask else None
    h, w = 33, 47
    Kc = K.copy(); Kc[0, 2] -= 6; Kc[1, 2] -= 5
    tar2 = tar.copy(); tar2[:3, 3] = [-0.03, 0.02, 0.01]
    tar_e = torch.tensor(np.stack([tar, tar2]))
    tar_k = torch.tensor(np.stack([Kc, Kc]))
    rgbl, extra = model(h, w, tar_e.to(dev), tar_k.to(dev))
    rgbl_o, extra_o = mpv_oracle.mpi_forward(stack_cpu, mask_cpu, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, tar_k)
    assert rgbl.shape == rgbl_o.shape == (2, 4 if loop_mask else 3, h, w)
    assert float((rgbl.cpu() - rgbl_o).abs().max()) <= 1e-4
    assert set(extra) == set(extra_o) == {"sparsity", "rgb_smooth", "a_smooth", "density"}
    for k in extra:
        assert abs(extra[k].item() - extra_o[k].item()) <= 2e-5 * max(1.0, abs(extra_o[k].item())), k
    g = synth.hash_uniform(tuple(rgbl.shape), seed=9) - 0.5
    wts = {"sparsity": 0.004, "rgb_smooth": 0.2, "a_smooth": 0.5, "density": 0.02}
    tot = (rgbl * g.to(dev)).sum() + sum(extra[k].sum() * wts[k] for k in extra) * 50
    tot_o = (rgbl_o * g).sum() + sum(extra_o[k].sum() * wts[k] for k in extra_o) * 50
    params = [model.stack] + ([model.stack_mask] if loop_mask else [])
    params_o = [stack_cpu] + ([mask_cpu] if loop_mask else [])
    grads = torch.autograd.grad(tot, params)
    grads_o = torch.autograd.grad(tot_o, params_o)
    for gg, go in zip(grads, grads_o):
        d = (gg.cpu() - go).abs()
        scale = max(1.0, float(go.abs().max()))
        assert float(d.max()) <= 5e-3 * scale and float((d > 1e-4 * scale).float().mean()) <= 1e-3
        assert float(go.abs().max()) > 0


@pytest.mark.parametrize("normalise,bg", [(False, ""), (True, "0.2#0.4#0.6")])
def test_mpmesh_layer_terms_match_oracle(dev, normalise, bg):
    """the terms that read materialised layers (MPI.py:558-566, 622-645; off in the shipped configs, slow path): l_smooth over the
    slot-ordered loop-mask layers, edge-weighted d_smooth of the normalised disparity map, normalize_blendweight_fordepth -- against
    mpv_oracle.mpi_forward, which golden G17 (a2 / a3) pins to the reference's own MPMesh.forward."""
    from videoloop3d_amd.MPI import MPMesh
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    args = make_args_mpi(l_smooth_loss_weight=0.3, d_smooth_loss_weight=0.1, edge_scale=4.0, normalize_blendweight_fordepth=normalise, bg_color=bg)
    model = MPMesh(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev).train()
    with torch.no_grad():
        model.stack.copy_(synth.make_plane_stack(*model.stack.shape[:4], seed=5) * 0.7)
        model.stack_mask.copy_(synth.hash_uniform(tuple(model.stack_mask.shape), seed=6) * 3 - 2)
    stack_cpu = model.stack.detach().cpu().clone().requires_grad_(True)
    mask_cpu = model.stack_mask.detach().cpu().clone().requires_grad_(True)
    h, w = 33, 47
    Kc = K.copy(); Kc[0, 2] -= 6; Kc[1, 2] -= 5
    tar_e, tar_k = torch.tensor(tar[None]), torch.tensor(Kc[None])
    rgbl, extra = model(h, w, tar_e.to(dev), tar_k.to(dev))
    rgbl_o, extra_o = mpv_oracle.mpi_forward(stack_cpu, mask_cpu, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, tar_k)
    assert float((rgbl.cpu() - rgbl_o).abs().max()) <= 1e-4
    assert set(extra) == set(extra_o) == {"sparsity", "rgb_smooth", "a_smooth", "density", "d_smooth", "l_smooth"}
    for k in extra:
        assert abs(extra[k].item() - extra_o[k].item()) <= 3e-5 * max(1.0, abs(extra_o[k].item())), k
    tot = (extra["d_smooth"] + extra["l_smooth"]).sum()
    tot_o = (extra_o["d_smooth"] + extra_o["l_smooth"]).sum()
    grads = torch.autograd.grad(tot, [model.stack, model.stack_mask])
    grads_o = torch.autograd.grad(tot_o, [stack_cpu, mask_cpu])
    for gg, go in zip(grads, grads_o):
        scale = max(1e-6, float(go.abs().max()))
        assert float((gg.cpu() - go).abs().max()) <= 2e-3 * scale and float(go.abs().max()) > 0
    # the variables dict carries the materialised tensors on this path (MPI.py:585-592)
    ext = tar_e @ torch.tensor(ref_extrin)[None].inverse()
    _, var = model.render(h, w, ext.to(dev), tar_k.to(dev), need_layers=True)
    K_ = var["mpi"].shape[3]
    assert var["mpi"].shape == (1, h, w, K_, 4) and var["loopmask3d"].shape == (1, h, w, K_, 1) and var["blend_weight"].shape == (1, h, w, K_)
    assert var["disp_norm"].shape == (1, h, w)


def test_sparsified_mpmesh_trains_with_its_quad_map(dev):
    """train_3d.py:282-285: the last epochs of stage 1 train the SPARSIFIED mesh.  MPMesh.forward after sparsify_faces renders with the quad
    map (a sample in a culled quad is not covered: MPI.py:483-487, 544-548) -- value, regularisers (hit-slot order) and gradient against
    the oracle with the same map, and the loop-mask channel is gone (MPI.py:440-441)."""
    from videoloop3d_amd.MPI import MPMesh
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    args = make_args_mpi(mpi_h_verts=5, mpi_w_verts=7, optimizer='adam', lrate=0.05)
    model = MPMesh(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev).train()
    D, _, Hs, Ws, _ = model.stack.shape
    with torch.no_grad():
        st = synth.make_plane_stack(D, 1, Hs, Ws, seed=5) * 0.7
        yy, xx = torch.meshgrid(torch.arange(Hs).float(), torch.arange(Ws).float(), indexing="ij")
        for d in range(D):           # an alpha blob per plane: part of every plane is culled, planes 4 and 5 entirely
            blob = 6.0 * torch.exp(-(((yy - Hs * (0.3 + 0.1 * d)) / 9) ** 2 + ((xx - Ws * (0.2 + 0.12 * d)) / 12) ** 2)) - 4.5
            st[d, 0, :, :, 3] = blob if d < 4 else -8.0
        model.stack.copy_(st)
        model.stack_mask.copy_(synth.hash_uniform(tuple(model.stack_mask.shape), seed=6) * 3 - 2)
    model.sparsify_faces(erode_num=1, alpha_thresh=0.05)
    keep = model.quad_keep.cpu()
    assert 0.1 < float(keep.float().mean()) < 0.7 and not keep[4:].any() and not model.learn_loop_mask
    stack_cpu = model.stack.detach().cpu().clone().requires_grad_(True)
    h, w = 33, 47
    Kc = K.copy(); Kc[0, 2] -= 6; Kc[1, 2] -= 5
    tar_e, tar_k = torch.tensor(tar[None]), torch.tensor(Kc[None])
    rgbl, extra = model(h, w, tar_e.to(dev), tar_k.to(dev))
    rgbl_o, extra_o = mpv_oracle.mpi_forward(stack_cpu, None, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, tar_k, quad_keep=keep)
    assert rgbl.shape == (1, 3, h, w) and float((rgbl.cpu() - rgbl_o).abs().max()) <= 1e-4
    for k in ("sparsity", "rgb_smooth", "a_smooth", "density"):
        assert abs(extra[k].item() - extra_o[k].item()) <= 2e-5 * max(1.0, abs(extra_o[k].item())), k
    # ... and NOT the plane-indexed reading (culled regions differenced as sigmoid(rgb) layers): the quad map reached the kernels
    _, extra_p = mpv_oracle.mpi_forward(stack_cpu.detach(), None, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, tar_k)
    assert abs(extra_p["rgb_smooth"].item() - extra_o["rgb_smooth"].item()) > 1e-3
    g = synth.hash_uniform(tuple(rgbl.shape), seed=9) - 0.5
    wts = {"sparsity": 0.004, "rgb_smooth": 0.2, "a_smooth": 0.5, "density": 0.02}
    tot = (rgbl * g.to(dev)).sum() + sum(extra[k].sum() * wts[k] for k in wts) * 50
    tot_o = (rgbl_o * g).sum() + sum(extra_o[k].sum() * wts[k] for k in wts) * 50
    (gg,) = torch.autograd.grad(tot, model.stack)
    (go,) = torch.autograd.grad(tot_o, stack_cpu)
    d = (gg.cpu() - go).abs()
    scale = max(1.0, float(go.abs().max()))
    assert float(d.max()) <= 5e-3 * scale and float((d > 1e-4 * scale).float().mean()) <= 1e-3
    before = model.stack.detach().clone()
    opt = model.get_optimizer()
    model.stack.grad = gg
    opt.step()                       # the stage-1 optimiser steps the sparsified model
    assert float((model.stack.detach() - before).abs().max()) > 1e-3


@pytest.mark.parametrize("C,si,layout", [(4, True, "nhwc"), (3, True, "nchw"), (4, False, "nhwc"), (3, False, "nhwc")])
def test_stage1_loss_equals_the_torch_chain(dev, C, si, layout):
    """train_3d.py:200-220 (loop-mask entropy on the clamped label, scale-invariant MSE) as vl3d_stage1_loss against the same lines in torch:
    both losses and the gradient w.r.t. the module's output, for the NHWC-backed view MPMesh.forward returns and a plain NCHW tensor."""
    from videoloop3d_amd.MPI import image_and_loop_loss
    B, h, w = 2, 37, 53
    base = synth.hash_uniform((B, h, w, C), seed=3).to(dev)
    if C == 4:
        base[..., 3] = base[..., 3] * 1.1 - 0.05                                     # a few labels outside the clamp
    rgbl = (base.permute(0, 3, 1, 2) if layout == "nhwc" else base.permute(0, 3, 1, 2).contiguous()).requires_grad_(True)
    target = synth.hash_uniform((B, 3, h, w), seed=4).to(dev)
    tm = (synth.hash_uniform((B, h, w), seed=5).to(dev) > 0.5).float()
    img, loop = image_and_loop_loss(rgbl, target, tm if C == 4 else None, scale_invariant=si)
    r = rgbl.detach().clone().requires_grad_(True)
    rgb = r[:, :3]
    loop_t = torch.zeros((), device=dev)
    if C == 4:
        lm = torch.clamp(r[:, -1], 0.001, 1 - 0.001)
        loop_t = -(tm * torch.log(lm) + (1 - tm) * torch.log(1 - lm)).mean()
    if si:
        sc = torch.exp(torch.log((target + 0.01) / (rgb.detach() + 0.01)).mean())
        rgb = rgb * ((sc + 3) / 4)
    img_t = ((rgb - target) ** 2).mean()
    assert abs(float(img) - float(img_t)) <= 2e-6 * max(1.0, float(img_t)) and abs(float(loop) - float(loop_t)) <= 2e-6 * max(1.0, float(loop_t))
    (ga,) = torch.autograd.grad(0.7 * img + 1.3 * loop, rgbl)
    (gb,) = torch.autograd.grad(0.7 * img_t + 1.3 * loop_t, r)
    assert float((ga - gb).abs().max()) <= 1e-6 * max(1.0, float(gb.abs().max())) + 1e-9
    if C == 4:
        assert float(gb[:, 3].abs().max()) > 0 and float((gb[:, 3] == 0).float().mean()) > 0.01       # clamped labels carry no gradient


@pytest.mark.parametrize("which", ["both", "sparsity", "density"])
def test_pixel_terms_equal_the_torch_chain(dev, which):
    """MPI.py:599-603, 647-650 / MPV.py:511-515, 533-536 as vl3d_pixel_terms: mean |a|_1 / max(|a|_2, eps) and mean |alpha - 1|, values and
    gradients, including pixels no plane covers (sum a^2 = 0: the ratio's denominator sits on its floor, zero gradient to it)."""
    from videoloop3d_amd.MPV import _PixelTerms, sparsity_ratio
    T, h, w = 3, 29, 41
    a = synth.hash_uniform((T, h, w, 8), seed=7).to(dev)
    a[:, :5, :7] = 0                                                             # uncovered pixels
    asum = torch.stack([a.sum(-1), (a * a).sum(-1)], -1).requires_grad_(True)
    alpha = (synth.hash_uniform((T, h, w), seed=8).to(dev) * 1.2).requires_grad_(True)
    alpha.data[0, 0, :4] = 1.0                                                   # |alpha - 1| at its kink
    out = _PixelTerms.apply(alpha if which != "sparsity" else None, asum if which != "density" else None, 1e-4)
    as2, al2 = asum.detach().clone().requires_grad_(True), alpha.detach().clone().requires_grad_(True)
    sp_t, dn_t = sparsity_ratio(as2, 1e-4).mean(), (al2 - 1).abs().mean()
    if which != "density":
        assert abs(float(out[0]) - float(sp_t)) <= 2e-6 * float(sp_t)
        (g1,) = torch.autograd.grad(out[0] * 1.7, asum)
        (g2,) = torch.autograd.grad(sp_t * 1.7, as2)
        assert float((g1 - g2).abs().max()) <= 1e-6 * float(g2.abs().max())
    if which != "sparsity":
        assert abs(float(out[1]) - float(dn_t)) <= 2e-6 * float(dn_t)
        (g1,) = torch.autograd.grad(out[1] * 0.3, alpha)
        (g2,) = torch.autograd.grad(dn_t * 0.3, al2)
        assert torch.equal(g1, g2) or float((g1 - g2).abs().max()) <= 1e-9


@pytest.mark.parametrize("gain", [True, False])
def test_loss_prologue_equals_the_torch_chain(dev, gain):
    """MPV.py:484-507 (loop padding, scale-invariant gain, layout) as three kernels against the same lines in torch, value and gradient."""
    from videoloop3d_amd.MPV import _LoopPrologue
    T, F, h, w, pad = 7, 9, 37, 53, 2
    rgb = synth.hash_uniform((T, h, w, 3), seed=3).to(dev).requires_grad_(True)
    # the target is a CROP of a larger clip, as the stage-2 dataset hands it out: read through its strides (vl3d_loop_gain_strided), no copy
    clip = synth.hash_uniform((F, 3, h + 9, w + 14), seed=4).to(dev)
    res = clip[:, :, 5:5 + h, 6:6 + w]
    assert not res.is_contiguous()
    x, _ = _LoopPrologue.apply(rgb, res if gain else None, pad)
    if gain:
        x_c, _ = _LoopPrologue.apply(rgb, res.contiguous(), pad)
        assert torch.equal(x, x_c)
    r = rgb.detach().clone().requires_grad_(True)
    rp = r.permute(0, 3, 1, 2)
    rp_pad = torch.cat([rp, rp[:pad]], 0)
    if gain:
        scale = torch.exp(torch.log((res.mean(dim=0) + 0.01) / (rp.detach().mean(dim=0) + 0.01)).mean())
        rp_pad = rp_pad * ((scale + 3) / 4)
    x_t = rp_pad.permute(1, 0, 2, 3)[None]
    assert x.shape == x_t.shape and float((x - x_t).abs().max()) <= 2e-6
    g = synth.hash_uniform(tuple(x.shape), seed=5).to(dev) - 0.5
    (ga,) = torch.autograd.grad((x * g).sum(), rgb, retain_graph=True)
    (gb,) = torch.autograd.grad((x_t * g).sum(), r)
    assert ga.is_contiguous() and float((ga - gb).abs().max()) <= 2e-6
    # a strided upstream gradient (the loss trims to the patch grid and writes through strides)
    big = torch.zeros((1, 3, T + pad + 1, h, w + 3), device=dev)
    gv = big[:, :, :T + pad, :, :w]
    gv.copy_(g)
    (gc,) = torch.autograd.grad(x, rgb, gv)
    assert torch.equal(gc, ga)


def test_loop_mask_fifth_channel_equals_the_label_pass(dev):
    """The loop mask composited as a fifth channel of the colour pass (vl3d_render_fwd_mask / _bwd_mask) against the separate label pass
    (a second render of a (mask logit, -, -, alpha logit) stack): label, gradient to the mask texture, and colours / regulariser terms /
    stack gradient untouched by the extra channel.  Owner-computes tile path, the atomics fallback (variant 1), with and without the
    layer regularisers, poses on the host and on the device."""
    import dataclasses
    from videoloop3d_amd.MPI import MPMesh
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    h, w = 33, 47
    Kc = K.copy(); Kc[0, 2] -= 6; Kc[1, 2] -= 5
    tar_e, tar_k = torch.tensor(tar)[None], torch.tensor(Kc)[None]

    def run(two_pass, reg, variant=0, host_pose=False):
        kw = {} if reg else dict(sparsity_loss_weight=0.0, rgb_smooth_loss_weight=0.0, a_smooth_loss_weight=0.0)
        args = make_args_mpi(loop_mask_two_pass=two_pass, **kw)
        m = MPMesh(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev).train()
        m.spec = dataclasses.replace(m.spec, variant=variant)
        with torch.no_grad():
            m.stack.copy_(synth.make_plane_stack(*m.stack.shape[:4], seed=5) * 0.7)
            m.stack_mask.copy_(synth.hash_uniform(tuple(m.stack_mask.shape), seed=6) * 3 - 2)
        if host_pose == "numpy":      # (what nn.DataParallel's scatter leaves on the host)
            rgbl, extra = m(h, w, tar_e.numpy(), tar_k.numpy())
        else:
            rgbl, extra = m(h, w, tar_e if host_pose else tar_e.to(dev), tar_k if host_pose else tar_k.to(dev))
        g = (synth.hash_uniform(tuple(rgbl.shape), seed=9) - 0.5).to(dev)
        tot = (rgbl * g).sum() + sum(v.sum() for v in extra.values()) * 10
        gs, gm = torch.autograd.grad(tot, [m.stack, m.stack_mask])
        return rgbl.detach(), {k: float(v) for k, v in extra.items()}, gs, gm

    for reg in (True, False):
        ref = run(True, reg)
        for variant, host_pose in ((0, False), (1, False), (0, True), (0, "numpy")):
            got = run(False, reg, variant, host_pose)
            assert float((got[0] - ref[0]).abs().max()) <= 2e-6, (reg, variant)
            assert got[1].keys() == ref[1].keys() and all(abs(got[1][k] - ref[1][k]) <= 1e-6 * max(1.0, abs(ref[1][k])) for k in ref[1])
            tol = 2e-5 if variant == 1 else 2e-6             # (the atomics fallback sums in a different order)
            assert float((got[2] - ref[2]).abs().max()) <= tol * max(1.0, float(ref[2].abs().max())), (reg, variant)
            assert float((got[3] - ref[3]).abs().max()) <= tol * max(1.0, float(ref[3].abs().max())), (reg, variant)
            assert float(ref[3].abs().max()) > 0
        # the flat 64 x 8 regions of the default and the 64 x 16 ones of variant 3: the same bits in both gradients
        a0, a3 = run(False, reg, 0), run(False, reg, 3)
        assert torch.equal(a0[2], a3[2]) and torch.equal(a0[3], a3[3]) and torch.equal(a0[0], a3[0])


def test_loop_mask_channel_is_refused_for_other_conventions(dev):
    from videoloop3d_amd import _lib as L
    from videoloop3d_amd.render import RenderSpec, _desc
    stack = torch.zeros((2, 1, 8, 8, 4), device=dev)
    mask = torch.zeros((2, 1, 8, 8), device=dev)
    out = torch.zeros((1, 4, 4, 3), device=dev)
    a = torch.zeros((1, 4, 4), device=dev)
    homos = torch.eye(3, device=dev).repeat(2, 1, 1)
    desc = _desc(stack, 4, 4, RenderSpec(), 0, 0)            # the utils_mpi convention
    with torch.cuda.device(dev):
        rc = L.lib().vl3d_render_fwd_mask(desc, L.ptr(stack), L.ptr(mask), L.ptr(homos), L.ptr(out), L.ptr(a), L.ptr(a.clone()), None, None, None,
                                          L.stream_ptr(dev))
    assert rc == 3 and b"loop-mask" in L.lib().vl3d_last_error()


def test_one_pass_adam_on_stage1_parameters_matches_torch_adam(dev):
    """MPMesh.get_optimizer: tiles.TileAdam without a quad map walks ANY contiguous float32 parameter (the plane stack and the
    [D,1,Hs,Ws] loop-mask texture) in one pass; same parameters as torch.optim.Adam after several steps with a changing lr."""
    from videoloop3d_amd.MPI import MPMesh
    H, W = 44, 60
    K, ref_extrin, _ = scene(H, W)
    args = make_args_mpi(optimizer="adam", lrate=0.05, lrate_decay=100)
    m = MPMesh(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev)
    opt = m.get_optimizer()
    assert type(opt).__name__ == "TileAdam"
    ref = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
    opt_t = torch.optim.Adam(ref, lr=0.05, betas=(0.9, 0.999))
    for it in range(6):
        for g_, o_ in ((opt.param_groups, opt), (opt_t.param_groups, opt_t)):
            for grp in g_:
                grp["lr"] = 0.05 * 0.9 ** it
        for i, (p, q) in enumerate(zip(m.parameters(), ref)):
            g = synth.hash_uniform(tuple(p.shape), seed=30 + 7 * it + i).to(dev) - 0.5
            p.grad, q.grad = g.clone(), g.clone()
        opt.step()
        opt_t.step()
    for p, q in zip(m.parameters(), ref):
        assert float((p - q).abs().max()) <= 2e-6


def test_mpmesh_eval(dev):
    from videoloop3d_amd.MPI import MPMesh
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    args = make_args_mpi()
    model = MPMesh(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev).eval()
    with torch.no_grad():
        rgbl, extra = model(H, W, torch.tensor(tar)[None].to(dev), torch.tensor(K)[None].to(dev))
        rgbl_o, _ = mpv_oracle.mpi_forward(model.stack.detach().cpu(), model.stack_mask.detach().cpu(), args, H, W, ref_extrin, K, 1.0,
                                           100.0, H, W, torch.tensor(tar)[None], torch.tensor(K)[None], training=False)
    assert extra == {} and float((rgbl.cpu() - rgbl_o).abs().max()) <= 1e-4


def test_lod_render_matches_oracle(dev):
    """MPV.py:140-146: after lod(factor) the planes are smaller but keep their extent -- the render must equal the oracle's
    render of the resized stack with the plane-pixel -> texel scale (w'-1)/(mpi_w-1)."""
    from oracle import mpi_oracle as MO
    from videoloop3d_amd.MPV import MPMeshVid
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    args = make_args(rgb_smooth_loss_weight=0.0, a_smooth_loss_weight=0.0)
    model = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev).eval()
    with torch.no_grad():
        model.stack.copy_(synth.make_plane_stack(*model.stack.shape[:4], seed=5))
    model.lod(0.5)
    hs, ws = model.stack.shape[2:4]
    assert (hs, ws) == (int(model.mpi_h * 0.5), int(model.mpi_w * 0.5))
    tar_e, tar_k = torch.tensor(tar)[None], torch.tensor(K)[None]
    with torch.no_grad():
        rgb, _ = model(H, W, tar_e.to(dev), tar_k.to(dev))
    homos = model.plane_homographies((tar_e @ torch.tensor(ref_extrin)[None].inverse()).to(dev), tar_k.to(dev)).cpu()
    spec = MO.RenderSpec(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post",
                         scale=((ws - 1) / (model.mpi_w - 1), (hs - 1) / (model.mpi_h - 1)))
    rgb_o, _, _ = MO.render_planes(model.stack.detach().cpu(), homos, H, W, spec)
    assert float((rgb.permute(0, 2, 3, 1).cpu() - rgb_o).abs().max()) <= 1e-4


def test_stage2_driver_trains(dev):
    """train_3dvid.py:262-290 on the device: two pyramid levels, shuffled crops, adaptive lr; the looping loss goes down."""
    from videoloop3d_amd import train_3dvid as drv
    from videoloop3d_amd.MPV import MPMeshVid
    H, W = 40, 56
    K, ref_extrin, tar = scene(H, W)
    args = make_args(mpv_frm_num=5, mpi_d=4, rgb_smooth_loss_weight=0.05, a_smooth_loss_weight=0.05,
                     pyr_minimal_dim=-1, pyr_stage="3", N_iters=7, pyr_factor=0.5, pyr_num_epoch=0,
                     patch_h_size=24, patch_w_size=32, patch_h_stride=16, patch_w_stride=24,
                     lrate=0.5, lrate_decay=30, lrate_adaptive=True, optimizer="adam", optimize_verts_gain=1,
                     add_intrin_noise=True, swd_loss_weight=1.0)
    torch.manual_seed(0)
    model = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev)
    vids = [synth.make_video(8, H, W, seed=21 + v, device=dev)[0].permute(1, 0, 2, 3).contiguous() for v in range(2)]   # [F,3,H,W]
    poses = torch.stack([torch.tensor(np.linalg.inv(ref_extrin))[:3], torch.tensor(np.linalg.inv(tar))[:3]]).float()
    intr = torch.tensor(K).float()[None].repeat(2, 1, 1)
    cfg = {"loss_name": "gpnn_lm", "patch_size": 3, "patcht_size": 3, "stride": 2, "stridet": 1, "alpha": 10000,
           "rou": "-2", "scaling": 0.1, "dist_fn": "mse", "macro_block": 65, "factor": 1}
    log = []
    import tempfile, os
    save_dir = tempfile.mkdtemp()
    args.i_weights = 2
    n = drv.train(model, args, vids, poses, intr, [cfg, dict(cfg, loss_gain=2.0)], H, W, device=dev,
                  on_step=lambda lvl, ep, it, loss, swd, extra: log.append((lvl, float(loss), sorted(extra))),
                  generator=torch.Generator().manual_seed(1), save_dir=save_dir)
    # checkpoints every 2 epochs with the reference's keys (train_3dvid.py:295-306); the last one reloads into a fresh model
    assert sorted(os.listdir(save_dir)) == ["l0_epoch_0001.tar", "l1_epoch_0000.tar", "l1_epoch_0002.tar"]
    ck = torch.load(os.path.join(save_dir, "l1_epoch_0002.tar"), weights_only=False)
    assert {"epoch_i", "epoch_total_step", "iter_total_step", "pyr_i", "train_factor", "hw", "network_state_dict"} <= set(ck)
    fresh = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev)
    fresh.init_from_mpi(ck["network_state_dict"])
    assert fresh.stack.shape == model.stack.shape
    # level 0: 20x28 frames < crop -> 1 crop x 2 views x 3 epochs; level 1: 40x56 -> 2x2 crops x 2 views x 4 epochs
    assert n == 2 * 3 + 8 * 4 == len(log)
    assert model.stack.shape[2:4] == (model.mpi_h, model.mpi_w)
    assert all(np.isfinite(l) for _, l, _ in log) and log[0][2] == ["a_smooth", "rgb_smooth"]
    fine = [l for lvl, l, _ in log if lvl == 1]
    assert np.mean(fine[-8:]) < np.mean(fine[:8])


def test_sparsified_mpi_to_video_training(dev):
    """MPI.py:288-442 -> MPV.py:235-288 on the dense stack: after sparsify_faces + init_from_mpi, culled quads render nothing and
    never change, static quads stay one texture shared by all frames through optimiser steps, dynamic quads become per-frame."""
    from videoloop3d_amd import tiles
    from videoloop3d_amd.MPI import MPMesh
    from videoloop3d_amd.MPV import MPMeshVid
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    a1 = make_args_mpi(learn_loop_mask=True, mpi_h_verts=5, mpi_w_verts=7, sparsify_rmfirstlayer=0)
    mpi = MPMesh(a1, H, W, ref_extrin, K, 1.0, 100.0).to(dev)
    Hs, Ws = mpi.stack.shape[2:4]
    with torch.no_grad():
        mpi.stack[2, 0, 6:30, 8:50, 3] = 2.0
        mpi.stack[4, 0, 10:40, 20:60, 3] = 1.0
        mpi.stack_mask[4, 0, 15:30, 28:50] = 4.0
    mpi.sparsify_faces(erode_num=1)
    keep_t = tiles.quad_to_texel_mask(mpi.quad_keep, Hs, Ws)
    dyn_t = tiles.quad_to_texel_mask(mpi.quad_dyn, Hs, Ws)
    static_t = keep_t & ~dyn_t
    assert bool(static_t.any()) and bool(dyn_t.any()) and bool((~keep_t).any())

    a2 = make_args(mpv_frm_num=5, mpi_d=a1.mpi_d, mpi_h_scale=a1.mpi_h_scale, mpi_w_scale=a1.mpi_w_scale, rgb_smooth_loss_weight=0.0, a_smooth_loss_weight=0.0, lrate=0.05, lrate_decay=30,
                   optimizer="adam", optimize_verts_gain=1)
    vid = MPMeshVid(a2, H, W, ref_extrin, K, 1.0, 100.0).to(dev)
    vid.init_from_mpi(mpi.state_dict())
    start = vid.stack.detach().clone()
    # culled quads are invisible: samples that fall into them are uncovered, and no sample of a kept quad reads their texels
    tar_e, tar_k = torch.tensor(tar)[None].to(dev), torch.tensor(K)[None].to(dev)
    vid.eval()
    with torch.no_grad():
        rgb0, _ = vid(H, W, tar_e, tar_k)
        vid.stack[..., :3].masked_fill_((~keep_t)[:, None, :, :, None], 9.0)
        rgb1, _ = vid(H, W, tar_e, tar_k)
        assert torch.equal(rgb0, rgb1)
    # a few optimiser steps against a random video
    vid.train()
    opt = vid.get_optimizer(0)
    res = synth.make_video(9, H, W, seed=31, device=dev)[0].permute(1, 0, 2, 3)[None].contiguous()      # [1,F,3,H,W]
    cfg = collate({"loss_name": "gpnn_lm", "patch_size": 3, "patcht_size": 3, "stride": 2, "stridet": 1, "alpha": 10000.0,
                   "rou": "-2", "scaling": 0.1, "macro_block": 65})
    for _ in range(3):
        _, extra = vid(H, W, tar_e, tar_k, res=res, losscfg=dict(cfg))
        opt.zero_grad()
        extra["swd"].mean().backward()
        opt.step()
    # a static texel is ONE parameter stored in frame 0 while training (optim.WindowAdam); reading the whole stack goes through
    # state_dict(), which replays the deferred updates and refreshes the other frames' slots
    now = vid.state_dict()["stack"].detach()
    assert torch.equal(now[..., 3][(~keep_t)[:, None].expand_as(now[..., 3])], start[..., 3][(~keep_t)[:, None].expand_as(now[..., 3])])
    st = static_t[:, None, :, :, None].expand_as(now)
    assert torch.equal(now[:, :1].expand_as(now)[st], now[st])                       # static texels identical in every frame
    assert not torch.equal(now[st], start[st])                                        # ... and they did learn
    dy = dyn_t[:, None, :, :, None].expand_as(now)
    assert not torch.equal(now[:, :1].expand_as(now)[dy], now[dy])                    # dynamic texels diverge between frames


def test_tie_static_grad_kernel_matches_definition(dev):
    """vl3d_tie_static_grad (in place, one kernel) == tiles.tie_static_grad (the rule in plain torch)."""
    from videoloop3d_amd import tiles
    torch.manual_seed(7)
    for (D, T, Hs, Ws, QH, QW) in ((3, 4, 37, 53, 5, 7), (2, 1, 8, 9, 2, 2), (4, 5, 64, 130, 3, 11)):
        keep = (torch.rand(D, QH, QW) < 0.6).to(dev)
        dyn = keep & (torch.rand(D, QH, QW) < 0.5).to(dev)
        g = torch.rand(D, T, Hs, Ws, 4, device=dev)
        want = tiles.tie_static_grad(g.clone(), keep, dyn)
        got = tiles.tie_static_grad_hip(g.clone(), keep, dyn)
        assert float((got - want).abs().max()) <= 1e-6 * T
        assert torch.equal(got == 0, want == 0)


def test_tile_adam_matches_torch_adam(dev):
    """TileAdam (vl3d_adam_step_tiles) == torch.optim.Adam on a tile-culled stack over several steps with changing lr; culled texels
    (zero gradient) are not touched at all."""
    from videoloop3d_amd import tiles
    torch.manual_seed(11)
    D, T, Hs, Ws, QH, QW = 3, 4, 37, 53, 5, 7
    keep = (torch.rand(D, QH, QW) < 0.5).to(dev)
    kt = tiles.quad_to_texel_mask(keep, Hs, Ws)[:, None, :, :, None].float()
    p0 = torch.randn(D, T, Hs, Ws, 4, device=dev)
    pa, pb = p0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    oa = torch.optim.Adam([pa], lr=0.05, betas=(0.9, 0.999), eps=6e-8)
    ob = tiles.TileAdam([pb], lr=0.05, betas=(0.9, 0.999), eps=6e-8, quad_keep=keep)
    for it in range(5):
        g = torch.randn_like(p0) * kt * (0.1 + it)
        for o, p in ((oa, pa), (ob, pb)):
            o.param_groups[0]["lr"] = 0.05 / (1 + it)
            p.grad = g.clone()
            o.step()
        assert float((pa - pb).abs().max()) <= 2e-6
    assert torch.equal(pb.detach() * (1 - kt), p0 * (1 - kt))
    dense = tiles.TileAdam([p0.clone().requires_grad_(True)], lr=0.01, eps=6e-8)       # quad_keep None: every texel
    dense.param_groups[0]["params"][0].grad = torch.ones_like(p0)
    dense.step()
    assert float((dense.param_groups[0]["params"][0] - (p0 - 0.01)).abs().max()) <= 1e-6


def test_one_pass_adam_skips_untouched_texels_exactly(dev):
    """adam_tiles_k neither reads the parameter nor writes anything where g = m = v = 0 (texels no view has reached: the margins of planes
    stored larger than the frame): the same bits as torch.optim.Adam there (whose update of such a texel is p - 0) and the usual 2e-6 everywhere else, step after step -- a texel that receives its
    first gradient later joins in, one whose gradient returns to 0 keeps decaying its moments."""
    from videoloop3d_amd import tiles
    torch.manual_seed(5)
    D, T, Hs, Ws = 3, 1, 37, 53
    p0 = torch.randn(D, T, Hs, Ws, 4, device=dev)
    pa, pb = p0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    oa = torch.optim.Adam([pa], lr=0.05, betas=(0.9, 0.999), eps=1e-8)
    ob = tiles.TileAdam([pb], lr=0.05, betas=(0.9, 0.999), eps=1e-8)
    yy, xx = torch.meshgrid(torch.arange(Hs, device=dev), torch.arange(Ws, device=dev), indexing="ij")
    never = ((yy < 6) | (xx > 44))[None, None, :, :, None]                      # no gradient in any step
    for it in range(6):
        seen = ((yy + xx) % 7 < 2 + it)[None, None, :, :, None] & ~never        # grows: first gradients arrive in later steps; some -0.0 too
        g = torch.randn_like(p0) * seen
        g = torch.where((g == 0) & (torch.rand_like(g) < 0.3), -torch.zeros_like(g), g)
        if it == 4:
            g = g * 0                                                           # a step without any gradient: moments decay, parameters still move
        for o, p in ((oa, pa), (ob, pb)):
            p.grad = g.clone()
            o.step()
        assert float((pa - pb).abs().max()) <= 2e-6, it
        untouched = (ob.state[pb]["exp_avg"] == 0) & (ob.state[pb]["exp_avg_sq"] == 0)
        assert torch.equal(pa.detach()[untouched], pb.detach()[untouched])      # (torch's own update of a texel with g = m = v = 0 is p - 0)
    assert torch.equal((pb.detach() * never), (p0 * never)) and torch.equal((pa.detach() * never), (p0 * never))
    sa, sb = oa.state[pa], ob.state[pb]
    assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 1e-6 and float((sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max()) <= 1e-6
    assert float(sb["exp_avg"].abs().max()) > 0 and not bool((sb["exp_avg"] * never).any()) and not bool((sb["exp_avg_sq"] * never).any())


def test_tile_adam_static_texels_are_one_parameter(dev):
    """TileAdam with quad_dyn: a static texel is updated once from frame 0 (frame-summed gradient left there by
    tie_static_grad_hip(frame0_only=True)) and the value is written to all T copies == torch.optim.Adam on the fully tied gradient."""
    from videoloop3d_amd import tiles
    torch.manual_seed(12)
    D, T, Hs, Ws, QH, QW = 3, 5, 37, 53, 5, 7
    keep = (torch.rand(D, QH, QW) < 0.6).to(dev)
    dyn = keep & (torch.rand(D, QH, QW) < 0.5).to(dev)
    kt = tiles.quad_to_texel_mask(keep, Hs, Ws)[:, None, :, :, None].float()
    p0 = torch.randn(D, 1, Hs, Ws, 4, device=dev).expand(D, T, Hs, Ws, 4).contiguous()      # all copies equal at the start
    pa, pb = p0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    oa = torch.optim.Adam([pa], lr=0.05, betas=(0.9, 0.999), eps=6e-8)
    ob = tiles.TileAdam([pb], lr=0.05, betas=(0.9, 0.999), eps=6e-8, quad_keep=keep, quad_dyn=dyn)
    for it in range(4):
        g = torch.randn_like(p0) * kt
        pa.grad = tiles.tie_static_grad(g.clone(), keep, dyn)                                  # the rule, fully tied
        pb.grad = tiles.tie_static_grad_hip(g.clone(), keep, dyn, assume_culled_zero=True, frame0_only=True)
        oa.step()
        ob.step()
        assert float((pa - pb).abs().max()) <= 2e-6
    static_t = (tiles.quad_to_texel_mask(keep, Hs, Ws) & ~tiles.quad_to_texel_mask(dyn, Hs, Ws))[:, None, :, :, None].expand_as(p0)
    assert torch.equal(pb.detach()[:, :1].expand_as(p0)[static_t], pb.detach()[static_t])    # the copies stayed identical


def _sparsified_pair(dev, frames=5):
    """a sparsified stage-1 MPI handed to two stage-2 models: the dense one and one to be packed"""
    from videoloop3d_amd.MPI import MPMesh
    from videoloop3d_amd.MPV import MPMeshVid
    H, W = 44, 60
    K, ref_extrin, tar = scene(H, W)
    a1 = make_args_mpi(learn_loop_mask=True, mpi_h_verts=5, mpi_w_verts=7, sparsify_rmfirstlayer=0)
    mpi = MPMesh(a1, H, W, ref_extrin, K, 1.0, 100.0).to(dev)
    with torch.no_grad():
        mpi.stack.copy_(synth.make_plane_stack(*mpi.stack.shape[:4], seed=7) * 0.5)
        mpi.stack[..., 3] = -6.0
        mpi.stack[2, 0, 6:30, 8:50, 3] = 2.0
        mpi.stack[4, 0, 10:40, 20:60, 3] = 1.0
        mpi.stack[1, 0, 30:46, 4:30, 3] = 0.5
        mpi.stack_mask[4, 0, 15:30, 28:50] = 4.0
    mpi.sparsify_faces(erode_num=1)
    a2 = make_args(mpv_frm_num=frames, mpi_d=a1.mpi_d, mpi_h_scale=a1.mpi_h_scale, mpi_w_scale=a1.mpi_w_scale, lrate=0.05, lrate_decay=30,
                   optimizer="adam", optimize_verts_gain=1, mpi_h_verts=5, mpi_w_verts=7)
    sd = mpi.state_dict()
    dense = MPMeshVid(a2, H, W, ref_extrin, K, 1.0, 100.0).to(dev)
    dense.init_from_mpi(sd)
    packed = MPMeshVid(a2, H, W, ref_extrin, K, 1.0, 100.0).to(dev)
    packed.init_from_mpi({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sd.items()}, packed=True)      # the checkpoint stays on the host
    return dense, packed, (H, W, K, tar)


def test_packed_model_trains_and_renders_like_the_dense_one(dev):
    """§8f-2: static blocks stored once, dynamic blocks per frame, culled blocks not at all (MPI.py:364-436, MPV.py:235-288).  A packed
    model has the dense + quad-map model's BITS: evaluation images (whole clip, frame subsets), losses of training iterations over
    shifting crops and poses, and the parameters afterwards -- while its texture memory is a fraction of the dense stack's."""
    from videoloop3d_amd import tiles
    dense, packed, (H, W, K, tar) = _sparsified_pair(dev)
    assert packed.packed is not None and "stack" not in dict(packed.named_parameters())
    assert packed.packed.pool_bytes < 0.6 * packed.packed.dense_bytes
    D, T, Hs, Ws = dense.stack.shape[:4]
    tar_e, tar_k = torch.tensor(tar)[None].to(dev), torch.tensor(K)[None].to(dev)
    for m in (dense, packed):
        m.eval()
    with torch.no_grad():
        # evaluation renders of the packed model read the POOL (vl3d_render_fwd_packed: block table -> 16-byte texel per tap), never an
        # unpacked stack -- and have the dense culled render's bits
        unpack = packed.packed.unpack_frames
        packed.packed.unpack_frames = lambda *a, **k: (_ for _ in ()).throw(AssertionError("an evaluation render unpacked the pool"))
        for ts in (None, torch.tensor([3]), torch.tensor([4, 0, 2])):
            ra, _ = dense(H, W, tar_e, tar_k, ts=ts)
            rb, _ = packed(H, W, tar_e, tar_k, ts=ts)
            assert torch.equal(ra, rb)
        h_, w_ = 24, 32                                         # a crop view (shifted principal point)
        Kc = K.copy(); Kc[0, 2] -= 13; Kc[1, 2] -= 7
        assert torch.equal(dense(h_, w_, tar_e, torch.tensor(Kc)[None].to(dev), ts=torch.tensor([1, 1, 4]))[0],
                           packed(h_, w_, tar_e, torch.tensor(Kc)[None].to(dev), ts=torch.tensor([1, 1, 4]))[0])
        packed.packed.unpack_frames = unpack
        # the one-launch unpack of chosen frames (vl3d_packed_unpack_frames) == the plane-by-plane torch path
        lay, pool = packed.packed, packed.stack_pool.data
        assert torch.equal(lay.unpack_frames(pool, [4, 0, 2]), torch.stack([lay.unpack_plane(pool, d, [4, 0, 2]) for d in range(D)], 0))
    res = synth.make_video(9, 24, 32, seed=31, device=dev)[0].permute(1, 0, 2, 3)[None].contiguous()      # [1,F,3,h,w]
    cfg = collate({"loss_name": "gpnn_lm", "patch_size": 3, "patcht_size": 3, "stride": 2, "stridet": 1, "alpha": 10000.0,
                   "rou": "-2", "scaling": 0.1, "macro_block": 65})
    opts = []
    for m in (dense, packed):
        m.train()
        opts.append(m.get_optimizer(0))
    for it in range(7):
        Kc = K.copy()
        Kc[0, 2] -= 4 + (it % 3) * 9
        Kc[1, 2] -= 3 + (it % 2) * 11
        te = tar.copy()
        te[:3, 3] += [0.01 * (it % 4), -0.005 * (it % 3), 0.0]
        losses = []
        for m, opt in zip((dense, packed), opts):
            _, extra = m(24, 32, torch.tensor(te)[None], torch.tensor(Kc)[None], res=res, losscfg=dict(cfg))
            loss = extra["swd"].mean() + 0.2 * extra["rgb_smooth"].mean() + 0.2 * extra["a_smooth"].mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        assert losses[0] == losses[1]
    now = dense.state_dict()["stack"]
    sd_p = packed.state_dict()
    assert "stack" not in sd_p and "stack_pool" in sd_p
    keep_t = tiles.quad_to_texel_mask(dense.quad_keep, Hs, Ws)
    for d in range(D):
        pl = packed.stack_plane(d)
        m = keep_t[d][None, :, :, None].expand_as(pl)
        assert torch.equal(pl[m], now[d][m])
    # checkpoint round trip of the packed model, and lod() of both
    from videoloop3d_amd.MPV import MPMeshVid
    again = MPMeshVid(packed.args, H, W, packed.ref_extrin.cpu().numpy(), K, 1.0, 100.0).to(dev)
    again.init_from_mpi(sd_p)
    assert torch.equal(again.stack_pool.detach(), packed.stack_pool.detach()) and torch.equal(again.packed.blocks, packed.packed.blocks)
    for m in (dense, packed):
        m.lod(0.75)
        m.eval()
    with torch.no_grad():
        ra, _ = dense(H, W, tar_e, tar_k, ts=torch.tensor([1, 3]))
        rb, _ = packed(H, W, tar_e, tar_k, ts=torch.tensor([1, 3]))
    assert torch.equal(ra, rb)


def test_offline_renderer_reads_sparsified_and_packed_models_in_place(dev):
    """render_video.render_frames on a sparsified dense model (frames read where they lie, vl3d_render_fwd_frames_culled) and on its packed
    twin (the pool through the block table, frame indices uploaded once): the frames of the loop over the module's eval forward, and the two
    models agree -- a camera per frame, a fixed camera's run with a wrap-around, any chunk size."""
    from videoloop3d_amd import render_video as RV
    dense, packed, (H, W, K, tar) = _sparsified_pair(dev)
    n = 7
    ext = np.stack([tar.copy() for _ in range(n)]).astype(np.float32)
    for i in range(n):
        ext[i, 0, 3] += 0.004 * i
        ext[i, 1, 3] -= 0.003 * i
    intr = np.stack([K.astype(np.float32)] * n)
    fixed = np.stack([tar.astype(np.float32)] * n)
    for extr, ts in ((ext, [0, 1, 2, 3, 4, 0, 1]), (fixed, [3, 4, 0, 1, 2, 3, 4]), (ext, [2, 2, 4, 1, 0, 3, 3])):
        loop_d = RV.render_frames(dense, H, W, extr, intr, ts, in_place=False)
        loop_p = RV.render_frames(packed, H, W, extr, intr, ts, in_place=False)
        assert torch.equal(loop_d, loop_p) and float(loop_d.float().std()) > 1.0
        for chunk in (64, 3, 1):
            assert torch.equal(RV.render_frames(dense, H, W, extr, intr, ts, max_batch=chunk), loop_d), (ts, chunk)
            assert torch.equal(RV.render_frames(packed, H, W, extr, intr, ts, max_batch=chunk), loop_d), (ts, chunk)


def test_packed_model_exports_the_reference_layout_like_the_dense_one(dev):
    dense, packed, _ = _sparsified_pair(dev, frames=3)
    a, b = dense.reference_state_dict(), packed.reference_state_dict()
    assert set(a) == set(b)
    for k in a:
        if torch.is_tensor(a[k]):
            assert torch.equal(a[k].cpu(), b[k].cpu()), k
        else:
            assert a[k] == b[k], k


@pytest.mark.parametrize("sparse", [False, True])
def test_stage1_crop_aware_optimiser_equals_the_whole_stack_adam(dev, sparse):
    """MPMesh.get_optimizer() with args.crop_aware_adam -> optim.Stage1Adam: the crop-aware engine for the plane stack (compact window leaf,
    deferred zero-gradient updates replayed exactly) + the one-pass Adam for the loop-mask texture, against the default one pass over the
    whole stack (tiles.TileAdam) on copies of one model over shuffled crops of shifting views: the same losses along the way, the same
    parameters after the flush state_dict() does.  sparse: the model after sparsify_faces (train_3d.py:282-286; quad map, no loop mask) --
    its step is taken inside the render's backward (vl3d_render_bwd_adam, T = 1), bit-identical to the window path's two kernels."""
    from videoloop3d_amd.MPI import MPMesh
    from videoloop3d_amd.optim import Stage1Adam
    H, W, h, w = 66, 80, 30, 40            # (planes 85 x 104: TileAdam walks the loop-mask texture in 16-byte groups)
    K, ref_extrin, tar = scene(H, W)
    variants = [dict(), dict(crop_aware_adam=True), dict(crop_aware_adam=True, fused_adam_backward=False)]
    torch.manual_seed(3)
    proto = MPMesh(make_args_mpi(mpi_h_verts=7, mpi_w_verts=9, optimizer="adam", lrate=0.02, lrate_decay=100), H, W, ref_extrin, K, 1.0, 100.0).to(dev)
    with torch.no_grad():
        D_, _, Hs_, Ws_, _ = proto.stack.shape
        st = synth.make_plane_stack(D_, 1, Hs_, Ws_, seed=5) * 0.7
        yy, xx = torch.meshgrid(torch.arange(Hs_).float(), torch.arange(Ws_).float(), indexing="ij")
        for d in range(D_):          # an alpha blob per plane: part of every plane is culled by sparsify_faces, plane 5 entirely
            blob = 6.0 * torch.exp(-(((yy - Hs_ * (0.3 + 0.1 * d)) / 16) ** 2 + ((xx - Ws_ * (0.2 + 0.12 * d)) / 22) ** 2)) - 4.5
            st[d, 0, :, :, 3] = blob if d < 5 else -8.0
        proto.stack.copy_(st.to(dev))
        proto.stack_mask.copy_((synth.hash_uniform(tuple(proto.stack_mask.shape), seed=6) * 3 - 2).to(dev))
    if sparse:
        proto.sparsify_faces(erode_num=1, alpha_thresh=0.05)
        assert 0.1 < float(proto.quad_keep.float().mean()) < 0.9 and not bool(proto.quad_keep[5].any())
    sd = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in proto.state_dict().items()}
    models = []
    for kw in variants:
        m = MPMesh(make_args_mpi(mpi_h_verts=7, mpi_w_verts=9, optimizer="adam", lrate=0.02, lrate_decay=100, **kw), H, W, ref_extrin, K, 1.0, 100.0).to(dev)
        m.init_from_mpi(sd)
        m.train()
        models.append((m, m.get_optimizer()))
    assert type(models[0][1]).__name__ == "TileAdam" and isinstance(models[1][1], Stage1Adam) and isinstance(models[2][1], Stage1Adam)
    assert models[1][1].window.fused_backward and not models[2][1].window.fused_backward
    wts = {"sparsity": 0.004, "rgb_smooth": 0.2, "a_smooth": 0.5, "density": 0.02}
    offs = [(0, 0), (30, 35), (10, 20), (36, 40), (0, 40), (36, 0), (18, 25), (0, 0), (36, 40)]
    for it, (oy, ox) in enumerate(offs):
        Kc = K.copy()
        Kc[0, 2] -= ox
        Kc[1, 2] -= oy
        te = tar.copy()
        te[:3, 3] += [0.01 * (it % 3), -0.005 * (it % 2), 0.0]
        g = (synth.hash_uniform((1, 3 if sparse else 4, h, w), seed=40 + it) - 0.5).to(dev)
        losses = []
        for m, opt in models:
            for grp in opt.param_groups:
                grp["lr"] = 0.02 * 0.9 ** it
            opt.zero_grad()
            rgbl, extra = m(h, w, torch.tensor(te)[None], torch.tensor(Kc)[None])       # poses on the host, as the DataLoader yields them
            loss = (rgbl * g).sum() + sum(extra[k].sum() * wts[k] for k in wts) * 20
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        for l in losses[1:]:
            assert abs(l - losses[0]) <= 2e-5 * max(1.0, abs(losses[0])), (it, losses)
        if sparse:       # fused and two-kernel window paths: the same bits, every iteration
            assert losses[1] == losses[2]
            assert torch.equal(models[1][0].stack.data, models[2][0].stack.data), it
    if sparse:
        assert models[1][1].window.fused_steps == len(offs) and models[2][1].window.fused_steps == 0
    assert float((models[1][0].stack.detach() - models[0][0].stack.detach()).abs().max()) > 1e-4          # updates really are deferred ...
    sds = [m.state_dict() for m, _ in models]                                                              # ... until the flush
    keep = slice(None)
    if sparse:
        from videoloop3d_amd import tiles
        keep = tiles.quad_to_texel_mask(proto.quad_keep.cpu(), *proto.stack.shape[2:4]).to(dev)[:, None, :, :, None].expand_as(proto.stack)
    for sd_b in sds[1:]:
        diff = (sds[0]["stack"] - sd_b["stack"])[keep].abs()
        assert float((diff > 2e-5).float().mean()) <= 1e-3 and float(diff.max()) <= 2e-3 and float(diff.mean()) <= 1e-6
        if not sparse:
            assert float((sds[0]["stack_mask"] - sd_b["stack_mask"]).abs().max()) <= 2e-5
    # evaluation renders read the whole (current) stack
    for m, _ in models:
        m.eval()
    with torch.no_grad():
        imgs = [m(H, W, torch.tensor(tar)[None].to(dev), torch.tensor(K)[None].to(dev))[0] for m, _ in models]
    assert float((imgs[0] - imgs[1]).abs().max()) <= 2e-4
