"""The HIP modules against what the reference's OWN MPI.py / MPV.py computed (golden G17, tests/golden/make_golden_r04.py: the reference's
render / forward arithmetic downstream of a harness rasteriser) on IDENTICAL weights:
  * sparsified checkpoints (G15 / G16) are read onto the tile lattice texel for texel (tiles.stack_from_reference_state) and rendered with
    their quad maps: MPMesh.forward (MPI.py:544-548, 596-652) and MPMeshVid.forward (MPV.py:389-556), values and gradients;
  * dense MPMeshVid weights go through atlas_to_stack + atlas_exact (the reference's cell pitch and neighbour-cell bleed).
Tolerance 1e-4 max-abs (BASELINE.json north_star) on images, relative on sums / gradients."""
import numpy as np
import pytest
import torch

import refmod as RM
from videoloop3d_amd import tiles

pytestmark = pytest.mark.gpu
R4 = RM.R4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import __graft_entry__ as g
    g.build()
    return torch.device("cuda:0")


def _close(a, b, tol, what):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    err = float((a - b).abs().max())
    assert err <= tol, (what, err)


def _rel(a, b, what, rel=1e-4):
    _close(a, b, rel * max(1.0, float(torch.as_tensor(b).abs().max())), what)


def test_sparsified_mpmesh_forward_matches_the_reference(dev):
    from videoloop3d_amd.MPI import MPMesh
    g15, g = RM.load("g15_sparsify"), RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    args = R4.make_args(learn_loop_mask=True, **over, **RM.REG)
    h, w, tar_e, K_crop, K_full = RM.crop_view(g)
    sd = RM.state_dict_of(g15, "sd_")
    m = MPMesh(args, H, W, ref_extrin, K, 1.0, 100.0)
    m.init_from_mpi(sd, tile_layout="lattice")
    m = m.to(dev)
    assert m.is_sparse and m.stack.shape[2:4] == (37, 55) and not m.learn_loop_mask
    m.train()
    rgbl, extra = m(h, w, tar_e, K_crop)
    assert rgbl.shape == (1, 3, h, w)
    _close(rgbl, g["b_rgb"], 1e-4, "rgb")
    for k in ("sparsity", "rgb_smooth", "a_smooth", "density"):
        _rel(extra[k], g[f"b_extra_{k}"], k)
    total = (rgbl * torch.from_numpy(g["b_G"]).to(dev)).sum() + sum(getattr(args, k + "_loss_weight") * v.sum() for k, v in extra.items())
    (gs,) = torch.autograd.grad(total, m.stack)
    g_dyn, g_static = RM.lattice_grad_from_reference(sd, over["mpi_h_verts"], over["mpi_w_verts"], over["mpi_d"], 1,
                                                     torch.from_numpy(g["b_grad_atlas"]), torch.from_numpy(g["b_grad_atlas_dyn"]))
    _rel(gs[:, 0], g_dyn[:, 0] + g_static, "stack gradient")
    m.eval()
    with torch.no_grad():
        _close(m(H, W, tar_e, K_full)[0], g["b_eval_rgb_full"], 1e-4, "eval")
    # the quad map is what makes the regularisers the reference's: the same stack rendered as a dense one composites the same image
    # (culled texels are transparent) but differences run over planes, not hit slots
    m.train()
    m.is_sparse = False
    _, extra_nq = m(h, w, tar_e, K_crop)
    assert abs(float(extra_nq["rgb_smooth"]) - float(g["b_extra_rgb_smooth"].item())) > 1e-3


def test_trained_tile_checkpoint_stage1_matches_the_reference(dev):
    """golden G19 (j): the stage-1 model AFTER sparsify_faces with both atlases perturbed everywhere (train_3d.py:282-285 trains it for the last
    epochs: the two copies of a border sample drift apart) -- MPMesh in the tile-exact layout against the reference's own forward: image,
    regulariser terms, the gradient of every tile texel, evaluation render; one crop-aware training step moves kept tiles only."""
    from videoloop3d_amd.MPI import MPMesh
    g = RM.load("g19_trained_tiles")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    hv, wv, D = over["mpi_h_verts"], over["mpi_w_verts"], over["mpi_d"]
    args = R4.make_args(learn_loop_mask=True, **over, **RM.REG)
    h, w, tar_e, K_crop, K_full = RM.crop_view(g)
    sd = RM.state_dict_of(g, "j_sd_")
    m = MPMesh(args, H, W, ref_extrin, K, 1.0, 100.0)
    m.init_from_mpi(sd)
    m = m.to(dev)
    assert m.is_sparse and m.tile_own == (10, 10) and m.stack.shape[2:4] == (40, 60) and m.spec.tile == (10, 10)
    m.train()
    rgbl, extra = m(h, w, tar_e, K_crop)
    _close(rgbl, g["j_rgb"], 1e-4, "rgb")
    for k in ("sparsity", "rgb_smooth", "a_smooth", "density"):
        _rel(extra[k], g[f"j_extra_{k}"], k)
    total = (rgbl * torch.from_numpy(g["j_G"]).to(dev)).sum() + sum(getattr(args, k + "_loss_weight") * v.sum() for k, v in extra.items())
    (gs,) = torch.autograd.grad(total, m.stack)
    g_dyn, g_static = RM.own_grad_from_reference(sd, hv, wv, D, 1, torch.from_numpy(g["j_grad_atlas"]), torch.from_numpy(g["j_grad_atlas_dyn"]))
    _rel(gs[:, 0], g_dyn[:, 0] + g_static, "gradient of every tile texel")
    m.eval()
    with torch.no_grad():
        _close(m(H, W, tar_e, K_full)[0], g["j_eval_rgb_full"], 1e-4, "eval")
    # the checkpoint goes back out bit for bit (tiles a face references)
    out = m.reference_state_dict()
    for key, faces, gw_ in (("atlas", "faces", "self.atlas_grid_w"), ("atlas_dyn", "faces_dyn", "self.atlas_grid_dyn_w")):
        for k_ in range(sd[faces].shape[0] // 2):
            ys, xs = slice((k_ // sd[gw_]) * 10, (k_ // sd[gw_]) * 10 + 10), slice((k_ % sd[gw_]) * 10, (k_ % sd[gw_]) * 10 + 10)
            assert torch.equal(out[key][..., ys, xs].cpu(), sd[key][..., ys, xs]), (key, k_)
    # a training step through the crop-aware engine (window copy of the tile planes, the step inside the backward)
    m.train()
    m.args.crop_aware_adam = True
    opt = m.get_optimizer()
    from videoloop3d_amd.optim import Stage1Adam
    assert isinstance(opt, Stage1Adam) and opt.window.tile == (10, 10)
    opt.acknowledge_fused_backward()
    before = m.stack.detach().clone()
    rgbl, extra = m(h, w, tar_e, K_crop)
    ((rgbl * torch.from_numpy(g["j_G"]).to(dev)).sum() + sum(getattr(args, k + "_loss_weight") * v.sum() for k, v in extra.items())).backward()
    opt.step()
    opt.flush()
    kept = tiles.quad_to_texel_mask(m.quad_keep, 40, 60, (10, 10))
    moved = (m.stack.detach() - before).abs().amax((1, 4))
    assert float(moved[kept].max()) > 0 and float(moved[~kept].max()) == 0.0


def test_sparsified_mpmeshvid_forward_matches_the_reference(dev):
    from videoloop3d_amd.MPV import MPMeshVid
    g15, g = RM.load("g15_sparsify"), RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    args = RM.mpv_args(5)
    h, w, tar_e, K_crop, K_full = RM.crop_view(g)
    res = torch.from_numpy(g["res"]).to(dev)
    sd = RM.state_dict_of(g15, "sd_", atlas_dyn=torch.from_numpy(g["d_atlas_dyn"]))          # the stage-2 model of G17 (d): 5 dynamic frames
    v = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0)
    v.init_from_mpi(sd, tile_layout="lattice")      # (the shared-border representation of rounds 4-5: this checkpoint's border copies agree)
    v = v.to(dev)
    v._install_tie_hook()
    assert v.frm_num == 5 and v.is_sparse and v.tile_full == (10, 10) and v.tile_own is None
    v.train()
    _, extra = v(h, w, tar_e, K_crop, res=res, losscfg=R4.collate(RM.LOSS_CFGS["other"]))
    for k in ("swd", "sparsity", "rgb_smooth", "a_smooth", "density"):
        _rel(extra[k], g[f"d_extra_{k}"], k)
    total = sum(RM.MPV_WEIGHTS[k] * x.sum() for k, x in extra.items())
    total.backward()
    gs = v.stack.grad.cpu()
    g_dyn, g_static = RM.lattice_grad_from_reference(sd, over["mpi_h_verts"], over["mpi_w_verts"], over["mpi_d"], 5,
                                                     torch.from_numpy(g["d_grad_atlas"]), torch.from_numpy(g["d_grad_atlas_dyn"]))
    want_sum = g_dyn.sum(1) + g_static
    Hs, Ws = gs.shape[2:4]
    dyn_t = tiles.quad_to_texel_mask(v.quad_dyn.cpu(), Hs, Ws)
    keep_t = tiles.quad_to_texel_mask(v.quad_keep.cpu(), Hs, Ws)
    static_t = keep_t & ~dyn_t
    scale = max(1.0, float(want_sum.abs().max()))
    # static texels are ONE texture (MPV.py:389-392: the static atlas has one frame): every frame's copy carries the frame sum
    for t in (0, 3):
        _close(gs[:, t][static_t], want_sum[static_t], 1e-4 * scale, f"static texels, frame {t}")
    _close(gs.sum(1)[dyn_t], want_sum[dyn_t], 1e-4 * scale, "dynamic texels, frame sum")
    pure = dyn_t & (g_static.abs().sum(-1) == 0)
    assert int(pure.sum()) > 1000
    _close(gs[pure[:, None].expand(-1, 5, -1, -1)], g_dyn[pure[:, None].expand(-1, 5, -1, -1)], 1e-4 * scale, "dynamic texels per frame")
    assert float(gs[~keep_t[:, None].expand(-1, 5, -1, -1)].abs().max()) == 0.0
    v.eval()
    with torch.no_grad():
        ev = v(H, W, tar_e, K_full)[0]
        _close(ev, g["d_eval_rgb_full"], 1e-4, "eval")
    # the path a training run takes -- the crop-aware optimiser's window copy, culled kernels on the window -- computes the reference's terms too
    v.zero_grad(set_to_none=True)
    v.train()
    opt = v.get_optimizer(0)
    _, extra_w = v(h, w, tar_e, K_crop, res=res, losscfg=R4.collate(RM.LOSS_CFGS["other"]))
    for k in ("swd", "sparsity", "rgb_smooth", "a_smooth", "density"):
        _rel(extra_w[k], g[f"d_extra_{k}"], k + " (window path)")
    sum(RM.MPV_WEIGHTS[k] * x.sum() for k, x in extra_w.items()).backward()
    opt.step()
    # ... and the PACKED form of the same checkpoint (static blocks once, dynamic blocks per frame, culled blocks nowhere) renders the reference's
    # image straight from its pool (vl3d_render_fwd_packed)
    v2 = MPMeshVid(RM.mpv_args(5), H, W, ref_extrin, K, 1.0, 100.0)
    v2.init_from_mpi(sd, tile_layout="lattice")
    v2 = v2.to(dev)
    v2.pack_()
    v2.eval()
    assert v2.packed is not None and v2.packed.pool_bytes < 0.8 * v2.packed.dense_bytes
    with torch.no_grad():
        ev2 = v2(H, W, tar_e, K_full)[0]
    assert torch.equal(ev2, ev)
    with torch.no_grad():
        sub = v2(h, w, tar_e, K_crop, ts=torch.tensor([4, 2]))[0]
    assert torch.equal(sub, _eval_frames(sd, dev, H, W, ref_extrin, K, h, w, tar_e, K_crop))


def _eval_frames(sd, dev, H, W, ref_extrin, K, h, w, tar_e, K_crop):
    """the dense model of the same checkpoint rendering frames [4, 2] of the crop view (what the packed model must reproduce bit for bit)."""
    from videoloop3d_amd.MPV import MPMeshVid
    d = MPMeshVid(RM.mpv_args(5), H, W, ref_extrin, K, 1.0, 100.0)
    d.init_from_mpi(sd, tile_layout="lattice")
    d = d.to(dev).eval()
    with torch.no_grad():
        return d(h, w, tar_e, K_crop, ts=torch.tensor([4, 2]))[0]


def test_trained_tile_checkpoint_matches_the_reference(dev):
    """Golden G19 (tests/golden/make_golden_r06.py): a TRAINED stage-2 checkpoint -- every tile texel perturbed independently, so the two copies
    of every border sample that neighbouring tiles hold differ (static next to dynamic included).  The tile-exact layout (init_from_mpi's
    default) holds it exactly and the HIP module reproduces the reference's own forward: every `extra` term, the gradient of EVERY TILE TEXEL of
    both atlases, evaluation renders -- through the dense stack, the crop-aware optimiser's window path and the packed pool.  The shared-border
    lattice of rounds 4-5 misses the same checkpoint's image by ~0.27."""
    from videoloop3d_amd.MPV import MPMeshVid
    g = RM.load("g19_trained_tiles")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    hv, wv, D = over["mpi_h_verts"], over["mpi_w_verts"], over["mpi_d"]
    h, w, tar_e, K_crop, K_full = RM.crop_view(g)
    res = torch.from_numpy(g["res"]).to(dev)
    sd = RM.state_dict_of(g, "h_sd_")
    v = MPMeshVid(RM.mpv_args(5), H, W, ref_extrin, K, 1.0, 100.0)
    v.init_from_mpi(sd)
    v = v.to(dev)
    v._install_tie_hook()
    assert v.tile_own == (10, 10) and v.stack.shape == (D, 5, 40, 60, 4) and v.spec.tile == (10, 10)
    v.train()
    _, extra = v(h, w, tar_e, K_crop, res=res, losscfg=R4.collate(RM.LOSS_CFGS["other"]))
    for k in ("swd", "sparsity", "rgb_smooth", "a_smooth", "density"):
        _rel(extra[k], g[f"h_extra_{k}"], k)
    sum(RM.MPV_WEIGHTS[k] * x.sum() for k, x in extra.items()).backward()
    gs = v.stack.grad.cpu()
    g_dyn, g_static = RM.own_grad_from_reference(sd, hv, wv, D, 5, torch.from_numpy(g["h_grad_atlas"]), torch.from_numpy(g["h_grad_atlas_dyn"]))
    tile = (10, 10)
    dyn_t = tiles.quad_to_texel_mask(v.quad_dyn.cpu(), 40, 60, tile)
    static_t = tiles.quad_to_texel_mask((v.quad_keep & ~v.quad_dyn).cpu(), 40, 60, tile)
    scale = float(g_dyn.abs().max())
    assert int(dyn_t.sum()) == 80 * 100 and int(static_t.sum()) == 16 * 100 and scale > 1e-5
    sel = dyn_t[:, None].expand(-1, 5, -1, -1)
    _close(gs[sel], g_dyn[sel], 1e-4 * scale, "every texel of every dynamic tile, per frame")
    for t in (0, 2, 4):      # a static tile is ONE texture (MPV.py:389-392): every frame's copy carries the reference's gradient of that one tile
        _close(gs[:, t][static_t], g_static[static_t], 1e-4 * scale, f"static tiles, frame {t}")
    assert float(gs[(~(dyn_t | static_t))[:, None].expand(-1, 5, -1, -1)].abs().max()) == 0.0
    v.eval()
    with torch.no_grad():
        ev = v(H, W, tar_e, K_full)[0]
        _close(ev, g["h_eval_rgb_full"], 1e-4, "eval, full frame")
        _close(v(h, w, tar_e, K_crop, ts=torch.tensor([3, 1]))[0], g["h_eval_rgb_crop_ts"], 1e-4, "eval, frames [3, 1] of the crop")
    # the shared-border reader on the same checkpoint (what rounds 4-5 rendered): wrong by the size of the border drift
    vl = MPMeshVid(RM.mpv_args(5), H, W, ref_extrin, K, 1.0, 100.0)
    vl.init_from_mpi(sd, tile_layout="lattice")
    vl = vl.to(dev).eval()
    with torch.no_grad():
        err_lattice = float((vl(H, W, tar_e, K_full)[0].cpu() - torch.from_numpy(g["h_eval_rgb_full"])).abs().max())
    assert err_lattice > 0.05, err_lattice
    # the path a training run takes: the crop-aware optimiser's compact window copy, culled kernels on a WINDOW of the tile planes.  Two-kernel
    # step (fused_adam_backward off) so that the window leaf's gradient can be read: the reference's per-tile-texel gradients again
    v.zero_grad(set_to_none=True)
    v.train()
    v.args.fused_adam_backward = False
    opt = v.get_optimizer(0)
    _, extra_w = v(h, w, tar_e, K_crop, res=res, losscfg=R4.collate(RM.LOSS_CFGS["other"]))
    for k in ("swd", "sparsity", "rgb_smooth", "a_smooth", "density"):
        _rel(extra_w[k], g[f"h_extra_{k}"], k + " (window path)")
    sum(RM.MPV_WEIGHTS[k] * x.sum() for k, x in extra_w.items()).backward()
    (y0, x0, wh, ww), leaf, _ = opt.pending
    gw = leaf.grad.cpu()
    selw = dyn_t[:, None, y0:y0 + wh, x0:x0 + ww].expand(-1, 5, -1, -1)
    _close(gw[selw], g_dyn[:, :, y0:y0 + wh, x0:x0 + ww][selw], 1e-4 * scale, "window leaf gradient, dynamic tiles")
    _close(gw.sum(1)[static_t[:, y0:y0 + wh, x0:x0 + ww]], g_static[:, y0:y0 + wh, x0:x0 + ww][static_t[:, y0:y0 + wh, x0:x0 + ww]], 1e-4 * scale,
           "window leaf gradient, static tiles (summed over the frames by the step)")
    outside = torch.ones((40, 60), dtype=torch.bool)
    outside[y0:y0 + wh, x0:x0 + ww] = False
    assert float(g_dyn[:, :, outside].abs().max()) == 0.0 and float(g_static[:, outside].abs().max()) == 0.0      # the window holds every texel the crop reaches
    before = v.stack.detach().clone()
    opt.step()
    opt.flush()
    moved = (v.stack.detach() - before).abs().amax((1, 4)).cpu()
    assert float(moved[dyn_t | static_t].max()) > 0 and float(moved[~(dyn_t | static_t)].max()) == 0.0
    d0, y_, x_ = static_t.nonzero()[0].tolist()      # (static tiles stay one texture after the step)
    assert bool((v.stack.detach()[d0, :, y_, x_] == v.stack.detach()[d0, :1, y_, x_]).all())
    # the fused step (the default: the backward applies Adam where it would have stored the gradient) leaves the same parameters as the two kernels
    def one_step(fused, packed):
        m = MPMeshVid(RM.mpv_args(5), H, W, ref_extrin, K, 1.0, 100.0)
        m.init_from_mpi(sd, packed=packed) if packed else m.init_from_mpi(sd)
        m = m.to(dev)
        m.train()
        m.args.fused_adam_backward = fused
        o = m.get_optimizer(0)
        if fused:
            o.acknowledge_fused_backward()
        _, ex = m(h, w, tar_e, K_crop, res=res, losscfg=R4.collate(RM.LOSS_CFGS["other"]))
        sum(RM.MPV_WEIGHTS[k] * x.sum() for k, x in ex.items()).backward()
        o.step()
        o.flush()
        return m
    a, b, c = one_step(False, False), one_step(True, False), one_step(True, True)
    assert b._window_opt.fused_steps == 1 and c.packed is not None
    assert torch.equal(a.stack.detach(), v.stack.detach()) and torch.equal(a.stack.detach(), b.stack.detach())
    for d_ in range(D):
        assert torch.equal(c.stack_plane(d_)[:, (dyn_t | static_t)[d_]], a.stack.detach()[d_][:, (dyn_t | static_t)[d_]]), d_
    # ... and the PACKED pool of the checkpoint renders the reference's image straight from its blocks (vl3d_render_fwd_packed)
    v2 = MPMeshVid(RM.mpv_args(5), H, W, ref_extrin, K, 1.0, 100.0)
    v2.init_from_mpi(sd, packed=True)
    v2 = v2.to(dev).eval()
    assert v2.packed is not None and v2.packed.tile == (10, 10)
    with torch.no_grad():
        assert torch.equal(v2(H, W, tar_e, K_full)[0], ev)
    # the offline renderer (scripts/script_render_video.py loads the LAST checkpoint -- a trained one -- and renders frame by frame): its in-place
    # frame path on the tile-exact dense stack and on the packed pool gives the module's own evaluation frames
    from videoloop3d_amd import render_video as RV
    ve, vi = tar_e.repeat(4, 1, 1).numpy(), K_full.repeat(4, 1, 1).numpy()
    rt = np.array([0, 1, 2, 4])
    want8 = RV.to8b(ev[rt].permute(0, 2, 3, 1))
    v.eval()
    vfresh = MPMeshVid(RM.mpv_args(5), H, W, ref_extrin, K, 1.0, 100.0)
    vfresh.init_from_mpi(sd)
    vfresh = vfresh.to(dev).eval()
    for model in (vfresh, v2):
        got8 = RV.render_frames(model, H, W, ve, vi, rt)
        assert int((got8.int() - want8.int()).abs().max()) <= 1      # (uint8 of the same floats: the in-place kernels' bits)
    # the trained model goes back out as the reference's checkpoint, bit for bit where a face references it
    out = v2.reference_state_dict()
    for key, faces, gw_ in (("atlas", "faces", "self.atlas_grid_w"), ("atlas_dyn", "faces_dyn", "self.atlas_grid_dyn_w")):
        for k_ in range(sd[faces].shape[0] // 2):
            ys, xs = slice((k_ // sd[gw_]) * 10, (k_ // sd[gw_]) * 10 + 10), slice((k_ % sd[gw_]) * 10, (k_ % sd[gw_]) * 10 + 10)
            assert torch.equal(out[key][..., ys, xs].cpu(), sd[key][..., ys, xs]), (key, k_)
    print(f"G19: shared-border lattice image error {err_lattice:.3f}; tile-exact layout within 1e-4")


@pytest.mark.parametrize("which", ["other", "ref", "plain"])
def test_dense_mpmeshvid_atlas_exact_matches_the_reference(dev, which):
    from videoloop3d_amd.MPV import MPMeshVid, atlas_to_stack, stack_to_atlas
    g = RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    bg = "0.2#0.4#0.6" if which == "ref" else ""
    args = RM.mpv_args(5, bg=bg, regs={})
    h, w, tar_e, K_crop, K_full = RM.crop_view(g)
    v = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0, atlas_exact=True)
    with torch.no_grad():
        v.stack.copy_(atlas_to_stack(torch.from_numpy(g["c_atlas_dyn"]), over["mpi_d"], over["atlas_grid_h"]))
    v = v.to(dev)
    v.eval()
    with torch.no_grad():
        _close(v(H, W, tar_e, K_full)[0], g[f"c_{which}_eval_rgb_full"], 1e-4, "eval")
        _close(v(h, w, tar_e, K_crop, ts=torch.tensor([3, 1]))[0], g[f"c_{which}_eval_rgb_crop_ts"], 1e-4, "eval ts")
    if which != "plain":
        return
    v.train()
    res = torch.from_numpy(g["res"]).to(dev)
    _, extra = v(h, w, tar_e, K_crop, res=res, losscfg=R4.collate(RM.LOSS_CFGS["other"]))
    assert sorted(extra) == ["swd"]
    _rel(extra["swd"], g["c_plain_extra_swd"], "swd")
    (gs,) = torch.autograd.grad(extra["swd"].sum(), v.stack)
    _rel(stack_to_atlas(gs, over["atlas_grid_h"]), g["c_plain_grad_atlas_dyn"], "grad atlas_dyn")


def test_dense_mpmeshvid_second_layout_matches_the_reference(dev):
    """golden (e): 3 x 2 cells, non-square plane scales, normalize_verts, another view -- atlas_exact on the HIP kernels."""
    from videoloop3d_amd.MPV import MPMeshVid, atlas_to_stack, stack_to_atlas
    g = RM.load("g17_forward")
    H, W, over = R4.SHAPES["B"]
    K, ref_extrin, _ = R4.scene(H, W, angle_deg=-1.7, trans=(-0.031, 0.022, -0.004))
    args = R4.make_args(mpv_frm_num=4, mpv_isloop=True, init_std=0.5, scale_invariant=True, swd_patch_size=3, swd_patcht_size=3,
                        swd_stride=2, swd_stridet=1, **over)
    h, w = (int(v) for v in g["e_hw_crop"])
    tar_e, K_crop, K_full = (torch.from_numpy(g[k]) for k in ("e_tar_extrin", "e_tar_intrin_crop", "e_tar_intrin_full"))
    v = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0, atlas_exact=True)
    with torch.no_grad():
        v.stack.copy_(atlas_to_stack(torch.from_numpy(g["e_atlas_dyn"]), over["mpi_d"], over["atlas_grid_h"]))
    v = v.to(dev)
    v.eval()
    with torch.no_grad():
        _close(v(H, W, tar_e, K_full)[0], g["e_eval_rgb_full"], 1e-4, "eval")
        _close(v(h, w, tar_e, K_crop, ts=torch.tensor([2, 0]))[0], g["e_eval_rgb_crop_ts"], 1e-4, "eval ts")
    v.train()
    _, extra = v(h, w, tar_e, K_crop, res=torch.from_numpy(g["e_res"]).to(dev), losscfg=R4.collate(RM.LOSS_CFGS["other"]))
    _rel(extra["swd"], g["e_extra_swd"], "swd")
    (gs,) = torch.autograd.grad(extra["swd"].sum(), v.stack)
    _rel(stack_to_atlas(gs, over["atlas_grid_h"]), g["e_grad_atlas_dyn"], "grad atlas_dyn")


@pytest.mark.parametrize("layout", ["exact", "lattice"])
@pytest.mark.parametrize("packed", [False, True])
def test_stage2_training_from_the_reference_checkpoint(dev, packed, layout):
    """The hand-over the reference's pipeline makes (train_3d.py sparsifies and saves; train_3dvid.py:205-211 loads it with init_from_mpi and trains
    the pyramid): the REFERENCE's stage-1 checkpoint (G15) into this package's stage-2 driver, dense and packed -- two pyramid levels on the tile
    lattice (lod follows the reference's tile sizes), the loss goes down, and the trained model exports a checkpoint with the reference's keys,
    face lists and tile geometry."""
    from videoloop3d_amd import synth, train_3dvid as drv
    from videoloop3d_amd.MPV import MPMeshVid
    g15 = RM.load("g15_sparsify")
    H, W, over, K, ref_extrin, tar = RM.case_A()
    sd = RM.state_dict_of(g15, "sd_")
    args = R4.make_args(mpv_frm_num=5, mpv_isloop=True, init_std=0.1, scale_invariant=True, swd_patch_size=3, swd_patcht_size=3, swd_stride=2,
                        swd_stridet=1, rgb_smooth_loss_weight=0.05, a_smooth_loss_weight=0.05, pyr_minimal_dim=-1, pyr_stage="2", N_iters=5,
                        pyr_factor=0.5, pyr_num_epoch=0, patch_h_size=20, patch_w_size=28, patch_h_stride=12, patch_w_stride=20, lrate=0.5,
                        lrate_adaptive=True, add_intrin_noise=True, swd_loss_weight=1.0, **over)
    model = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0).to(dev)
    model.init_from_mpi(sd, packed=packed, tile_layout=layout)
    assert (model.packed is not None) == packed and model.tile_full == (10, 10) and model.is_sparse
    assert model.tile_own == ((10, 10) if layout == "exact" else None)
    vids = [synth.make_video(8, H, W, seed=21 + v, device=dev)[0].permute(1, 0, 2, 3).contiguous() for v in range(2)]
    poses = torch.stack([torch.tensor(np.linalg.inv(ref_extrin))[:3], torch.tensor(np.linalg.inv(tar))[:3]]).float()
    intr = torch.tensor(K).float()[None].repeat(2, 1, 1)
    cfg = {"loss_name": "gpnn_lm", "patch_size": 3, "patcht_size": 3, "stride": 2, "stridet": 1, "alpha": 10000, "rou": "-2", "scaling": 0.1,
           "dist_fn": "mse", "macro_block": 65, "factor": 1}
    log = []
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        n = drv.train(model, args, vids, poses, intr, [cfg, dict(cfg, loss_gain=2.0)], H, W, device=dev,
                      on_step=lambda lvl, ep, it, loss, swd, extra: log.append((lvl, float(loss))), generator=torch.Generator().manual_seed(1))
    assert n == len(log) and all(np.isfinite(l) for _, l in log)
    # level 0: tiles of max(int(10 * 0.5), 2) = 5 texels, level 1: the checkpoint's own 10 (MPV.py:146-151)
    assert model.stack_dims()[2:4] == ((4 * 10, 6 * 10) if layout == "exact" else (4 * 9 + 1, 6 * 9 + 1))
    fine = [l for lvl, l in log if lvl == 1]
    assert np.mean(fine[-6:]) < np.mean(fine[:6])
    out = model.reference_state_dict()
    for k in ("faces", "uvfaces", "faces_dyn", "uvfaces_dyn", "uvs", "uvs_dyn"):
        assert torch.equal(out[k].cpu(), sd[k]) if out[k].dtype == torch.int64 else float((out[k].cpu() - sd[k]).abs().max()) <= 1e-6, k
    assert out["atlas"].shape == sd["atlas"].shape and out["atlas_dyn"].shape == (5,) + tuple(sd["atlas_dyn"].shape[1:])
    assert float((out["atlas_dyn"][0] - out["atlas_dyn"][3]).abs().max()) > 1e-3            # the dynamic tiles have become per-frame textures
    for k in sd:
        if k.startswith("self."):
            assert out[k] == sd[k], k


@pytest.mark.parametrize("name", ["gpnn", "mse", "avg"])
def test_other_loss_entries_match_the_reference(dev, name):
    """golden (f): MPMeshVid.forward with loss_name 'gpnn' (config_parser.py:56 default: Patch3DGPNNDirectLoss), 'mse', 'avg' on the HIP path."""
    from videoloop3d_amd.MPV import MPMeshVid, atlas_to_stack, stack_to_atlas
    g = RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    args = RM.mpv_args(5, regs={})
    h, w, tar_e, K_crop, _ = RM.crop_view(g)
    v = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0, atlas_exact=True)
    with torch.no_grad():
        v.stack.copy_(atlas_to_stack(torch.from_numpy(g["c_atlas_dyn"]), over["mpi_d"], over["atlas_grid_h"]))
    v = v.to(dev).train()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, extra = v(h, w, tar_e, K_crop, res=torch.from_numpy(g["res"]).to(dev), losscfg=R4.collate(RM.OTHER_LOSSES[name]))
    _rel(extra["swd"], g[f"f_{name}_extra_swd"], "swd")
    (gs,) = torch.autograd.grad(extra["swd"].sum(), v.stack)
    _rel(stack_to_atlas(gs, over["atlas_grid_h"]), g[f"f_{name}_grad_atlas_dyn"], "grad atlas_dyn")


def test_no_loop_padding_no_gain_matches_the_reference(dev):
    """golden (g): mpv_isloop off, scale_invariant off (the parser's defaults; the shipped config sets both)."""
    from videoloop3d_amd.MPV import MPMeshVid, atlas_to_stack, stack_to_atlas
    g = RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    args = RM.mpv_args(5, regs={})
    args.mpv_isloop, args.scale_invariant = False, False
    h, w, tar_e, K_crop, _ = RM.crop_view(g)
    v = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0, atlas_exact=True)
    with torch.no_grad():
        v.stack.copy_(atlas_to_stack(torch.from_numpy(g["c_atlas_dyn"]), over["mpi_d"], over["atlas_grid_h"]))
    v = v.to(dev).train()
    _, extra = v(h, w, tar_e, K_crop, res=torch.from_numpy(g["res"]).to(dev), losscfg=R4.collate(RM.LOSS_CFGS["other"]))
    _rel(extra["swd"], g["g_extra_swd"], "swd")
    (gs,) = torch.autograd.grad(extra["swd"].sum(), v.stack)
    _rel(stack_to_atlas(gs, over["atlas_grid_h"]), g["g_grad_atlas_dyn"], "grad atlas_dyn")


def test_dense_mpmesh_atlas_exact_matches_the_reference(dev):
    """golden (a0) / (a): the reference's stage-1 MPMesh.forward on its dense atlas + loop-mask texture -- rgb AND the composited label
    (MPI.py:568-583), the gradient w.r.t. both textures, the evaluation render -- against MPMesh(atlas_exact=True) on the HIP kernels."""
    from videoloop3d_amd.MPI import MPMesh
    from videoloop3d_amd.MPV import atlas_to_stack, stack_to_atlas
    g15, g = RM.load("g15_sparsify"), RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    args = R4.make_args(learn_loop_mask=True, **over)
    h, w, tar_e, K_crop, K_full = RM.crop_view(g)
    m = MPMesh(args, H, W, ref_extrin, K, 1.0, 100.0, atlas_exact=True)
    with torch.no_grad():
        m.stack.copy_(atlas_to_stack(torch.from_numpy(g15["in_atlas"]), over["mpi_d"], over["atlas_grid_h"]))
        m.stack_mask.copy_(atlas_to_stack(torch.from_numpy(g15["in_atlas_mask"]), over["mpi_d"], over["atlas_grid_h"])[..., 0])
    m = m.to(dev).train()
    rgbl, extra = m(h, w, tar_e, K_crop)
    assert extra == {} and rgbl.shape == (1, 4, h, w)
    _close(rgbl, g["a0_rgbl"], 1e-4, "rgbl")
    gs, gm = torch.autograd.grad((rgbl * torch.from_numpy(g["a0_G"]).to(dev)).sum(), [m.stack, m.stack_mask])
    _rel(stack_to_atlas(gs, over["atlas_grid_h"]), g["a0_grad_atlas"], "grad atlas")
    _rel(stack_to_atlas(gm[..., None], over["atlas_grid_h"]), g["a0_grad_atlas_mask"], "grad atlas_mask")
    assert float(abs(g["a0_grad_atlas_mask"]).max()) > 1e-3
    m.eval()
    with torch.no_grad():
        _close(m(H, W, tar_e, K_full)[0], g["a_eval_rgbl_full"], 1e-4, "eval")
