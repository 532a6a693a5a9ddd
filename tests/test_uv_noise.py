"""add_uv_noise of the reference (MPV.py:420-423, MPI.py:519-522; config_parser.py:48; off in every shipped configuration): every sample's UV is
jittered by half a texel while training.  The reference draws from torch.rand (not reproducible from outside), so what is pinned is the
SEMANTICS -- amplitude and distribution of the jitter, one draw per (pixel, layer) shared by the frames, coverage decided at the unjittered
position -- through the oracle's restatement of the kernels' counter-hash field (oracle/mpi_oracle.uv_jitter_field), which the GPU tests hold the
kernels to."""
import types

import numpy as np
import pytest
import torch

from oracle import mpi_oracle as MO
from videoloop3d_amd import synth

TOL = 1e-4
MPV_O = dict(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post")


def _homos(D, H, W, scale=1.5):
    from videoloop3d_amd.utils_mpi import compute_homography, make_depths
    ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
    tar_e = tar_e.clone()
    tar_e[:3, 3] *= scale
    return compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                              make_depths(D, 1.0, 100.0).flip(0)[None])[0]


def test_jitter_field_is_half_a_texel_uniform_and_a_pure_function():
    """hpix * (2 rand - 1) with hpix = 1 / (size - 1) in normalised [-1, 1] coordinates is +-0.5 texel, uniform (MPV.py:420-423)."""
    j = MO.uv_jitter_field(1234567, 6, 90, 120)
    assert j.shape == (6, 90, 120, 2) and j.dtype == torch.float32
    assert float(j.min()) >= -0.5 and float(j.max()) < 0.5
    assert abs(float(j.mean())) < 2e-3 and abs(float(j.var()) - 1.0 / 12.0) < 2e-3
    assert abs(float((j[..., 0] * j[..., 1]).mean())) < 2e-3                          # x and y draws are independent
    assert abs(float((j[:, :, 1:] * j[:, :, :-1]).mean())) < 2e-3                     # neighbouring pixels are
    assert abs(float((j[1:] * j[:-1]).mean())) < 2e-3                                 # ... and so are neighbouring layers
    assert torch.equal(j, MO.uv_jitter_field(1234567, 6, 90, 120))
    assert not torch.equal(j, MO.uv_jitter_field(1234568, 6, 90, 120))
    # a window of the frame sees the same field (the hash takes the FRAME pixel): crops / row bands / tilings agree
    assert torch.equal(MO.uv_jitter_field(99, 3, 20, 30, row0=7, col0=11), MO.uv_jitter_field(99, 3, 40, 60)[:, 7:27, 11:41])


def test_oracle_noise_moves_the_taps_not_the_coverage():
    D, T, Hs, Ws, H, W = 4, 2, 40, 56, 48, 64
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=3)
    homos = _homos(D, H, W, scale=3.0)                     # plane borders inside the frame
    l0, c0 = MO.sample_layers(stack, homos, H, W, MO.RenderSpec(**MPV_O))
    l1, c1 = MO.sample_layers(stack, homos, H, W, MO.RenderSpec(uv_noise_seed=5, **MPV_O))
    assert torch.equal(c0, c1) and 0.05 < float(c0.float().mean()) < 0.999
    assert float((l0 - l1).abs().max()) > 1e-3
    assert torch.equal(l1[0], MO.sample_layers(stack[:, :1], homos, H, W, MO.RenderSpec(uv_noise_seed=5, **MPV_O))[0][0])      # one field for all frames


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import __graft_entry__ as g
    g.build()
    return torch.device("cuda:0")


def _maxabs(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("T,keep_frac", [(1, None), (3, None), (2, 0.4)])
def test_render_with_uv_noise_matches_the_oracle(dev, T, keep_frac):
    """forward (one-frame kernel, plain and tile-culled) and the atomics backward draw the oracle's field: image, alpha and the stack gradient
    within the north star's 1e-4; seed 0 is the noise-free render; another seed is another image."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, Hs, Ws, H, W = 6, 150, 200, 139, 187
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=13)
    homos = _homos(D, H, W)
    keep = None
    if keep_frac is not None:
        torch.manual_seed(3)
        keep = torch.rand(D, 6, 9) < keep_frac
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    g_a = synth.hash_uniform((T, H, W), seed=6) - 0.5
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, _ = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(uv_noise_seed=777, **MPV_O), quad_keep=keep)
    (gs_o,) = torch.autograd.grad([rgb_o, alpha_o], s_cpu, [g_rgb, g_a])
    s_gpu = stack.to(dev).requires_grad_(True)
    qk = None if keep is None else keep.to(dev)
    import dataclasses
    rgb, alpha = render_planes(s_gpu, homos.to(dev), H, W, dataclasses.replace(RenderSpec.mpv(), uv_noise_seed=777), quad_keep=qk)
    (gs,) = torch.autograd.grad([rgb, alpha], s_gpu, [g_rgb.to(dev), g_a.to(dev)])
    assert _maxabs(rgb, rgb_o) <= TOL and _maxabs(alpha, alpha_o) <= TOL
    assert _maxabs(gs, gs_o) <= TOL * max(1.0, float(gs_o.abs().max()))
    rgb0, _ = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec.mpv(), quad_keep=qk)
    rgb_o0, _, _ = MO.render_planes(stack, homos, H, W, MO.RenderSpec(**MPV_O), quad_keep=keep)
    assert _maxabs(rgb0, rgb_o0) <= TOL and _maxabs(rgb0, rgb) > 1e-3
    rgb2, _ = render_planes(s_gpu, homos.to(dev), H, W, dataclasses.replace(RenderSpec.mpv(), uv_noise_seed=778), quad_keep=qk)
    assert _maxabs(rgb2, rgb) > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("culled", [False, True])
def test_regularisers_with_uv_noise_match_the_oracle(dev, culled):
    """the fused forward with the layer regularisers (dense: render_fwd_reg_k + the slot kernel over irregular pairs; tile-culled: the slot
    kernel that composites as it goes) samples at the jittered positions too: image, the four smoothness sums (hit-slot order) and the
    gradient of both against the oracle."""
    import dataclasses
    from videoloop3d_amd.render import RenderSpec, render_planes_with_regularisers
    D, T, Hs, Ws, H, W = 5, 2, 90, 120, 83, 111
    torch.manual_seed(5)
    keep = (torch.rand(D, 4, 6) < 0.5) if culled else None
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=17)
    homos = _homos(D, H, W, scale=1.2)
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, _, layers = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(uv_noise_seed=4242, **MPV_O), return_layers=True, quad_keep=keep)
    sums_o = torch.stack([(layers[:, :, 1:, :, :3] - layers[:, :, :-1, :, :3]).abs().sum(), (layers[:, 1:, :, :, :3] - layers[:, :-1, :, :, :3]).abs().sum(),
                          (layers[:, :, 1:, :, 3] - layers[:, :, :-1, :, 3]).abs().sum(), (layers[:, 1:, :, :, 3] - layers[:, :-1, :, :, 3]).abs().sum()])
    wts = torch.tensor([1e-4, 2e-4, 3e-4, 4e-4])
    (gs_o,) = torch.autograd.grad((rgb_o * g_rgb).sum() + (sums_o * wts).sum(), s_cpu)
    s_gpu = stack.to(dev).requires_grad_(True)
    spec = dataclasses.replace(RenderSpec.mpv(), uv_noise_seed=4242)
    rgb, alpha, sums, _ = render_planes_with_regularisers(s_gpu, homos.to(dev), H, W, spec, quad_keep=None if keep is None else keep.to(dev))
    (gs,) = torch.autograd.grad((rgb * g_rgb.to(dev)).sum() + (sums * wts.to(dev)).sum(), s_gpu)
    assert _maxabs(rgb, rgb_o) <= TOL and _maxabs(alpha, alpha_o) <= TOL
    assert float(((sums.detach().cpu() - sums_o.detach()).abs() / sums_o.detach().abs().clamp_min(1.0)).max()) <= 1e-4
    assert _maxabs(gs, gs_o) <= TOL * max(1.0, float(gs_o.abs().max()))


def _args(**kw):
    a = dict(mpv_frm_num=4, mpv_isloop=True, mpi_h_scale=1.1, mpi_w_scale=1.1, mpi_d=5, atlas_grid_h=1, init_std=0.3,
             rgb_mlp_type="direct", rgb_activate="sigmoid", alpha_activate="sigmoid", bg_color="", scale_invariant=True, fp16=False,
             swd_patch_size=3, swd_patcht_size=3, swd_stride=2, swd_stridet=1, sparsity_loss_weight=0.0, rgb_smooth_loss_weight=0.2,
             a_smooth_loss_weight=0.2, density_loss_weight=0.0, d_smooth_loss_weight=0.0, optimizer="adam", lrate=5e-3, lrate_decay=30,
             add_uv_noise=True)
    a.update(kw)
    return types.SimpleNamespace(**a)


@pytest.mark.gpu
def test_module_trains_with_add_uv_noise(dev):
    """MPMeshVid(args.add_uv_noise): training forwards draw a new field each (two forwards differ; torch.manual_seed reproduces them), eval is the
    noise-free image, and a training iteration steps through the atomics backward + the window step kernel (the fused step declines)."""
    import warnings
    from videoloop3d_amd.MPV import MPMeshVid
    H, W, h, w = 96, 128, 48, 64
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], np.float64)
    torch.manual_seed(5)
    m = MPMeshVid(_args(), H, W, np.eye(4), K, 1.0, 100.0).to(dev).train()
    tar = np.eye(4)
    tar[:3, 3] = [0.03, 0.01, 0.0]
    ext, intr = torch.tensor(tar)[None], torch.tensor(K)[None]
    res = synth.hash_uniform((1, 9, 3, H, W), seed=8, device=dev)
    cfg = dict(loss_name=["gpnn_lm"], loss_gain=torch.tensor([1.0]), macro_block=torch.tensor([65]), patch_size=torch.tensor([3]),
               stride=torch.tensor([2]), patcht_size=torch.tensor([3]), stridet=torch.tensor([1]), alpha=torch.tensor([10000.0]),
               dist_fn=["mse"], rou=["-2"], scaling=torch.tensor([0.1]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():
            torch.manual_seed(11)
            a1 = m.render(H, W, ext, intr, list(range(4)))[0].clone()
            a2 = m.render(H, W, ext, intr, list(range(4)))[0].clone()
            torch.manual_seed(11)
            a3 = m.render(H, W, ext, intr, list(range(4)))[0].clone()
            m.eval()
            e1 = m.render(H, W, ext, intr, list(range(4)))[0].clone()
            e2 = m.render(H, W, ext, intr, list(range(4)))[0].clone()
            m.train()
        assert torch.equal(a1, a3) and torch.equal(e1, e2)
        assert float((a1 - a2).abs().max()) > 1e-4 and float((a1 - e1).abs().max()) > 1e-4
        assert float((a1 - e1).abs().mean()) < 0.05                  # half a texel of a smooth texture: a small perturbation
        opt = m.get_optimizer(0)
        before = m.state_dict()["stack"].clone()
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            _, extra = m(H, W, ext, intr, res=res, losscfg=dict(cfg))
            loss = extra["swd"].sum() + 0.2 * extra["rgb_smooth"].sum() + 0.2 * extra["a_smooth"].sum()
            loss.backward()
            opt.step()
        assert getattr(opt, "fused_steps", 0) == 0
        after = m.state_dict()["stack"]
        assert torch.isfinite(after).all() and float((after - before).abs().max()) > 1e-4


@pytest.mark.gpu
def test_stage1_loop_mask_with_uv_noise_matches_the_oracle(dev):
    """MPMesh with add_uv_noise AND learn_loop_mask (MPI.py:519-522 with :568-583; configs/mpi_base.txt ships learn_loop_mask): the colour samples are
    jittered, the loop mask is sampled at the PLAIN positions and composited with the jittered samples' detached alphas -- two sampling positions
    per layer (vl3d_label_noise_fwd / _bwd).  Image, label and the gradients w.r.t. both textures against the oracle on the same jitter field."""
    from oracle import mpv_oracle
    from videoloop3d_amd.MPI import MPMesh
    H, W, h, w = 40, 56, 30, 44
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], np.float64)
    args = types.SimpleNamespace(mpi_h_scale=1.1, mpi_w_scale=1.1, mpi_d=5, atlas_grid_h=1, init_std=0.3, rgb_mlp_type="direct", rgb_activate="sigmoid",
                                 alpha_activate="sigmoid", learn_loop_mask=True, add_uv_noise=True, mpi_h_verts=4, mpi_w_verts=4, bg_color="", optimizer="adam",
                                 lrate=0.01, lrate_decay=100, sparsity_loss_weight=0.0, rgb_smooth_loss_weight=0.0, a_smooth_loss_weight=0.0,
                                 density_loss_weight=0.0, d_smooth_loss_weight=0.0, l_smooth_loss_weight=0.0, normalize_blendweight_fordepth=False)
    m = MPMesh(args, H, W, np.eye(4), K, 1.0, 100.0)
    with torch.no_grad():
        m.stack.copy_(synth.make_plane_stack(5, 1, m.mpi_h, m.mpi_w, seed=3))
        m.stack_mask.copy_((synth.hash_uniform((5, 1, m.mpi_h, m.mpi_w), seed=9) - 0.5) * 4)
    m = m.to(dev).train()
    tar = np.eye(4)
    tar[:3, 3] = [0.04, -0.02, 0.0]
    ext, intr = torch.tensor(tar)[None].float(), torch.tensor(K)[None].float()
    intr_c = intr.clone()
    intr_c[:, 0, 2] -= 5
    intr_c[:, 1, 2] -= 4
    torch.manual_seed(21)
    seed = int(torch.randint(1, 2 ** 31 - 1, (1,)))                      # the draw MPMesh.render makes for this view
    torch.manual_seed(21)
    rgbl, _ = m(h, w, ext, intr_c)
    assert rgbl.shape == (1, 4, h, w)
    G = (synth.hash_uniform(tuple(rgbl.shape), seed=4) - 0.5).to(dev)
    gs, gm = torch.autograd.grad((rgbl * G).sum(), [m.stack, m.stack_mask])
    st, mk = m.stack.detach().cpu().requires_grad_(True), m.stack_mask.detach().cpu().requires_grad_(True)
    rgbl_o, _ = mpv_oracle.mpi_forward(st, mk, args, H, W, np.eye(4), K, 1.0, 100.0, h, w, ext, intr_c, uv_noise_seeds=[seed])
    gs_o, gm_o = torch.autograd.grad((rgbl_o * G.cpu()).sum(), [st, mk])
    assert float((rgbl.cpu() - rgbl_o).abs().max()) <= TOL
    assert float((gs.cpu() - gs_o).abs().max()) <= TOL * max(1.0, float(gs_o.abs().max()))
    assert float((gm.cpu() - gm_o).abs().max()) <= TOL * max(1.0, float(gm_o.abs().max())) and float(gm_o.abs().max()) > 1e-3
    # the two positions matter: the label with the mask ALSO sampled at the jittered positions is a different image
    lab_same, _ = mpv_oracle.mpi_forward(st.detach(), mk.detach(), args, H, W, np.eye(4), K, 1.0, 100.0, h, w, ext, intr_c)      # no noise at all
    assert float((rgbl_o[:, 3] - lab_same[:, 3]).abs().max()) > 1e-3
    # eval: no noise (the fused label channel again)
    m.eval()
    with torch.no_grad():
        ev = m(h, w, ext, intr_c)[0]
    assert float((ev.cpu() - lab_same).abs().max()) <= TOL
