"""Tile culling on the dense stack (videoloop3d_amd/tiles.py; MPI.py:288-442, MPV.py:235-288, utils.py:298-317).  CPU only."""
import types

import pytest

import numpy as np
import torch

from videoloop3d_amd import tiles


def _unfold_filter(alpha, reduce):
    """utils.py:298-317 as written there: 3x3 unfold with zero padding, max (dilate) or min (erode) over the window."""
    b, l, h, w = alpha.shape
    u = torch.nn.Unfold(3, dilation=1, padding=1, stride=1)(alpha.reshape(-1, 1, h, w))
    return (u.max(dim=1)[0] if reduce == "max" else u.min(dim=1)[0]).reshape_as(alpha)


def test_erode_dilate_match_the_unfold_form():
    torch.manual_seed(0)
    a = torch.rand(2, 3, 17, 23)
    assert torch.equal(tiles.dilate(a), _unfold_filter(a, "max"))
    assert torch.equal(tiles.erode(a), _unfold_filter(a, "min"))
    assert float(tiles.erode(torch.ones(1, 1, 5, 5))[0, 0, 0, 0]) == 0.0     # zero padding erodes the border


def test_classify_quads():
    a = torch.zeros(2, 40, 60)
    a[0, 10:20, 15:30] = 0.9          # a solid blob: kept
    a[1, 30:33, 50:52] = 0.5          # a 3x2 speck: eroded away (erode_num = 2 needs > 4 px)
    a[1, 2:12, 2:12] = 0.02           # below the alpha threshold
    m = torch.zeros(2, 40, 60)
    m[0, 12:18, 18:28] = 0.9
    keep, dyn = tiles.classify_quads(a, m, 4, 6, erode_num=1)
    assert keep.shape == (2, 4, 6) and not keep[1].any()
    # quads are 9.75 x 9.83 px; the blob dilated by 3 px touches rows 7..22, cols 12..32
    assert keep[0].int().tolist() == [[0, 1, 1, 1, 0, 0], [0, 1, 1, 1, 0, 0], [0, 1, 1, 1, 0, 0], [0, 0, 0, 0, 0, 0]]
    assert dyn[0].int().tolist() == [[0, 0, 0, 0, 0, 0], [0, 1, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0]]
    assert (dyn & ~keep).sum() == 0
    keep2, dyn2 = tiles.classify_quads(a, None, 4, 6, erode_num=1, rmfirstlayer=1)
    assert not keep2[0].any() and torch.equal(keep2, dyn2)


def test_texel_mask_covers_every_tap_of_a_kept_quad():
    torch.manual_seed(1)
    Hs, Ws, QH, QW = 37, 53, 5, 7
    q = torch.rand(3, QH, QW) > 0.6
    tm = tiles.quad_to_texel_mask(q, Hs, Ws)
    ch, cw = (Hs - 1) / QH, (Ws - 1) / QW
    ys, xs = torch.rand(4000) * (Hs - 1), torch.rand(4000) * (Ws - 1)
    qy, qx = (ys / ch).floor().clamp(max=QH - 1).long(), (xs / cw).floor().clamp(max=QW - 1).long()
    for d in range(3):
        inside = q[d, qy, qx]
        for dy in (0, 1):
            for dx in (0, 1):
                ty, tx = (ys.floor().long() + dy).clamp(max=Hs - 1), (xs.floor().long() + dx).clamp(max=Ws - 1)
                assert bool(tm[d, ty[inside], tx[inside]].all())
    assert not tiles.quad_to_texel_mask(torch.zeros(1, QH, QW, dtype=torch.bool), Hs, Ws).any()
    assert tiles.quad_to_texel_mask(torch.ones(1, QH, QW, dtype=torch.bool), Hs, Ws).all()


def test_tie_static_grad_sums_over_frames():
    keep = torch.tensor([[[True, True, False]]])
    dyn = torch.tensor([[[False, True, False]]])
    g = torch.rand(1, 4, 9, 31, 2)
    t = tiles.tie_static_grad(g, keep, dyn)
    # quads are 10 px wide: texels 0..8 are read by the static quad only, 9..20 also by the dynamic one, 21.. only by the culled one
    assert torch.allclose(t[:, :, :, :9], g[:, :, :, :9].sum(1, keepdim=True).expand(-1, 4, -1, -1, -1))
    assert torch.equal(t[:, :, :, 9:21], g[:, :, :, 9:21])
    assert float(t[:, :, :, 21:].abs().max()) == 0.0


def _args(**kw):
    a = dict(mpi_h_scale=1.0, mpi_w_scale=1.0, mpi_d=3, rgb_mlp_type="direct", rgb_activate="sigmoid", alpha_activate="sigmoid",
             bg_color="", learn_loop_mask=True, mpi_h_verts=5, mpi_w_verts=7, sparsify_rmfirstlayer=0,
             mpv_frm_num=4, mpv_isloop=True, init_std=0.5, scale_invariant=True, fp16=False,
             swd_patch_size=3, swd_patcht_size=3, swd_stride=2, swd_stridet=1)
    a.update(kw)
    return types.SimpleNamespace(**a)


def test_sparsify_and_init_from_mpi():
    from videoloop3d_amd.MPI import ALPHA_INIT_VAL, MPMesh
    from videoloop3d_amd.MPV import MPMeshVid
    H, W = 40, 60
    K = np.array([[50., 0, 30], [0, 50., 20], [0, 0, 1]])
    mpi = MPMesh(_args(), H, W, np.eye(4), K, 1.0, 100.0)
    with torch.no_grad():
        mpi.stack[0, 0, 10:20, 15:30, 3] = 3.0
        mpi.stack[2, 0, 5:30, 5:50, 3] = 1.0
        mpi.stack_mask[2, 0, 12:24, 20:40] = 4.0
    mpi.sparsify_faces(erode_num=1, alpha_thresh=0.03, loop_thresh=0.5)
    assert mpi.is_sparse and mpi.has_dyn and not hasattr(mpi, "stack_mask") and not mpi.learn_loop_mask
    keep, dyn = mpi.quad_keep, mpi.quad_dyn
    assert keep[0].any() and not keep[1].any() and dyn[2].any() and not dyn[0].any()      # untouched texels (ALPHA_INIT_VAL) count as -10
    # culled texels render nothing, kept texels keep their logits
    assert float(mpi.stack[1, 0, :, :, 3].max()) == tiles.CULLED_ALPHA
    assert float(mpi.stack[0, 0, 12, 20, 3]) == 3.0 and float(mpi.stack[0, 0, 38, 58, 3]) == tiles.CULLED_ALPHA
    sd = mpi.state_dict()
    assert sd["self.is_sparse"] is True and "quad_keep" in sd and sd["self.quad_h"] == 4

    vid = MPMeshVid(_args(), H, W, np.eye(4), K, 1.0, 100.0)
    vid.init_from_mpi(sd)
    assert vid.stack.shape == (3, 4, 40, 60, 4) and vid.is_sparse and vid.stack.requires_grad
    assert all(torch.equal(vid.stack[:, t], mpi.stack[:, 0]) for t in range(4))
    # the hook: static texels get the summed gradient in every frame, culled ones none
    g = torch.rand_like(vid.stack)
    (vid.stack * g).sum().backward()
    gr = vid.stack.grad
    assert torch.allclose(gr[0, 1, 12, 20], g[0, :, 12, 20].sum(0)) and torch.allclose(gr[0, 3, 12, 20], g[0, :, 12, 20].sum(0))
    assert torch.equal(gr[2, 1, 15, 30], g[2, 1, 15, 30])                                   # dynamic
    assert float(gr[1].abs().max()) == 0.0                                                    # culled plane
    # an un-sparsified MPI loads as all-dynamic
    dense = MPMesh(_args(learn_loop_mask=False), H, W, np.eye(4), K, 1.0, 100.0)
    vid2 = MPMeshVid(_args(), H, W, np.eye(4), K, 1.0, 100.0)
    vid2.init_from_mpi(dense.state_dict())
    assert not vid2.is_sparse and vid2.quad_keep is None and vid2._tie_hook is None
    # lod keeps the tying alive on the resized parameter
    vid.lod(0.5)
    assert vid._tie_hook is not None and vid.stack.shape[2:4] == (20, 30)


def test_lod_of_a_sparsified_model_does_not_bleed_culled_logits():
    """MPV.py:157-164 resizes every tile on its own; on the dense stack the culled texels hold CULLED_ALPHA (-1e4), so lod() must
    resample with the kept-texel mask as the weight: at every level of the shipped pyramid (down to ~0.2x and back up) the texels
    kept quads can read hold convex combinations of KEPT values only, and the culling is re-applied at the new resolution."""
    from videoloop3d_amd.MPV import MPMeshVid
    H, W = 80, 120
    K = np.array([[100., 0, 60], [0, 100., 40], [0, 0, 1]])
    vid = MPMeshVid(_args(mpv_frm_num=2), H, W, np.eye(4), K, 1.0, 100.0)
    torch.manual_seed(1)
    keep = torch.zeros(3, 4, 6, dtype=torch.bool)
    keep[0, 1:3, 1:4] = True            # a kept block with culled quads all around it
    keep[2] = torch.rand(4, 6) < 0.5
    vid.register_buffer("quad_keep", keep)
    vid.register_buffer("quad_dyn", keep.clone())
    vid.is_sparse = vid.has_dyn = True
    with torch.no_grad():
        vid.stack.uniform_(-3.0, 3.0)
        tiles.cull_stack_(vid.stack.data, keep)
    lo, hi = -3.0, 3.0
    for factor in (0.25, 0.5, 1.0, 0.2, 0.75):
        vid.lod(factor)
        D, T, hs, ws, _ = vid.stack.shape
        kept_t = tiles.quad_to_texel_mask(keep, hs, ws)[:, None].expand(D, T, hs, ws)
        v = vid.stack.detach()
        assert bool(torch.isfinite(v).all())
        assert float(v[kept_t].min()) >= lo - 1e-4 and float(v[kept_t].max()) <= hi + 1e-4, factor     # no -37 / -7509 logits
        assert bool((v[..., 3][~kept_t] == tiles.CULLED_ALPHA).all())                                    # culling re-applied
        assert vid._tie_hook is not None
    # an un-sparsified model takes the plain filter (torchvision Resize semantics: args.lod_antialias), untouched by the fix
    dense = MPMeshVid(_args(mpv_frm_num=2), H, W, np.eye(4), K, 1.0, 100.0)
    ref = torch.nn.functional.interpolate(dense.stack.detach().permute(0, 1, 4, 2, 3).reshape(6, 4, 80, 120), size=(40, 60),
                                          mode="bilinear", align_corners=False, antialias=bool(getattr(dense.args, "lod_antialias", False))).reshape(3, 2, 4, 40, 60).permute(0, 1, 3, 4, 2)
    dense.lod(0.5)
    assert torch.allclose(dense.stack.detach(), ref, atol=1e-6)


def test_stage2_checkpoint_roundtrip(tmp_path):
    """train_3dvid.py:295-306 / scripts/script_render_video.py:116-119: a saved MPMeshVid state_dict (incl. the "self.*" scalars and
    the quad maps, possibly at a pyramid level) is loaded back with init_from_mpi."""
    from videoloop3d_amd.MPV import MPMeshVid
    H, W = 40, 60
    K = np.array([[50., 0, 30], [0, 50., 20], [0, 0, 1]])
    a = MPMeshVid(_args(), H, W, np.eye(4), K, 1.0, 100.0)
    a.register_buffer("quad_keep", torch.rand(3, 4, 6) < 0.5)
    a.register_buffer("quad_dyn", a.quad_keep & (torch.rand(3, 4, 6) < 0.5))
    a.is_sparse = a.has_dyn = True
    a.lod(0.5)
    torch.save({"epoch_i": 3, "network_state_dict": a.state_dict()}, tmp_path / "l0_epoch_0003.tar")
    ck = torch.load(tmp_path / "l0_epoch_0003.tar", weights_only=False)
    b = MPMeshVid(_args(mpv_frm_num=7), H, W, np.eye(4), K, 1.0, 100.0)
    b.init_from_mpi(ck["network_state_dict"])
    assert b.frm_num == 4 and torch.equal(b.stack.detach(), a.stack.detach()) and b.spec.scale == a.spec.scale
    assert b.is_sparse and torch.equal(b.quad_keep, a.quad_keep) and torch.equal(b.quad_dyn, a.quad_dyn) and b._tie_hook is not None


def test_reference_checkpoint_is_read_onto_the_dense_stack():
    """MPV.py:235-288 on a checkpoint in the REFERENCE's format (face lists + packed static / dynamic atlases, restated by
    oracle/ckpt_oracle.py from MPI.py:296-436): the tiles land on the right texels of the dense stack, the quad maps come back,
    static content is shared by all frames."""
    from oracle.ckpt_oracle import pack_reference_state
    from videoloop3d_amd.MPV import MPMeshVid
    torch.manual_seed(0)
    D, T, hv, wv = 3, 4, 4, 5
    H, W = 1 + 8 * (hv - 1), 1 + 8 * (wv - 1)
    keep = torch.rand(D, hv - 1, wv - 1) < 0.7
    dyn = keep & (torch.rand(D, hv - 1, wv - 1) < 0.5)

    def closed(mask):                                   # closed plane-pixel rectangles of the quads of a map
        m = torch.zeros(D, H, W, dtype=torch.bool)
        for d_, qy, qx in mask.nonzero().tolist():
            m[d_, 8 * qy:8 * qy + 9, 8 * qx:8 * qx + 9] = True
        return m

    stack = torch.randn(D, T, H, W, 4)
    static_only = (closed(keep & ~dyn) & ~closed(dyn))[:, None, :, :, None]
    stack = torch.where(static_only, stack[:, :1].expand_as(stack), stack)       # static tiles hold ONE frame (MPI.py:380-395)
    sd = pack_reference_state(stack, keep, dyn, hv, wv, torch.linspace(1, 2, D))
    assert "stack" not in sd and sd["atlas"].shape[0] == 1 and sd["atlas_dyn"].shape[0] == T
    s2, k2, d2 = tiles.stack_from_reference_state(sd, H, W, hv, wv, T)
    assert torch.equal(k2, keep) and torch.equal(d2, dyn)
    ck = closed(keep)
    assert float((s2 - stack).abs().amax((1, 4))[ck].max()) <= 1e-4
    assert bool((s2[..., 3][(~ck)[:, None].expand(D, T, H, W)] == tiles.CULLED_ALPHA).all())
    # through the module: frame count taken from the dynamic atlas, maps registered, static tying installed
    a = _args(mpi_d=D, mpv_frm_num=2, mpi_h_verts=hv, mpi_w_verts=wv)
    vid = MPMeshVid(a, H, W, np.eye(4), np.array([[30., 0, W / 2], [0, 30., H / 2], [0, 0, 1]]), 1.0, 100.0)
    vid.init_from_mpi(sd, tile_layout="lattice")
    assert vid.frm_num == T and vid.stack.shape == (D, T, H, W, 4) and vid.is_sparse and vid._tie_hook is not None
    assert torch.equal(vid.quad_keep, keep) and torch.equal(vid.quad_dyn, dyn)
    assert torch.allclose(vid.planedepth, torch.linspace(1, 2, D))


def test_u8_cache_follows_the_source_tensor():
    """tiles.as_u8: the uint8 form of a (constant) quad map is cached per source tensor and version -- an in-place change of the map or a new map
    must not be served from the cache."""
    from videoloop3d_amd import tiles
    m = torch.zeros((2, 3, 4), dtype=torch.bool)
    m[0, 1, 2] = True
    a = tiles.as_u8(m)
    assert a.dtype == torch.uint8 and a.is_contiguous() and torch.equal(a.bool(), m)
    assert tiles.as_u8(m) is a                                  # same tensor, same version: the cached conversion
    m[1, 0, 0] = True                                           # in place: the version counter moves
    b = tiles.as_u8(m)
    assert b is not a and torch.equal(b.bool(), m)
    m2 = m.clone()
    assert tiles.as_u8(m2) is not b and torch.equal(tiles.as_u8(m2).bool(), m2)
    u = torch.ones((2, 2), dtype=torch.uint8)
    assert tiles.as_u8(u) is u and tiles.as_u8(None) is None    # already bytes / absent: passed through


@pytest.mark.parametrize("shape", [(12, 12, 9, 9), (9, 9, 12, 12), (7, 7, 9, 9), (10, 10, 5, 5), (12, 11, 6, 7), (2, 2, 5, 3), (5, 5, 2, 2), (6, 6, 6, 6)])
def test_tile_resize_without_antialiasing_is_interpolates_bilinear(shape):
    """MPV._resize_tiles(antialias=False) spells F.interpolate(mode='bilinear', align_corners=False) with gathers (the per-tile lod of a tile-exact
    model, MPV.py:157-162 under torchvision 0.11): same taps and weights, equal to a few units in the last place."""
    from videoloop3d_amd.MPV import _resize_tiles
    th, tw, h, w = shape
    x = torch.randn(37, 4, th, tw, generator=torch.Generator().manual_seed(th * 100 + h))
    ref = torch.nn.functional.interpolate(x, size=(h, w), mode="bilinear", align_corners=False, antialias=False)
    out = _resize_tiles(x, h, w, False)
    assert out.shape == ref.shape and torch.allclose(out, ref, rtol=0, atol=4e-6)
    aa = torch.nn.functional.interpolate(x, size=(h, w), mode="bilinear", align_corners=False, antialias=True)
    assert torch.equal(_resize_tiles(x, h, w, True), aa)
