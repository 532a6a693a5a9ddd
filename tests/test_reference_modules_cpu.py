"""Module-level parity against what the reference's OWN MPI.py / MPV.py produce (goldens G14-G17, tests/golden/make_golden_r04.py):
constructors, sparsify_faces + state_dict, init_from_mpi, and render / forward downstream of a harness rasteriser.  CPU only: the
host-side product code (classifier, checkpoint reader, exporter, geometry) and the oracles (atlas_oracle, ckpt_oracle, mpv_oracle) are
checked here; tests/test_gpu_reference_modules.py runs the HIP modules against the same goldens."""
import copy

import numpy as np
import pytest
import torch

import refmod as RM
from oracle import atlas_oracle as AO, ckpt_oracle as CO, mpv_oracle
from videoloop3d_amd import export, tiles
from videoloop3d_amd.MPV import atlas_to_stack, stack_to_atlas
from videoloop3d_amd.utils_mpi import gen_mpi_vertices

R4 = RM.R4


# ---- G14: constructors ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["A", "B"])
def test_g14_constructor_geometry(name):
    from videoloop3d_amd.MPI import MPMesh
    from videoloop3d_amd.MPV import MPMeshVid
    g = RM.load("g14_constructors")
    H, W, over = R4.SHAPES[name]
    K, ref_extrin, _ = R4.scene(H, W)
    args = R4.make_args(learn_loop_mask=True, mpv_frm_num=3, init_std=0.3, **over)
    gh, D, hv, wv = args.atlas_grid_h, args.mpi_d, args.mpi_h_verts, args.mpi_w_verts
    gw = D // gh
    mpi = MPMesh(args, H, W, ref_extrin, K, 1.0, 100.0)
    mpv = MPMeshVid(copy.copy(args), H, W, ref_extrin, K, 1.0, 100.0)
    for tag, m in (("mpi", mpi), ("mpv", mpv)):
        p = f"{name}_{tag}_"
        Hs, Ws, ggh, ggw, Ah, Aw = (int(v) for v in g[p + "scalars"])
        assert (m.H_start, m.W_start) == (Hs, Ws) and (ggh, ggw) == (gh, gw)
        assert (Ah, Aw) == (gh * m.mpi_h, gw * m.mpi_w)                       # the atlas is the grid of the stack's planes
        assert np.array_equal(m.planedepth.numpy(), g[p + "planedepth"])
        assert np.array_equal(m.ref_intrin.numpy(), g[p + "ref_intrin"]) and np.array_equal(m.ref_extrin.numpy(), g[p + "ref_extrin"])
        # vertices: the product's gen_mpi_vertices on the product's shifted intrinsics == the `verts` property; `_verts` under normalize_verts
        verts = gen_mpi_vertices(m.mpi_h, m.mpi_w, m.ref_intrin_mpi, hv, wv, m.planedepth)
        assert np.allclose(verts.numpy(), g[p + "verts_property"], rtol=1e-6, atol=1e-6)
        sd = m.reference_state_dict()
        assert np.allclose(sd["_verts"].numpy(), g[p + "_verts"], rtol=1e-6, atol=1e-6)
    faces = export.quad_faces(D, hv, wv).reshape(-1, 3).numpy()
    assert np.array_equal(faces, g[f"{name}_mpi_faces"]) and np.array_equal(faces, g[f"{name}_mpi_uvfaces"])
    assert np.array_equal(faces, g[f"{name}_mpv_faces_dyn"]) and np.array_equal(faces, g[f"{name}_mpv_uvfaces_dyn"])
    assert g[f"{name}_mpv_faces"].shape == (0, 3) and g[f"{name}_mpv_uvs"].shape == (0, 2)       # MPV.py:95-100: the static lists start empty
    # UV layout: the oracle's restatement, and the product's per-axis closed form (tiles.cell_vertex_uvs), == the reference's tensors
    uvs_o = AO.reference_vertex_uvs(gh, gw, hv, wv).reshape(-1, 2).numpy()
    assert np.array_equal(uvs_o, g[f"{name}_mpi_uvs"]) and np.array_equal(uvs_o, g[f"{name}_mpv_uvs_dyn"])
    u, v = tiles.cell_vertex_uvs(gw, wv), tiles.cell_vertex_uvs(gh, hv)
    uv_p = torch.stack([u[None, :, None, :].expand(gh, gw, hv, wv), v[:, None, :, None].expand(gh, gw, hv, wv)], -1).reshape(-1, 2)
    assert np.allclose(uv_p.numpy(), g[f"{name}_mpi_uvs"], atol=2e-7)
    assert tuple(g[f"{name}_mpv_atlas_dyn_shape"]) == (3, 4, gh * mpv.mpi_h, gw * mpv.mpi_w)
    assert tuple(g[f"{name}_mpi_atlas_mask_shape"]) == (1, 1, gh * mpi.mpi_h, gw * mpi.mpi_w)
    assert float(g[f"{name}_mpv_dyn_alpha_init"].max()) == -2.0 == float(mpv.stack.detach()[..., 3].max())       # MPV.py:109-110
    assert float(mpi.stack.detach()[..., 3].max()) == -3.0                                                          # MPI.py:33, 103


# ---- G15: sparsify_faces --------------------------------------------------------------------------------------------------------
def _quad_map(q, D, QH, QW):
    m = torch.zeros((D, QH, QW), dtype=torch.bool)
    q = torch.from_numpy(q).long()
    m[q[:, 0], q[:, 1], q[:, 2]] = True
    return m


def _dense_product_mpi(g15, **kw):
    from videoloop3d_amd.MPI import MPMesh
    H, W, over, K, ref_extrin, _ = RM.case_A()
    args = R4.make_args(learn_loop_mask=True, **over, **kw)
    m = MPMesh(args, H, W, ref_extrin, K, 1.0, 100.0)
    with torch.no_grad():      # identical weights: the reference's atlas of plane cells IS the product's stack, cell by cell
        m.stack.copy_(atlas_to_stack(torch.from_numpy(g15["in_atlas"]), args.mpi_d, args.atlas_grid_h))
        m.stack_mask.copy_(atlas_to_stack(torch.from_numpy(g15["in_atlas_mask"]), args.mpi_d, args.atlas_grid_h)[..., 0])
    return m, args


@pytest.mark.parametrize("rm", [0, 1])
def test_g15_sparsify_classifies_like_the_reference(rm):
    g = RM.load("g15_sparsify")
    m, args = _dense_product_mpi(g, sparsify_rmfirstlayer=rm)
    e, at, lt = g["thresh"]
    m.sparsify_faces(erode_num=int(e), alpha_thresh=float(at), loop_thresh=float(lt))
    D, QH, QW = args.mpi_d, args.mpi_h_verts - 1, args.mpi_w_verts - 1
    pre = "rm1_" if rm else ""
    ks, kd = _quad_map(g[pre + "quads_static"], D, QH, QW), _quad_map(g[pre + "quads_dyn"], D, QH, QW)
    assert int(kd.sum()) >= 9 and int(ks.sum()) >= 9 and int((~(ks | kd)).sum()) >= 9          # all three classes are populated
    assert torch.equal(m.quad_keep, ks | kd) and torch.equal(m.quad_dyn, kd)
    assert m.is_sparse and m.has_dyn and not m.learn_loop_mask and not hasattr(m, "stack_mask")      # MPI.py:423-441
    # the plane-by-plane variant (textures without an atlas layout) is a different rule: it must NOT be what sparsify_faces ran
    a = m.alpha_activate(torch.where(torch.from_numpy(g["in_atlas"])[0, 3] == -3.0, torch.tensor(-10.0), torch.from_numpy(g["in_atlas"])[0, 3]))
    cells = atlas_to_stack(a[None, None], D, args.atlas_grid_h)[:, 0, :, :, 0]
    k2, _ = tiles.classify_quads(cells, None, QH, QW, int(e), float(at), float(lt), rm)
    assert not torch.equal(k2, ks | kd)


def test_g15_ckpt_oracle_packs_like_the_reference():
    """oracle/ckpt_oracle.sparsify_atlas (the test-side restatement of MPI.py:288-442) == the reference's state_dict."""
    g = RM.load("g15_sparsify")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    e, at, lt = g["thresh"]
    sd_o = CO.sparsify_atlas(torch.from_numpy(g["in_atlas"]), torch.from_numpy(g["in_atlas_mask"]), over["atlas_grid_h"], over["mpi_d"],
                             over["mpi_h_verts"], over["mpi_w_verts"], int(e), float(at), float(lt))
    sd = RM.state_dict_of(g, "sd_")
    for k in ("faces", "uvfaces", "faces_dyn", "uvfaces_dyn"):
        assert torch.equal(sd_o[k], sd[k]), k
    for k in ("uvs", "uvs_dyn", "atlas", "atlas_dyn"):
        assert sd_o[k].shape == sd[k].shape and float((sd_o[k] - sd[k]).abs().max()) <= 1e-6, k
    for k in sd:
        if k.startswith("self."):
            assert sd_o[k] == sd[k], k


def _check_state(out, sd, atol=1e-4):
    for k, v in sd.items():
        if torch.is_tensor(v):
            o = out[k]
            assert tuple(o.shape) == tuple(v.shape), (k, o.shape, v.shape)
            if v.dtype in (torch.int64, torch.int32):
                assert torch.equal(o.long(), v.long()), k
            elif v.numel():
                assert float((o.double() - v.double()).abs().max()) <= (atol if k.startswith("atlas") else 1e-6), k
        else:
            assert out[k] == v, (k, out[k], v)


def test_g15_reader_and_exporter_round_trip_the_reference_checkpoint():
    """tiles.stack_from_reference_state reads the reference's sparsified checkpoint texel for texel (tile lattice, quad maps) and
    export.reference_state_dict writes the reference's layout back: every tensor and every "self.*" scalar of the REAL state_dict."""
    from videoloop3d_amd.MPI import MPMesh
    g = RM.load("g15_sparsify")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    sd = RM.state_dict_of(g, "sd_")
    hv, wv, D = over["mpi_h_verts"], over["mpi_w_verts"], over["mpi_d"]
    st, keep, dyn = tiles.stack_from_reference_state(sd, 40, 60, hv, wv, 1)
    th = sd["atlas"].shape[2] // sd["self.atlas_grid_h"]
    assert st.shape == (D, 1, (hv - 1) * (th - 1) + 1, (wv - 1) * (th - 1) + 1, 4)
    ks, kd = _quad_map(g["quads_static"], D, hv - 1, wv - 1), _quad_map(g["quads_dyn"], D, hv - 1, wv - 1)
    assert torch.equal(keep, ks | kd) and torch.equal(dyn, kd)
    # every tile texel of the checkpoint sits on its lattice texel (interiors exactly, duplicated borders to the fp32 jitter of the two copies)
    (t_h, t_w), lists = tiles._aligned_tiles([("static", sd["faces"], sd["uvfaces"], sd["uvs"], sd["atlas"]),
                                              ("dyn", sd["faces_dyn"], sd["uvfaces_dyn"], sd["uvs_dyn"], sd["atlas_dyn"])], hv, wv)
    assert (t_h, t_w) == (th, th)
    for kind, d, vy, vx, y0, x0, atlas in lists:
        for i in range(0, len(d), 7):
            tile = atlas[0, :, int(y0[i]):int(y0[i]) + th, int(x0[i]):int(x0[i]) + th].permute(1, 2, 0)
            lat = st[int(d[i]), 0, int(vy[i]) * (th - 1):int(vy[i]) * (th - 1) + th, int(vx[i]) * (th - 1):int(vx[i]) * (th - 1) + th]
            assert torch.equal(lat[1:-1, 1:-1], tile[1:-1, 1:-1]) and float((lat - tile).abs().max()) <= 1e-4
    closed = torch.zeros(st.shape[0], *st.shape[2:4], dtype=torch.bool)           # closed texel rectangles of the kept quads
    for d_, qy, qx in keep.nonzero().tolist():
        closed[d_, qy * (th - 1):qy * (th - 1) + th, qx * (th - 1):qx * (th - 1) + th] = True
    assert bool((st[:, 0, :, :, 3][~closed] == tiles.CULLED_ALPHA).all()) and bool((st[:, 0, :, :, 3][closed] > -50).all())
    # through the module, and back out in the reference's layout
    m = MPMesh(R4.make_args(learn_loop_mask=True, **over), H, W, ref_extrin, K, 1.0, 100.0)
    m.init_from_mpi(sd)
    assert m.is_sparse and not m.learn_loop_mask and torch.equal(m.quad_keep, keep)
    assert m.spec.scale == ((st.shape[3] - 1) / 59, (st.shape[2] - 1) / 39)
    _check_state(m.reference_state_dict(), sd)


# ---- G16: MPMeshVid.init_from_mpi -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", ["exact", "lattice"])
def test_g16_init_from_mpi_of_the_sparsified_checkpoint(layout):
    """layout "exact" (the default): every quad keeps its tile with its own border texels -- the checkpoint comes back BIT FOR BIT;
    "lattice": neighbouring quads share their border texels (exact for this fresh checkpoint to the fp32 jitter of the duplicated samples)."""
    from videoloop3d_amd.MPV import MPMeshVid
    g15, g16 = RM.load("g15_sparsify"), RM.load("g16_init_from_mpi")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    args = R4.make_args(mpv_frm_num=4, mpv_isloop=True, init_std=0.2, **over)
    v = MPMeshVid(args, H, W, ref_extrin, K, 1.0, 100.0)
    kw = {} if layout == "exact" else dict(tile_layout="lattice")
    v.init_from_mpi(RM.state_dict_of(g15, "sd_"), **kw)           # the stage-1 checkpoint (one frame) -> T frames (MPV.py:254-260)
    ref = RM.state_dict_of(g16, "sparse_")
    assert v.frm_num == 4 == ref["atlas_dyn"].shape[0] and v.stack.shape[1] == 4 and v.tile_full == (10, 10) and v.is_sparse and v.has_dyn
    assert v.tile_own == ((10, 10) if layout == "exact" else None)
    assert v.stack.shape[2:4] == ((4 * 10, 6 * 10) if layout == "exact" else (4 * 9 + 1, 6 * 9 + 1))
    assert bool((v.stack.detach()[:, :1] == v.stack.detach()).all())          # every frame starts as the stage-1 texture
    _check_state(v.reference_state_dict(), ref, atol=0.0 if layout == "exact" else 1e-4)
    # ... and the stage-2 checkpoint itself (a T-frame dynamic atlas) reads back to the same model
    v2 = MPMeshVid(R4.make_args(mpv_frm_num=2, mpv_isloop=True, init_std=0.2, **over), H, W, ref_extrin, K, 1.0, 100.0)
    v2.init_from_mpi(ref, **kw)
    assert v2.frm_num == 4 and float((v2.stack.detach() - v.stack.detach()).abs().max()) <= 1e-6 and torch.equal(v2.quad_dyn, v.quad_dyn)
    # the product's own checkpoint of that model keeps the layout and the tile size for lod()
    v3 = MPMeshVid(copy.copy(args), H, W, ref_extrin, K, 1.0, 100.0)
    v3.init_from_mpi(v.state_dict())
    assert v3.tile_full == (10, 10) and v3.tile_own == v.tile_own and torch.equal(v3.stack.detach(), v.stack.detach()) and v3.spec == v.spec
    v3.lod(0.5)                                                    # tiles of max(int(10 * 0.5), 2) = 5 texels (MPV.py:146-151)
    assert v3.stack.shape[2:4] == ((4 * 5, 6 * 5) if layout == "exact" else (4 * 4 + 1, 6 * 4 + 1))
    if layout == "exact":
        # the reference's own operation, tile by tile (MPV.py:157-162): tile (d, qy, qx) of the new level is the resize of that tile alone
        assert v3.tile_own == (5, 5) and v3.spec.tile == (5, 5)
        t_old = v.stack.detach()[2, 1, 10:20, 30:40].permute(2, 0, 1)[None]
        want = torch.nn.functional.interpolate(t_old, size=(5, 5), mode="bilinear", align_corners=False, antialias=bool(getattr(v3.args, "lod_antialias", False)))[0].permute(1, 2, 0)
        if bool(v.quad_keep[2, 1, 3]):      # (the tiles' own resize spells F.interpolate's sum with gathers: equal to a few units in the last place)
            assert torch.allclose(v3.stack.detach()[2, 1, 5:10, 15:20], want, rtol=0, atol=5e-6)
        # at a pyramid level the export keeps the FULL atlas size under "self.atlas_full_*" (what the reference's lod scales from, MPV.py:149)
        sd5 = v3.reference_state_dict()
        assert sd5["atlas_dyn"].shape[-2] == sd5["self.atlas_grid_dyn_h"] * 5 and sd5["self.atlas_full_dyn_h"] == sd5["self.atlas_grid_dyn_h"] * 10


def test_g19_trained_checkpoint_round_trips_bit_for_bit():
    """golden G19 (tests/golden/make_golden_r06.py): a stage-2 checkpoint whose tiles were perturbed EVERYWHERE -- the two copies of every border
    sample differ, static tiles next to dynamic ones included.  state_dict -> model (tile-exact layout) -> reference_state_dict reproduces every
    tensor and scalar exactly; the shared-border reader cannot (its error on the atlases is the size of the perturbation)."""
    from videoloop3d_amd.MPV import MPMeshVid
    g = RM.load("g19_trained_tiles")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    sd = RM.state_dict_of(g, "h_sd_")
    v = MPMeshVid(RM.mpv_args(5), H, W, ref_extrin, K, 1.0, 100.0)
    v.init_from_mpi(sd)
    assert v.tile_own == (10, 10) and v.frm_num == 5
    out = v.reference_state_dict()

    def referenced(state):
        """the residual slots behind the last tile of an atlas (MPI.py:384, 392: copies of the last tile at sparsify time) belong to no face --
        parameters nothing ever reads; G19's perturbation moved them too, an export writes the last tile's copies again: compare the rest"""
        state = dict(state)
        for key, faces, gw in (("atlas", "faces", "self.atlas_grid_w"), ("atlas_dyn", "faces_dyn", "self.atlas_grid_dyn_w")):
            n, a = state[faces].shape[0] // 2, state[key].clone()
            gh_ = a.shape[-2] // 10
            for k in range(n, gh_ * state[gw]):
                a[..., (k // state[gw]) * 10:(k // state[gw]) * 10 + 10, (k % state[gw]) * 10:(k % state[gw]) * 10 + 10] = 0
            state[key] = a
        return state
    _check_state(referenced(out), referenced(sd), atol=0.0)
    # static tiles are ONE texture: every frame's copy of a static tile is the checkpoint's single static tile
    st = (v.quad_keep & ~v.quad_dyn)
    d, qy, qx = st.nonzero()[0].tolist()
    tile = v.stack.detach()[d, :, qy * 10:(qy + 1) * 10, qx * 10:(qx + 1) * 10]
    assert bool((tile == tile[:1]).all())
    # the lattice reader folds the duplicated border samples: its export differs from the checkpoint by the perturbation's size
    vl = MPMeshVid(RM.mpv_args(5), H, W, ref_extrin, K, 1.0, 100.0)
    vl.init_from_mpi(sd, tile_layout="lattice")
    err = float((vl.reference_state_dict()["atlas_dyn"] - sd["atlas_dyn"]).abs().max())
    assert err > 0.3, err
    # the stage-1 checkpoint of G19 (j): same reader, one frame
    sdj = RM.state_dict_of(g, "j_sd_")
    stj, kj, dj = tiles.stack_from_reference_state(sdj, 40, 60, over["mpi_h_verts"], over["mpi_w_verts"], 1, own_borders=True)
    assert stj.shape == (over["mpi_d"], 1, 40, 60, 4) and torch.equal(kj, v.quad_keep.cpu()) and torch.equal(dj, v.quad_dyn.cpu())


def test_g16_dense_checkpoint_loads_static_as_dynamic():
    """MPV.py:266-288: a dense stage-1 checkpoint becomes the dynamic atlas of every frame.  The product resamples the cell atlas
    (pitch (Aw-1)/(gw*(mpi_w-1))) onto its pitch-1 stack: == grid_sample of the atlas at the reference's UV of every plane pixel."""
    from videoloop3d_amd.MPV import MPMeshVid
    g15, g16 = RM.load("g15_sparsify"), RM.load("g16_init_from_mpi")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    ref = RM.state_dict_of(g16, "dense_")
    assert bool(g16["dense_atlas_dyn_equals_g15_atlas"]) and tuple(g16["dense_atlas_dyn_shape"]) == (4, 4, 80, 240)
    assert ref["faces"].shape == (0, 3) and ref["atlas"].shape == (1, 4, 1, 1) and not ref["self.is_sparse"]       # dummy static lists
    sd = dict(ref, atlas_dyn=torch.from_numpy(g15["in_atlas"]).expand(4, -1, -1, -1))
    v = MPMeshVid(R4.make_args(mpv_frm_num=4, mpv_isloop=True, **over), H, W, ref_extrin, K, 1.0, 100.0)
    v.init_from_mpi(sd)
    assert not v.is_sparse and v.stack.shape == (8, 4, 40, 60, 4)
    atlas = torch.from_numpy(g15["in_atlas"])
    ym, xm = torch.meshgrid(torch.arange(40.), torch.arange(60.), indexing="ij")
    for p in (0, 3, 7):
        u, vv = AO.plane_uv(xm, ym, p, 2, 4, 40, 60)
        want = torch.nn.functional.grid_sample(atlas, torch.stack([u, vv], -1)[None], mode="bilinear", align_corners=True)[0].permute(1, 2, 0)
        assert float((v.stack.detach()[p, 2] - want).abs().max()) <= 1e-4        # fp32 UV arithmetic against an atlas of 240 texels


# ---- G17: render / forward ------------------------------------------------------------------------------------------------------
def _close(a, b, tol, what):
    a, b = torch.as_tensor(a).detach().double(), torch.as_tensor(b).detach().double()
    err = float((a - b).abs().max())
    assert err <= tol, (what, err)


def _close_grad(a, b, what, rel=3e-5):
    """gradients: fp32 sums over thousands of pixels in another order -> relative to the largest entry."""
    _close(a, b, rel * max(1.0, float(torch.as_tensor(b).abs().max())), what)


@pytest.mark.parametrize("tag", ["a", "a2", "a3", "a0"])
def test_g17_mpmesh_dense_forward_oracle(tag):
    """mpv_oracle.mpi_forward on the reference's atlas == the reference's MPMesh.forward: rgb + loop-mask label, every regulariser
    (a2 / a3: l_smooth, edge-weighted d_smooth, bg colour, normalised blend weights), gradients to both textures."""
    g15, g = RM.load("g15_sparsify"), RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    extra_kw = {"a": {}, "a2": dict(l_smooth_loss_weight=0.3, d_smooth_loss_weight=0.1, bg_color="0.2#0.4#0.6"),
                "a3": dict(l_smooth_loss_weight=0.3, d_smooth_loss_weight=0.1, normalize_blendweight_fordepth=True),
                "a0": dict(sparsity_loss_weight=0.0, rgb_smooth_loss_weight=0.0, a_smooth_loss_weight=0.0, density_loss_weight=0.0)}[tag]
    args = R4.make_args(learn_loop_mask=True, **{**over, **RM.REG, **extra_kw})
    h, w, tar_e, K_crop, _ = RM.crop_view(g)
    atlas = torch.from_numpy(g15["in_atlas"]).clone().requires_grad_(True)
    mask = torch.from_numpy(g15["in_atlas_mask"]).clone().requires_grad_(True)
    rgbl, extra = mpv_oracle.mpi_forward(atlas, mask, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop, atlas_grid_h=over["atlas_grid_h"])
    _close(rgbl, g[f"{tag}_rgbl"], 2e-6, "rgbl")
    keys = sorted(k[len(tag) + 7:] for k in g.files if k.startswith(f"{tag}_extra_"))
    assert sorted(extra) == keys and ("l_smooth" in keys) == (tag in ("a2", "a3")) and (keys == []) == (tag == "a0")
    for k in keys:
        _close(extra[k], g[f"{tag}_extra_{k}"], 2e-6 * max(1.0, float(abs(g[f"{tag}_extra_{k}"]).max())), k)
    total = (rgbl * torch.from_numpy(g[f"{tag}_G"])).sum() + sum(getattr(args, k + "_loss_weight") * v.sum() for k, v in extra.items())
    ga, gm = torch.autograd.grad(total, [atlas, mask])
    _close_grad(ga, g[f"{tag}_grad_atlas"], "grad atlas")
    _close_grad(gm, g[f"{tag}_grad_atlas_mask"], "grad mask")
    assert float(abs(g[f"{tag}_grad_atlas_mask"]).max()) > 1e-3


def test_g17_mpmesh_dense_variables_and_eval():
    g15, g = RM.load("g15_sparsify"), RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    args = R4.make_args(learn_loop_mask=True, d_smooth_loss_weight=0.1, **over, **RM.REG)
    h, w, tar_e, K_crop, K_full = RM.crop_view(g)
    atlas, mask = torch.from_numpy(g15["in_atlas"]), torch.from_numpy(g15["in_atlas_mask"])
    rgbl, _ = mpv_oracle.mpi_forward(atlas, mask, args, H, W, ref_extrin, K, 1.0, 100.0, H, W, tar_e, K_full, training=False,
                                     atlas_grid_h=over["atlas_grid_h"])
    _close(rgbl, g["a_eval_rgbl_full"], 2e-6, "eval")
    # the rasteriser's hit count per pixel == the number of planes the analytic coverage test finds, K = its maximum
    p2f = torch.from_numpy(g["a_pix_to_face"])
    assert int(g["a_num_layers"]) == int((p2f >= 0).sum(-1).max()) == p2f.shape[-1]


def _lattice_case(g15, g, which):
    """the sparsified reference checkpoint of case (b) / (d) on the tile lattice + what the reference's atlas gradients mean there."""
    H, W, over, K, ref_extrin, _ = RM.case_A()
    hv, wv, D = over["mpi_h_verts"], over["mpi_w_verts"], over["mpi_d"]
    if which == "b":
        sd, T = RM.state_dict_of(g15, "sd_"), 1
    else:
        sd = RM.state_dict_of(g15, "sd_", atlas_dyn=torch.from_numpy(g["d_atlas_dyn"]))
        T = sd["atlas_dyn"].shape[0]
    st, keep, dyn = tiles.stack_from_reference_state(sd, 40, 60, hv, wv, T)
    g_dyn, g_static = RM.lattice_grad_from_reference(sd, hv, wv, D, T, torch.from_numpy(g[f"{which}_grad_atlas"]),
                                                     torch.from_numpy(g[f"{which}_grad_atlas_dyn"]))
    return st, keep, dyn, g_dyn, g_static


def test_g17_mpmesh_sparsified_forward_oracle():
    """MPI.py:544-548 (static + dynamic face lists) == the oracle on the tile lattice with the quad map: a sample in a culled quad is
    not covered, the smoothness terms run over hit slots."""
    g15, g = RM.load("g15_sparsify"), RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    args = R4.make_args(**over, **RM.REG)
    h, w, tar_e, K_crop, K_full = RM.crop_view(g)
    st, keep, dyn, g_dyn, g_static = _lattice_case(g15, g, "b")
    st = st.requires_grad_(True)
    rgb, extra = mpv_oracle.mpi_forward(st, None, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop, quad_keep=keep)
    _close(rgb, g["b_rgb"], 3e-6, "rgb")
    for k in ("sparsity", "rgb_smooth", "a_smooth", "density"):
        _close(extra[k], g[f"b_extra_{k}"], 3e-6, k)
    assert int(g["b_num_layers"]) < over["mpi_d"]                  # culling made some slot unused everywhere: K < mpi_d enters the means
    total = (rgb * torch.from_numpy(g["b_G"])).sum() + sum(getattr(args, k + "_loss_weight") * v.sum() for k, v in extra.items())
    (gs,) = torch.autograd.grad(total, st)
    _close_grad(gs[:, 0], g_dyn[:, 0] + g_static, "lattice gradient")
    rgb_e, _ = mpv_oracle.mpi_forward(st.detach(), None, args, H, W, ref_extrin, K, 1.0, 100.0, H, W, tar_e, K_full, training=False, quad_keep=keep)
    _close(rgb_e, g["b_eval_rgb_full"], 3e-6, "eval")
    # without the quad map the same stack renders the same IMAGE (culled texels are transparent) but not the same regularisers
    _, extra_nq = mpv_oracle.mpi_forward(st.detach(), None, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop)
    assert abs(float(extra_nq["rgb_smooth"]) - float(g["b_extra_rgb_smooth"].item())) > 1e-3


@pytest.mark.parametrize("which", ["other", "ref", "plain"])
def test_g17_mpmeshvid_dense_forward_oracle(which):
    g = RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    bg = "0.2#0.4#0.6" if which == "ref" else ""
    args = RM.mpv_args(5, bg=bg, regs={} if which == "plain" else RM.REG)
    h, w, tar_e, K_crop, K_full = RM.crop_view(g)
    res = torch.from_numpy(g["res"])
    atlas = torch.from_numpy(g["c_atlas_dyn"]).clone().requires_grad_(True)
    cfg = RM.LOSS_CFGS["other" if which == "plain" else which]
    _, extra = mpv_oracle.mpv_forward(atlas, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop, res=res, losscfg=R4.collate(cfg),
                                      atlas_grid_h=over["atlas_grid_h"])
    keys = sorted(k[len(which) + 9:] for k in g.files if k.startswith(f"c_{which}_extra_"))
    assert sorted(extra) == keys
    for k in keys:
        _close(extra[k], g[f"c_{which}_extra_{k}"], 3e-6 * max(1.0, float(abs(g[f"c_{which}_extra_{k}"]).max())), k)
    total = sum(RM.MPV_WEIGHTS[k] * v.sum() for k, v in extra.items())
    (ga,) = torch.autograd.grad(total, atlas)
    _close_grad(ga, g[f"c_{which}_grad_atlas_dyn"], "grad atlas_dyn")
    rgb, _ = mpv_oracle.mpv_forward(atlas.detach(), args, H, W, ref_extrin, K, 1.0, 100.0, H, W, tar_e, K_full, training=False,
                                    atlas_grid_h=over["atlas_grid_h"])
    _close(rgb, g[f"c_{which}_eval_rgb_full"], 3e-6, "eval")
    rgb_ts, _ = mpv_oracle.mpv_forward(atlas.detach(), args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop, ts=torch.tensor([3, 1]),
                                       training=False, atlas_grid_h=over["atlas_grid_h"])
    _close(rgb_ts, g[f"c_{which}_eval_rgb_crop_ts"], 3e-6, "eval ts")


def test_g17_mpmeshvid_sparsified_forward_oracle():
    g15, g = RM.load("g15_sparsify"), RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    args = RM.mpv_args(5)
    h, w, tar_e, K_crop, K_full = RM.crop_view(g)
    res = torch.from_numpy(g["res"])
    st, keep, dyn, g_dyn, g_static = _lattice_case(g15, g, "d")
    assert st.shape[1] == 5 and float((st[:, 0] - st[:, 1]).abs().max()) > 0.1          # per-frame dynamic content
    st = st.requires_grad_(True)
    _, extra = mpv_oracle.mpv_forward(st, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop, res=res,
                                      losscfg=R4.collate(RM.LOSS_CFGS["other"]), quad_keep=keep)
    for k in ("swd", "sparsity", "rgb_smooth", "a_smooth", "density"):
        _close(extra[k], g[f"d_extra_{k}"], 3e-6 * max(1.0, float(abs(g[f"d_extra_{k}"]).max())), k)
    total = sum(RM.MPV_WEIGHTS[k] * v.sum() for k, v in extra.items())
    (gs,) = torch.autograd.grad(total, st)
    # the static atlas is ONE texture: its gradient is the lattice gradient summed over the frames (static texels), the dynamic atlas' is per frame
    dyn_t = tiles.quad_to_texel_mask(dyn, *st.shape[2:4])
    keep_t = tiles.quad_to_texel_mask(keep, *st.shape[2:4])
    only_dyn = dyn_t & ~tiles.quad_to_texel_mask(keep & ~dyn, *st.shape[2:4])
    scale = max(1.0, float(g_dyn.abs().max()))
    _close(gs.sum(1), g_dyn.sum(1) + g_static, 3e-5 * scale, "frame-summed lattice gradient")
    _close(gs[only_dyn[:, None].expand(-1, 5, -1, -1)], g_dyn[only_dyn[:, None].expand(-1, 5, -1, -1)], 3e-5 * scale, "dynamic texels per frame")
    assert float(gs[~keep_t[:, None].expand(-1, 5, -1, -1)].abs().max()) == 0.0
    rgb, _ = mpv_oracle.mpv_forward(st.detach(), args, H, W, ref_extrin, K, 1.0, 100.0, H, W, tar_e, K_full, training=False, quad_keep=keep)
    _close(rgb, g["d_eval_rgb_full"], 3e-6, "eval")


def test_g19_trained_tiles_forward_oracle():
    """the oracle's tile-exact layout (mpi_oracle.sample_layers with RenderSpec.tile: corner UVs on tile-corner texel centres + the hard cut)
    against the reference's OWN forward on a trained checkpoint (golden G19 h / j): every `extra` term, the gradient of every tile texel of both
    atlases, the evaluation renders -- and the number the shared-border lattice of rounds 4-5 misses this checkpoint by."""
    g = RM.load("g19_trained_tiles")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    hv, wv, D = over["mpi_h_verts"], over["mpi_w_verts"], over["mpi_d"]
    args = RM.mpv_args(5)
    h, w, tar_e, K_crop, K_full = RM.crop_view(g)
    res = torch.from_numpy(g["res"])
    sd = RM.state_dict_of(g, "h_sd_")
    st, keep, dyn = tiles.stack_from_reference_state(sd, 40, 60, hv, wv, 5, own_borders=True)
    tile = (10, 10)
    st = st.requires_grad_(True)
    _, extra = mpv_oracle.mpv_forward(st, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop, res=res,
                                      losscfg=R4.collate(RM.LOSS_CFGS["other"]), quad_keep=keep, tile=tile)
    for k in ("swd", "sparsity", "rgb_smooth", "a_smooth", "density"):
        _close(extra[k], g[f"h_extra_{k}"], 3e-6 * max(1.0, float(abs(g[f"h_extra_{k}"]).max())), k)
    (gs,) = torch.autograd.grad(sum(RM.MPV_WEIGHTS[k] * v.sum() for k, v in extra.items()), st)
    g_dyn, g_static = RM.own_grad_from_reference(sd, hv, wv, D, 5, torch.from_numpy(g["h_grad_atlas"]), torch.from_numpy(g["h_grad_atlas_dyn"]))
    dyn_t = tiles.quad_to_texel_mask(dyn, 40, 60, tile)
    static_t = tiles.quad_to_texel_mask(keep & ~dyn, 40, 60, tile)
    scale = float(g_dyn.abs().max())                      # (the means of the loss make the gradients ~1e-4: tolerances RELATIVE to their size)
    assert scale > 1e-5 and float(g_static.abs().max()) > 1e-5
    _close(gs[dyn_t[:, None].expand(-1, 5, -1, -1)], g_dyn[dyn_t[:, None].expand(-1, 5, -1, -1)], 1e-4 * scale, "dynamic tile texels, per frame")
    _close(gs.sum(1)[static_t], g_static[static_t], 1e-4 * scale, "static tile texels (one texture: summed over the frames)")
    assert float(gs[(~(dyn_t | static_t))[:, None].expand(-1, 5, -1, -1)].abs().max()) == 0.0
    rgb, _ = mpv_oracle.mpv_forward(st.detach(), args, H, W, ref_extrin, K, 1.0, 100.0, H, W, tar_e, K_full, training=False, quad_keep=keep, tile=tile)
    _close(rgb, g["h_eval_rgb_full"], 3e-6, "eval")
    rgb2, _ = mpv_oracle.mpv_forward(st.detach(), args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop, ts=torch.tensor([3, 1]), training=False,
                                     quad_keep=keep, tile=tile)
    _close(rgb2, g["h_eval_rgb_crop_ts"], 3e-6, "eval crop, frames [3, 1]")
    # what the shared-border lattice (the round 4-5 reader: duplicated border samples folded into one texel) makes of this checkpoint
    stl, _, _ = tiles.stack_from_reference_state(sd, 40, 60, hv, wv, 5)
    rgb_l, _ = mpv_oracle.mpv_forward(stl, args, H, W, ref_extrin, K, 1.0, 100.0, H, W, tar_e, K_full, training=False, quad_keep=keep)
    err = float((rgb_l - torch.from_numpy(g["h_eval_rgb_full"])).abs().max())
    assert err > 0.02, err                               # (measured 0.2-0.3 here: DESIGN.md section 4 quotes it)
    # (j) the stage-1 model after the switch-over: MPMesh.forward on tiles
    sdj = RM.state_dict_of(g, "j_sd_")
    stj, kj, dj = tiles.stack_from_reference_state(sdj, 40, 60, hv, wv, 1, own_borders=True)
    stj = stj.requires_grad_(True)
    argsj = RM.mpi_args(**RM.REG)
    rgbj, extraj = mpv_oracle.mpi_forward(stj, None, argsj, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop, quad_keep=kj, tile=tile)
    _close(rgbj, g["j_rgb"], 3e-6, "stage-1 rgb")
    for k in ("sparsity", "rgb_smooth", "a_smooth", "density"):
        _close(extraj[k], g[f"j_extra_{k}"], 3e-6 * max(1.0, float(abs(g[f"j_extra_{k}"]).max())), "stage-1 " + k)
    totalj = (rgbj * torch.from_numpy(g["j_G"])).sum() + sum(getattr(argsj, k + "_loss_weight") * v.sum() for k, v in extraj.items())
    (gj,) = torch.autograd.grad(totalj, stj)
    gjd, gjs = RM.own_grad_from_reference(sdj, hv, wv, D, 1, torch.from_numpy(g["j_grad_atlas"]), torch.from_numpy(g["j_grad_atlas_dyn"]))
    _close(gj[:, 0], gjd[:, 0] + gjs, 1e-4 * float((gjd[:, 0] + gjs).abs().max()), "stage-1 tile gradient")
    print(f"lattice reader on the trained checkpoint G19: max |image error| = {err:.3f}")


def test_g17_second_layout_dense_mpmeshvid_oracle():
    """golden (e): a 3 x 2 atlas of cells, non-square plane scales, normalize_verts, another view -- the cell pitch / bleed of the atlas
    sampling and the geometry do not depend on the one layout the other cases use."""
    g = RM.load("g17_forward")
    H, W, over = R4.SHAPES["B"]
    K, ref_extrin, _ = R4.scene(H, W, angle_deg=-1.7, trans=(-0.031, 0.022, -0.004))
    args = R4.make_args(mpv_frm_num=4, mpv_isloop=True, init_std=0.5, scale_invariant=True, swd_patch_size=3, swd_patcht_size=3,
                        swd_stride=2, swd_stridet=1, **over)
    h, w = (int(v) for v in g["e_hw_crop"])
    tar_e, K_crop, K_full = (torch.from_numpy(g[k]) for k in ("e_tar_extrin", "e_tar_intrin_crop", "e_tar_intrin_full"))
    atlas = torch.from_numpy(g["e_atlas_dyn"]).clone().requires_grad_(True)
    _, extra = mpv_oracle.mpv_forward(atlas, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop, res=torch.from_numpy(g["e_res"]),
                                      losscfg=R4.collate(RM.LOSS_CFGS["other"]), atlas_grid_h=over["atlas_grid_h"])
    assert sorted(extra) == ["swd"]
    _close(extra["swd"], g["e_extra_swd"], 3e-6, "swd")
    (ga,) = torch.autograd.grad(extra["swd"].sum(), atlas)
    _close_grad(ga, g["e_grad_atlas_dyn"], "grad atlas_dyn")
    rgb, _ = mpv_oracle.mpv_forward(atlas.detach(), args, H, W, ref_extrin, K, 1.0, 100.0, H, W, tar_e, K_full, training=False,
                                    atlas_grid_h=over["atlas_grid_h"])
    _close(rgb, g["e_eval_rgb_full"], 3e-6, "eval")
    rgb_ts, _ = mpv_oracle.mpv_forward(atlas.detach(), args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop, ts=torch.tensor([2, 0]),
                                       training=False, atlas_grid_h=over["atlas_grid_h"])
    _close(rgb_ts, g["e_eval_rgb_crop_ts"], 3e-6, "eval ts")


@pytest.mark.parametrize("H,W,scale,hv,wv,D,gh,seed", [(36, 60, 1.0, 4, 6, 6, 2, 1),        # 12 x 12-texel tiles, 2 x 3 cells
                                                       (40, 40, 1.2, 5, 5, 4, 2, 2),        # square planes, 2 x 2 cells
                                                       (33, 65, 1.0, 3, 5, 6, 3, 3),        # odd sizes
                                                       (48, 64, 1.5, 7, 9, 8, 4, 4)])       # the shipped 1.5-1.6x plane scale, more quads
def test_reader_and_exporter_round_trip_oracle_checkpoints(H, W, scale, hv, wv, D, gh, seed):
    """The G15 round trip on other shapes: oracle/ckpt_oracle.sparsify_atlas (pinned to the reference by G15) sparsifies a random smooth atlas,
    the product reads the checkpoint onto its tile lattice and writes it back -- same quads, same atlases, same scalars."""
    from videoloop3d_amd.MPI import MPMesh
    args = R4.make_args(learn_loop_mask=True, mpi_h_scale=scale, mpi_w_scale=scale, mpi_h_verts=hv, mpi_w_verts=wv, mpi_d=D, atlas_grid_h=gh)
    K, ref_extrin, _ = R4.scene(H, W)
    m = MPMesh(args, H, W, ref_extrin, K, 1.0, 100.0)
    mh, mw, gw = m.mpi_h, m.mpi_w, D // gh
    th, _ = tiles.tile_lattice(gh, hv, gh * mh)
    tw, _ = tiles.tile_lattice(gw, wv, gw * mw)
    if th != tw:
        pytest.skip(f"tiles of {th} x {tw} texels: the reference's gen_quad_uvs steps both axes by the tile HEIGHT (MPI.py:409-410), square tiles only")
    atlas = torch.cat([R4.smooth_field((3, gh * mh, gw * mw), seed=100 + seed, scale=5.0) * 2.0,
                       R4.smooth_field((1, gh * mh, gw * mw), seed=200 + seed, scale=8.0) * 6.0 - 4.0], 0)[None].contiguous()
    mask = (R4.smooth_field((1, gh * mh, gw * mw), seed=300 + seed, scale=12.0) * 5.0 - 0.5)[None].contiguous()
    sd = CO.sparsify_atlas(atlas, mask, gh, D, hv, wv, 2, 0.05, 0.5)
    n_s, n_d = sd["faces"].shape[0] // 2, sd["faces_dyn"].shape[0] // 2
    if min(n_s, n_d) < 4:
        pytest.skip("too few static / dynamic quads for the reference's atlas grid rule")
    sd.update({"planedepth": m.planedepth.clone(), "ref_extrin": m.ref_extrin.clone(), "ref_intrin": m.ref_intrin.clone(),
               "_verts": gen_mpi_vertices(mh, mw, m.ref_intrin_mpi, hv, wv, m.planedepth)})
    # the product's own sparsify on the same weights keeps the same quads
    with torch.no_grad():
        m.stack.copy_(atlas_to_stack(atlas, D, gh))
        m.stack_mask.copy_(atlas_to_stack(mask, D, gh)[..., 0])
    m2 = MPMesh(copy.copy(args), H, W, ref_extrin, K, 1.0, 100.0)
    m2.init_from_mpi(sd, tile_layout="lattice")
    assert m2.stack.shape[2:4] == ((hv - 1) * (th - 1) + 1, (wv - 1) * (tw - 1) + 1)
    # ... and in the default, tile-exact layout the checkpoint comes back bit for bit
    m3 = MPMesh(copy.copy(args), H, W, ref_extrin, K, 1.0, 100.0)
    m3.init_from_mpi(sd)
    assert m3.tile_own == (th, tw) and m3.stack.shape[2:4] == ((hv - 1) * th, (wv - 1) * tw)
    _check_state(m3.reference_state_dict(), sd, atol=0.0)
    m.sparsify_faces(erode_num=2, alpha_thresh=0.05, loop_thresh=0.5)
    assert torch.equal(m.quad_keep, m2.quad_keep) and torch.equal(m.quad_dyn, m2.quad_dyn)
    assert int(m2.quad_keep.sum()) == n_s + n_d and int(m2.quad_dyn.sum()) == n_d
    # (duplicated border samples of neighbouring tiles differ by the fp32 jitter of their UVs times the atlas' slope -- logits of +-8 over
    #  a few texels here -- and the lattice holds their mean)
    _check_state(m2.reference_state_dict(), sd, atol=5e-4)


@pytest.mark.parametrize("name", ["gpnn", "mse", "avg"])
def test_g17_other_loss_entries_oracle(name):
    """golden (f): the other entries of MPMeshVid.losses through forward -- 'gpnn' (the parser's default: the direct loss), 'mse', 'avg'."""
    g = RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    args = RM.mpv_args(5, regs={})
    h, w, tar_e, K_crop, _ = RM.crop_view(g)
    atlas = torch.from_numpy(g["c_atlas_dyn"]).clone().requires_grad_(True)
    _, extra = mpv_oracle.mpv_forward(atlas, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop, res=torch.from_numpy(g["res"]),
                                      losscfg=R4.collate(RM.OTHER_LOSSES[name]), atlas_grid_h=over["atlas_grid_h"])
    _close(extra["swd"], g[f"f_{name}_extra_swd"], 3e-6, "swd")
    (ga,) = torch.autograd.grad(extra["swd"].sum(), atlas)
    _close_grad(ga, g[f"f_{name}_grad_atlas_dyn"], "grad atlas_dyn")


def test_g17_no_loop_padding_no_gain_oracle():
    """golden (g): mpv_isloop off (no frames appended, MPV.py:488-492) and scale_invariant off (MPV.py:499-504)."""
    g = RM.load("g17_forward")
    H, W, over, K, ref_extrin, _ = RM.case_A()
    args = RM.mpv_args(5, regs={})
    args.mpv_isloop, args.scale_invariant = False, False
    h, w, tar_e, K_crop, _ = RM.crop_view(g)
    atlas = torch.from_numpy(g["c_atlas_dyn"]).clone().requires_grad_(True)
    _, extra = mpv_oracle.mpv_forward(atlas, args, H, W, ref_extrin, K, 1.0, 100.0, h, w, tar_e, K_crop, res=torch.from_numpy(g["res"]),
                                      losscfg=R4.collate(RM.LOSS_CFGS["other"]), atlas_grid_h=over["atlas_grid_h"])
    _close(extra["swd"], g["g_extra_swd"], 3e-6, "swd")
    (ga,) = torch.autograd.grad(extra["swd"].sum(), atlas)
    _close_grad(ga, g["g_grad_atlas_dyn"], "grad atlas_dyn")
    assert abs(float(g["g_extra_swd"].item()) - float(g["c_plain_extra_swd"].item())) > 1e-4
