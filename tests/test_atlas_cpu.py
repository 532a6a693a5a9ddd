"""Atlas-cell sampling of the reference (MPV.py:37-44, 75-81, 394-439) on the dense stack: host logic of videoloop3d_amd/atlas.py
(per-plane records + apron copies) against oracle/atlas_oracle.py, which samples the ATLAS itself with grid_sample.  CPU only: the
per-plane render is emulated with the oracle's bilinear tap rule; the HIP kernels are checked in tests/test_gpu_atlas.py."""
import torch

from oracle import atlas_oracle as AO
from oracle import mpi_oracle as MO
from videoloop3d_amd import atlas as A
from videoloop3d_amd import synth
from videoloop3d_amd.MPV import atlas_to_stack, stack_to_atlas


def scene(D=8, grid_h=2, mh=22, mw=31, H=26, W=36, T=2, seed=5):
    from videoloop3d_amd.utils_mpi import compute_homography, make_depths
    grid_w = D // grid_h
    atlas = (synth.hash_uniform((T, 4, grid_h * mh, grid_w * mw), seed=seed) * 4 - 2)
    atlas[:, 3] -= 1.0
    ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
    tar_e = tar_e.clone()
    tar_e[:3, 3] *= 3.0
    homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                               make_depths(D, 1.0, 100.0).flip(0)[None])[0]
    # plane centred in a slightly LARGER frame: every plane's quad border (hard cut + neighbour-cell bleed) is inside the image
    homos = torch.tensor([[1.0, 0, (mw - W) / 2], [0, 1.0, (mh - H) / 2], [0, 0, 1.0]]) @ homos
    return atlas, homos, (D, grid_h, mh, mw, H, W, T)


def emulate_per_plane_render(padded, rec, H, W, pc=0.5):
    """what the VL3D_COORD_AFFINE_PLANES kernels compute, with the oracle's tap rule: per plane its own matrix and coverage box."""
    D, T = padded.shape[:2]
    layers = []
    for p in range(D):
        M = rec[p, :9].reshape(1, 3, 3)
        tx, ty = MO._homography_source_coords(H, W, M, pc)
        img = padded[p].permute(0, 3, 1, 2)                                  # T,4,h,w
        samp = MO._bilinear_zeros(img, tx.expand(T, H, W), ty.expand(T, H, W))
        cov = ((tx >= rec[p, 9]) & (tx <= rec[p, 10]) & (ty >= rec[p, 11]) & (ty <= rec[p, 12])).to(img.dtype)
        layers.append(torch.sigmoid(samp) * cov[:, None])
    layers = torch.stack(layers, -1).permute(0, 2, 3, 4, 1)
    rgb, bw = MO.overcompose(layers[..., 3], layers[..., :3])
    return rgb, bw.sum(-1)


def test_vertex_uvs_closed_form():
    """plane_uv == affine interpolation of the per-vertex UVs MPV.py:75-81 builds (re-executed literally in the oracle)."""
    gh, gw, hv, wv, mh, mw = 2, 4, 5, 7, 20, 30
    uv = AO.reference_vertex_uvs(gh, gw, hv, wv)
    for p in range(gh * gw):
        vy, vx = torch.meshgrid(torch.arange(hv), torch.arange(wv), indexing="ij")
        u, v = AO.plane_uv(vx * (mw - 1) / (wv - 1), vy * (mh - 1) / (hv - 1), p, gh, gw, mh, mw)
        assert float((u.reshape(-1) - uv[p, :, 0]).abs().max()) <= 1e-6 and float((v.reshape(-1) - uv[p, :, 1]).abs().max()) <= 1e-6


def test_apron_stack_and_plane_records_reproduce_the_atlas_sampling():
    """identical MPV weights: atlas -> atlas_to_stack -> aprons + per-plane records  ==  grid_sample on the atlas through the
    reference's UV layout, forward and gradient w.r.t. the atlas, in fp64 (exact statement) and fp32."""
    atlas, homos, (D, gh, mh, mw, H, W, T) = scene()
    for dt, tol in ((torch.float64, 1e-9), (torch.float32, 2e-5)):
        a_ref = atlas.to(dt).requires_grad_(True)
        rgb_o, alpha_o, layers = AO.render_atlas(a_ref, homos.to(dt), H, W, gh, mh, mw)
        g = (synth.hash_uniform(tuple(rgb_o.shape), seed=9) - 0.5).to(dt)
        (ga_o,) = torch.autograd.grad((rgb_o * g).sum() + 0.3 * alpha_o.sum(), a_ref)
        a_in = atlas.to(dt).requires_grad_(True)
        padded = A.stack_with_aprons(atlas_to_stack(a_in, D, gh), gh)
        rec = A.plane_records(homos.double(), gh, mh, mw).to(dt) if dt == torch.float32 else _records64(homos, gh, mh, mw)
        rgb, alpha = emulate_per_plane_render(padded, rec, H, W)
        (ga,) = torch.autograd.grad((rgb * g).sum() + 0.3 * alpha.sum(), a_in)
        assert float((rgb - rgb_o).abs().max()) <= tol and float((alpha - alpha_o).abs().max()) <= tol
        assert float((ga - ga_o).abs().max()) <= tol * max(1.0, float(ga_o.abs().max()))
    # the neighbour-cell bleed is really exercised: some covered sample lies beyond the last texel centre of its own cell
    xs, ys = MO._homography_source_coords(H, W, homos, 0.5)
    px, py = (D // gh * mw - 1) / (D // gh * (mw - 1)), (gh * mh - 1) / (gh * (mh - 1))
    assert bool(((xs[0] * px > mw - 1) & (xs[0] <= mw - 1) & (ys[0] >= 0) & (ys[0] <= mh - 1)).any())          # right bleed of cell (0,0)
    assert bool(((xs[1] * px - 1 / (D // gh) < 0) & (xs[1] >= 0) & (ys[1] >= 0) & (ys[1] <= mh - 1)).any())     # left bleed of cell (0,1)
    assert bool(((ys[0] * py > mh - 1) & (ys[0] <= mh - 1) & (xs[0] >= 0) & (xs[0] <= mw - 1)).any())          # bottom bleed
    # and a pitch-1 render of the same weights differs visibly (what the exact mode is for)
    from videoloop3d_amd.MPV import atlas_to_stack as a2s
    plain, _, _ = MO.render_planes(a2s(atlas, D, gh), homos, H, W, MO.RenderSpec(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post"))
    assert float((plain - rgb_o.float()).abs().max()) > 1e-3


def _records64(homos, gh, mh, mw):
    """plane_records without the final cast to fp32 (the module returns what the kernels take)."""
    D = homos.shape[0]
    gw = D // gh
    px, py = (gw * mw - 1) / (gw * (mw - 1)), (gh * mh - 1) / (gh * (mh - 1))
    rec = torch.zeros((D, 16), dtype=torch.float64)
    for p in range(D):
        i, j = p // gw, p % gw
        ox, oy = -j / gw + 1, -i / gh + 1
        Am = torch.tensor([[px, 0.0, ox], [0.0, py, oy], [0.0, 0.0, 1.0]], dtype=torch.float64)
        rec[p, :9] = (Am @ homos[p].double()).reshape(9)
        rec[p, 9:13] = torch.tensor([ox, ox + px * (mw - 1), oy, oy + py * (mh - 1)], dtype=torch.float64)
    return rec


def test_plane_records_match_the_fp64_statement():
    _, homos, (D, gh, mh, mw, H, W, T) = scene()
    assert float((A.plane_records(homos, gh, mh, mw).double() - _records64(homos, gh, mh, mw)).abs().max()) <= 2e-5


def test_stack_atlas_roundtrip_and_apron_gradient():
    atlas, _, (D, gh, mh, mw, H, W, T) = scene()
    st = atlas_to_stack(atlas, D, gh)
    assert torch.equal(stack_to_atlas(st, gh), atlas)
    s = st.clone().requires_grad_(True)
    P = A.stack_with_aprons(s, gh)
    assert P.shape == (D, T, mh + 2, mw + 2, 4) and torch.equal(P[:, :, 1:-1, 1:-1], st)
    gw = D // gh
    assert torch.equal(P[0, :, 1:-1, -1], st[1, :, :, 0]) and torch.equal(P[1, :, 1:-1, 0], st[0, :, :, -1])
    assert torch.equal(P[0, :, -1, 1:-1], st[gw, :, 0, :]) and torch.equal(P[0, :, -1, -1], st[gw + 1, :, 0, 0])
    assert float(P[gw - 1, :, :, -1].abs().max()) == 0 and float(P[0, :, 0].abs().max()) == 0          # the atlas ends there
    (g,) = torch.autograd.grad(P.sum(), s)
    # a texel mirrored into k aprons receives 1 + k
    assert float(g[1, 0, 5, 0, 0]) == 2.0 and float(g[0, 0, 5, 0, 0]) == 1.0 and float(g[gw + 1, 0, 0, 0, 0]) == 4.0
