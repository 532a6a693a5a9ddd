"""The TILE-EXACT layout of a tile-culled model (include/vl3d.h "Tile-exact layout"; MPI.py:380-418: every kept quad is a tile with its OWN
border row / column; MPV.py:394-427: a face's UVs span exactly its tile) -- kernel level: the HIP render / backward / regularisers on planes of
QH th x QW tw texels against the CPU oracle, which states the layout the reference's way (corner UVs on tile-corner texel centres +
grid_sample, oracle/mpi_oracle.sample_layers).  Module level (the reference's own forward on a TRAINED checkpoint, golden G19):
tests/test_gpu_reference_modules.py."""
import math

import pytest
import torch

from oracle import mpi_oracle as MO
from videoloop3d_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import __graft_entry__ as g
    g.build()
    return torch.device("cuda:0")


def maxabs(a, b):
    return float((a.detach().double().cpu() - torch.as_tensor(b).double().cpu()).abs().max())


def _tile_ran():
    from videoloop3d_amd import render as R
    return int(R.LAST_BWD_SCRATCH[:1].view(torch.int32).item())


def scene(D, H, W, QH, QW, th, tw, mpi_scale=1.1, seed=3, keep_frac=0.6, skew=True):
    """planes of (mpi_scale H) x (mpi_scale W) plane pixels carrying QH x QW tiles of th x tw texels; a view near the reference camera."""
    from videoloop3d_amd.utils_mpi import compute_homography, make_depths
    ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
    tar_e = tar_e.clone()
    tar_e[:3, 3] *= 1.4
    mpi_h, mpi_w = int(mpi_scale * H), int(mpi_scale * W)
    Kr = Kr.clone()
    Kr[0, 2] += (mpi_w - W) // 2
    Kr[1, 2] += (mpi_h - H) // 2
    depths = make_depths(D, 1.0, 100.0).flip(0)
    homos = compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3), depths[None])[0]
    if skew:
        a = math.radians(1.5)
        homos = homos @ torch.tensor([[math.cos(a) * 1.03, -math.sin(a), 1.7], [math.sin(a), math.cos(a) * 0.97, 1.2], [1e-5, -2e-5, 1.0]])
    sx, sy = QW * (tw - 1) / (mpi_w - 1), QH * (th - 1) / (mpi_h - 1)          # plane pixel -> LATTICE coordinate
    torch.manual_seed(seed)
    keep = torch.rand(D, QH, QW) < keep_frac
    return homos, (sx, sy), keep


def specs(scale, tile, variant=0, offset=(0.0, 0.0)):
    from videoloop3d_amd.render import RenderSpec
    import dataclasses
    p = dataclasses.replace(RenderSpec.mpv(scale=scale, offset=offset, variant=variant), tile=tile)
    o = MO.RenderSpec(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post", scale=scale, offset=offset, tile=tile)
    return p, o


@pytest.mark.parametrize("variant", [0, 1, 3, 5])
@pytest.mark.parametrize("T,th,tw,QH,QW,keep_frac", [(1, 16, 16, 9, 13, 0.6), (3, 14, 13, 9, 13, 0.35), (2, 4, 5, 30, 36, 1.0), (2, 2, 2, 40, 60, 0.7)])
def test_tile_exact_render_and_backward_match_the_oracle(dev, variant, T, th, tw, QH, QW, keep_frac):
    """forward + every backward kernel (owner-computes in 64- / 32-wide regions, atomics) on a stack whose tiles are INDEPENDENT (the two copies
    of a border sample differ): image, alpha and the gradient of every tile texel; culled tiles get exactly zero."""
    from videoloop3d_amd import tiles
    from videoloop3d_amd.render import render_planes
    D = 6
    H, W = int(QH * (th - 1) / 1.15), int(QW * (tw - 1) / 1.15)      # a tile lattice ~1.05x the planes' pixels: the owner-computes kernels' range
    homos, scale, keep = scene(D, H, W, QH, QW, th, tw, keep_frac=keep_frac)
    Hs, Ws = QH * th, QW * tw
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=19)
    sp, so = specs(scale, (th, tw), variant)
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    g_a = synth.hash_uniform((T, H, W), seed=6) - 0.5
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, _ = MO.render_planes(s_cpu, homos, H, W, so, quad_keep=keep)
    (gs_o,) = torch.autograd.grad([rgb_o, alpha_o], s_cpu, [g_rgb, g_a])
    s_gpu = stack.to(dev).requires_grad_(True)
    rgb, alpha = render_planes(s_gpu, homos.to(dev), H, W, sp, quad_keep=keep.to(dev))
    (gs,) = torch.autograd.grad([rgb, alpha], s_gpu, [g_rgb.to(dev), g_a.to(dev)])
    assert _tile_ran() == (0 if variant == 1 else 1)
    assert float(alpha_o.detach().max()) > 0.3
    assert maxabs(rgb, rgb_o) <= TOL and maxabs(alpha, alpha_o) <= TOL
    assert maxabs(gs, gs_o) <= TOL * max(1.0, float(gs_o.abs().max()))
    dead = ~tiles.quad_to_texel_mask(keep, Hs, Ws, (th, tw))
    if dead.any():
        assert float(gs.cpu()[dead[:, None].expand(D, T, Hs, Ws)].abs().max()) == 0.0


def test_tile_exact_equals_the_shared_border_lattice_on_duplicated_borders(dev):
    """a lattice stack (neighbouring quads share their border texels) exploded into tiles renders the same image, and the tile gradient summed
    back onto the lattice is the lattice gradient: the layout changes WHERE the two copies live, nothing else."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, H, W, QH, QW, th, tw = 5, 2, 96, 150, 7, 11, 12, 12
    homos, scale, keep = scene(D, H, W, QH, QW, th, tw, keep_frac=0.55)
    Hl, Wl = QH * (th - 1) + 1, QW * (tw - 1) + 1
    lat = synth.make_plane_stack(D, T, Hl, Wl, seed=23).to(dev)
    iy, ix = torch.arange(QH * th, device=dev), torch.arange(QW * tw, device=dev)
    ly, lx = (iy // th) * (th - 1) + iy % th, (ix // tw) * (tw - 1) + ix % tw
    own = lat[:, :, ly][:, :, :, lx].contiguous().requires_grad_(True)
    lat = lat.requires_grad_(True)
    g_rgb = (synth.hash_uniform((T, H, W, 3), seed=5) - 0.5).to(dev)
    sp, _ = specs(scale, (th, tw))
    r0, a0 = render_planes(lat, homos.to(dev), H, W, RenderSpec.mpv(scale=scale), quad_keep=keep.to(dev))
    r1, a1 = render_planes(own, homos.to(dev), H, W, sp, quad_keep=keep.to(dev))
    assert maxabs(r0, r1) <= 1e-5 and maxabs(a0, a1) <= 1e-5      # (the tile coordinate is the lattice coordinate + an integer: one more fp32 rounding)
    (g0,) = torch.autograd.grad(r0, lat, g_rgb)
    (g1,) = torch.autograd.grad(r1, own, g_rgb)
    back = torch.zeros_like(g0).index_add_(2, ly, torch.zeros((D, T, QH * th, Wl, 4), device=dev).index_add_(3, lx, g1))      # both copies of a border texel onto it
    assert maxabs(back, g0) <= 2e-5 * max(1.0, float(g0.abs().max()))


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("T", [1, 3])
def test_tile_exact_layer_regularisers_match_the_oracle(dev, variant, T):
    """rgb_smooth / a_smooth in hit-slot order (MPV.py:517-531) and the sparsity sums on tile-exact planes: one-pass culled forward, both backwards."""
    from videoloop3d_amd.render import render_planes_with_regularisers
    D, H, W, QH, QW, th, tw = 5, 83, 111, 6, 9, 10, 10
    homos, scale, keep = scene(D, H, W, QH, QW, th, tw, keep_frac=0.5, seed=5)
    stack = synth.make_plane_stack(D, T, QH * th, QW * tw, seed=17)
    sp, so = specs(scale, (th, tw), variant)
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, _, layers = MO.render_planes(s_cpu, homos, H, W, so, return_layers=True, quad_keep=keep)
    sums_o = torch.stack([(layers[:, :, 1:, :, :3] - layers[:, :, :-1, :, :3]).abs().sum(), (layers[:, 1:, :, :, :3] - layers[:, :-1, :, :, :3]).abs().sum(),
                          (layers[:, :, 1:, :, 3] - layers[:, :, :-1, :, 3]).abs().sum(), (layers[:, 1:, :, :, 3] - layers[:, :-1, :, :, 3]).abs().sum()])
    wts = torch.tensor([1e-4, 2e-4, 3e-4, 4e-4])
    (gs_o,) = torch.autograd.grad((rgb_o * g_rgb).sum() + (sums_o * wts).sum(), s_cpu)
    s_gpu = stack.to(dev).requires_grad_(True)
    rgb, alpha, sums, _ = render_planes_with_regularisers(s_gpu, homos.to(dev), H, W, sp, quad_keep=keep.to(dev))
    (gs,) = torch.autograd.grad((rgb * g_rgb.to(dev)).sum() + (sums * wts.to(dev)).sum(), s_gpu)
    assert maxabs(rgb, rgb_o) <= TOL
    assert float(((sums.cpu() - sums_o).abs() / sums_o.abs().clamp_min(1.0)).max()) <= 1e-4
    assert maxabs(gs, gs_o) <= TOL * max(1.0, float(gs_o.abs().max()))


@pytest.mark.parametrize("variant", [0, 1])
def test_tile_exact_window_of_a_plane(dev, variant):
    """the stack is a texel WINDOW of the tile-exact plane (crop-aware training: optim.WindowAdam's compact copy), window origin off the tile
    grid: image and the window's gradient equal the oracle's on the whole plane."""
    from videoloop3d_amd.render import render_planes
    D, T, QH, QW, th, tw = 5, 2, 8, 12, 10, 10
    Hf, Wf = 100, 150                      # the frame the planes were built for; the view is a crop of it
    h, w, h0, w0 = 48, 72, 30, 51
    homos, scale, keep = scene(D, Hf, Wf, QH, QW, th, tw, keep_frac=0.6, seed=7, skew=False)
    shift = torch.tensor([[1.0, 0, float(w0)], [0, 1.0, float(h0)], [0, 0, 1.0]])       # crop pixel -> frame pixel
    homos = homos @ shift
    Hs, Ws = QH * th, QW * tw
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=31)
    _, so = specs(scale, (th, tw))
    g_rgb = synth.hash_uniform((T, h, w, 3), seed=5) - 0.5
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, _ = MO.render_planes(s_cpu, homos, h, w, so, quad_keep=keep)
    (gs_o,) = torch.autograd.grad(rgb_o, s_cpu, g_rgb)
    nz = gs_o.abs().sum((0, 1, 4)) > 0
    ys, xs = nz.any(1).nonzero().flatten(), nz.any(0).nonzero().flatten()
    y0, y1 = max(int(ys.min()) - 3, 0) // 8 * 8, min(-(-(int(ys.max()) + 4) // 8) * 8, Hs)
    x0, x1 = max(int(xs.min()) - 3, 0) // 8 * 8, min(-(-(int(xs.max()) + 4) // 8) * 8, Ws)
    assert (y1 - y0) < Hs and (x1 - x0) < Ws and (y0 % th or x0 % tw)      # a proper window, not on the tile grid
    sp, _ = specs(scale, (th, tw), variant, offset=(-float(x0), -float(y0)))
    win = stack[:, :, y0:y1, x0:x1].contiguous().to(dev).requires_grad_(True)
    rgb, alpha = render_planes(win, homos.to(dev), h, w, sp, quad_keep=keep.to(dev), cull_window=(y0, x0, Hs, Ws))
    (gs,) = torch.autograd.grad(rgb, win, g_rgb.to(dev))
    assert maxabs(rgb, rgb_o) <= TOL and maxabs(alpha, alpha_o) <= TOL
    assert maxabs(gs, gs_o[:, :, y0:y1, x0:x1]) <= TOL * max(1.0, float(gs_o.abs().max()))


import os
_EXTRA = int(os.environ.get("VL3D_FUZZ_EXTRA", "0"))      # soak runs widen the fuzzer (tests/test_gpu_fuzz.py)


@pytest.mark.parametrize("seed", list(range(24 + _EXTRA)))
def test_tile_exact_fuzz(dev, seed):
    """random tile sizes (2 .. 17 per axis, anisotropic), quad grids, keep maps, rotated / scaled / perspective views (owner-computes path where its
    plan allows, atomics elsewhere), every second case a texel WINDOW of the planes at an origin off the tile grid, every third with the layer
    regularisers: image, alpha, sums and the gradient of every tile texel against the oracle."""
    import dataclasses
    from videoloop3d_amd.render import RenderSpec, render_planes, render_planes_with_regularisers
    g = torch.Generator().manual_seed(500 + seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    u = lambda lo, hi: float(torch.rand(1, generator=g)) * (hi - lo) + lo
    D, T = r(1, 5), r(1, 3)
    th, tw = r(2, 17), r(2, 17)
    QH, QW = r(1, max(1, 96 // th)), r(1, max(1, 140 // tw))
    Hs, Ws = QH * th, QW * tw
    Ll_h, Ll_w = QH * (th - 1), QW * (tw - 1)                      # lattice extent
    H, W = max(1, int(Ll_h * u(0.5, 1.3))), max(1, int(Ll_w * u(0.5, 1.3)))
    zoom = u(0.7, 1.5) if seed % 3 else u(0.97, 1.03)
    a = math.radians(u(-6, 6))
    base = torch.tensor([[math.cos(a) * zoom, -math.sin(a) * zoom, u(-4, 4)], [math.sin(a) * zoom, math.cos(a) * zoom, u(-4, 4)], [u(-2e-4, 2e-4), u(-2e-4, 2e-4), 1.0]])
    homos = torch.stack([base + torch.tensor([[0, 0, 0.9 * d], [0, 0, -0.5 * d], [0, 0, 0.0]]) for d in range(D)])
    scale = (Ll_w / max(W * zoom, 1.0) * u(0.9, 1.1), Ll_h / max(H * zoom, 1.0) * u(0.9, 1.1))      # frame -> roughly the whole lattice
    keep = torch.rand(D, QH, QW, generator=g) < [0.3, 0.7, 1.0][seed % 3]
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=300 + seed)
    with_reg = seed % 3 == 0 and D <= 4
    o_spec = MO.RenderSpec(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post", scale=scale, tile=(th, tw))
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    g_a = synth.hash_uniform((T, H, W), seed=6) - 0.5
    wts = torch.tensor([1e-4, 2e-4, 3e-4, 4e-4])
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, _, layers = MO.render_planes(s_cpu, homos, H, W, o_spec, return_layers=True, quad_keep=keep)
    obj_o = (rgb_o * g_rgb).sum() + (alpha_o * g_a).sum()
    if with_reg:
        sums_o = torch.stack([(layers[:, :, 1:, :, :3] - layers[:, :, :-1, :, :3]).abs().sum(), (layers[:, 1:, :, :, :3] - layers[:, :-1, :, :, :3]).abs().sum(),
                              (layers[:, :, 1:, :, 3] - layers[:, :, :-1, :, 3]).abs().sum(), (layers[:, 1:, :, :, 3] - layers[:, :-1, :, :, 3]).abs().sum()])
        obj_o = obj_o + (sums_o * wts).sum()
    (gs_o,) = torch.autograd.grad(obj_o, s_cpu)
    # every second case renders from a window of the planes (8-aligned like the optimiser's windows, clamped to the plane) that holds every touched texel
    y0 = x0 = 0
    y1, x1 = Hs, Ws
    if seed % 2:
        nz = gs_o.abs().sum((0, 1, 4)) > 0
        if bool(nz.any()):
            ys, xs = nz.any(1).nonzero().flatten(), nz.any(0).nonzero().flatten()
            y0, y1 = max(int(ys.min()) - 2, 0) // 8 * 8, min(-(-(int(ys.max()) + 3) // 8) * 8, Hs)
            x0, x1 = max(int(xs.min()) - 2, 0) // 8 * 8, min(-(-(int(xs.max()) + 3) // 8) * 8, Ws)
    p_spec = dataclasses.replace(RenderSpec.mpv(scale=scale, offset=(-float(x0), -float(y0))), tile=(th, tw))
    win = stack[:, :, y0:y1, x0:x1].contiguous().to(dev).requires_grad_(True)
    cw = (y0, x0, Hs, Ws) if seed % 2 else None
    if with_reg:
        rgb, alpha, sums, _ = render_planes_with_regularisers(win, homos.to(dev), H, W, p_spec, quad_keep=keep.to(dev), cull_window=cw)
        obj = (rgb * g_rgb.to(dev)).sum() + (alpha * g_a.to(dev)).sum() + (sums * wts.to(dev)).sum()
        assert float(((sums.cpu() - sums_o).abs() / sums_o.abs().clamp_min(1.0)).max()) <= 1e-4
    else:
        rgb, alpha = render_planes(win, homos.to(dev), H, W, p_spec, quad_keep=keep.to(dev), cull_window=cw)
        obj = (rgb * g_rgb.to(dev)).sum() + (alpha * g_a.to(dev)).sum()
    (gs,) = torch.autograd.grad(obj, win)
    assert maxabs(rgb, rgb_o) <= TOL and maxabs(alpha, alpha_o) <= TOL
    assert maxabs(gs, gs_o[:, :, y0:y1, x0:x1]) <= TOL * max(1.0, float(gs_o.abs().max()))
