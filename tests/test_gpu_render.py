"""Parity of the HIP render path (through the C ABI) against the golden vectors and the CPU oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import mpi_oracle as MO
from videoloop3d_amd import synth

pytestmark = pytest.mark.gpu
T_ = lambda a: torch.from_numpy(np.asarray(a))
TOL = 1e-4   # north star: <= 1e-4 max-abs in fp32


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import __graft_entry__ as g
    g.build()
    return torch.device("cuda:0")


def maxabs(a, b):
    return float((a.detach().double().cpu() - torch.as_tensor(b).double().cpu()).abs().max())


def bench_homos(D, H, W, near=1.0, far=100.0, scale=1.0):
    from videoloop3d_amd.utils_mpi import compute_homography, make_depths
    ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
    tar_e = tar_e.clone()
    tar_e[:3, 3] *= scale
    depths = make_depths(D, near, far).flip(0)
    return compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None],
                              torch.tensor([0., 0., 1.]).expand(1, D, 3), depths[None])[0]


# ---- unfused drop-ins vs the reference goldens ---------------------------------------------------------
def test_g2_warp_homography(dev, golden):
    from videoloop3d_amd.utils_mpi import warp_homography
    g = golden("g2_warp.npz")
    img = T_(g["images"]).to(dev).requires_grad_(True)
    out = warp_homography(int(g["h"]), int(g["w"]), T_(g["homos"]).to(dev), img)
    assert out.shape == tuple(g["out"].shape)
    assert maxabs(out, g["out"]) <= 1e-5
    (gi,) = torch.autograd.grad(out, img, T_(g["grad_out"]).to(dev))
    assert maxabs(gi, g["grad_images"]) <= 1e-4


def test_g3_overcompose(dev, golden):
    from videoloop3d_amd.utils_mpi import overcompose, overcomposeNto0
    g = golden("g3_overcompose.npz")
    a = T_(g["alpha"]).to(dev).requires_grad_(True)
    c = T_(g["content"]).to(dev).requires_grad_(True)
    rgb, bw = overcompose(a, c)
    assert maxabs(rgb, g["rgb"]) <= 1e-5 and maxabs(bw, g["blendweight"]) <= 1e-6
    ga, gc = torch.autograd.grad([rgb, bw], [a, c], [T_(g["g_rgb"]).to(dev), T_(g["g_bw"]).to(dev)])
    assert maxabs(gc, g["grad_content"]) <= 1e-5
    assert maxabs(ga, g["grad_alpha"]) <= 1e-4 * max(1.0, float(np.abs(g["grad_alpha"]).max()))
    m = T_(g["mpi"]).to(dev).requires_grad_(True)
    rgbN, bwN = overcomposeNto0(m, ret_mask=True)
    assert maxabs(rgbN, g["rgbN"]) <= 1e-5 and maxabs(bwN, g["bwN"]) <= 1e-6
    (gm,) = torch.autograd.grad(rgbN, m, T_(g["g_rgbN"]).to(dev))
    assert maxabs(gm, g["grad_mpi"]) <= 1e-4 * max(1.0, float(np.abs(g["grad_mpi"]).max()))
    # caller-supplied blendweight short-circuit (utils_mpi.py:121-125)
    rgb2 = overcomposeNto0(m.detach(), blendweight=T_(g["bwN"]).to(dev))
    assert maxabs(rgb2, g["rgbN"]) <= 1e-5


def test_overcompose_opaque_plane_exact_gradient(dev):
    """alpha == 1 somewhere: cumprod's backward in the reference is exact there; ours must be too."""
    from videoloop3d_amd.utils_mpi import overcompose
    torch.manual_seed(0)
    a = torch.rand(1, 4, 5, 6) * 0.9
    a[0, 1, 2, 3] = 1.0
    a[0, 0, 0, 0] = 1.0
    c = torch.randn(1, 4, 5, 6, 3)
    ao, co = a.clone().requires_grad_(True), c.clone().requires_grad_(True)
    r, b = MO.overcompose(ao, co)
    gr, gb = torch.randn_like(r), torch.randn_like(b)
    gao, gco = torch.autograd.grad([r, b], [ao, co], [gr, gb])
    ag, cg = a.to(dev).requires_grad_(True), c.to(dev).requires_grad_(True)
    r2, b2 = overcompose(ag, cg)
    ga, gc = torch.autograd.grad([r2, b2], [ag, cg], [gr.to(dev), gb.to(dev)])
    assert maxabs(ga, gao) <= 1e-4 and maxabs(gc, gco) <= 1e-5


# ---- fused render -----------------------------------------------------------------------------------
def test_g4_cfg1_fused_vs_reference_golden(dev, golden):
    """cfg1: 256x256, D=8, 1 view -- fused kernel vs sigmoid->warp_homography->overcomposeNto0 of the reference."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    g = golden("g4_cfg1_render.npz")
    stack = synth.make_plane_stack(8, 1, 256, 256, seed=2, device=dev).requires_grad_(True)
    rgb, alpha = render_planes(stack, T_(g["homos"]).to(dev), 256, 256, RenderSpec())
    assert maxabs(rgb[0].permute(2, 0, 1), g["rgb"]) <= TOL
    gout = (synth.hash_uniform((1, 3, 256, 256), seed=7) - 0.5)[0].permute(1, 2, 0)[None].to(dev)
    (gs,) = torch.autograd.grad(rgb, stack, gout)
    gs = gs[:, 0]
    assert maxabs(gs[:, 96:160, 96:160], g["grad_stack_crop"]) <= TOL
    s = g["grad_stack_sum"]
    assert abs(float(gs.double().sum()) - s[0]) <= 1e-3 * max(1.0, abs(s[0]))
    assert abs(float(gs.double().abs().sum()) - s[1]) <= 1e-4 * s[1]


MPV_KW = dict(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post")
SPECS = {
    "utils_mpi": (dict(), dict()),
    "mpv": (dict(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post"),) * 2,
    "mpv_atlas_pitch": (dict(pixel_center=0.5, coord_mode="affine", scale=(1.00124, 0.998), offset=(0.3, -0.2),
                             border="hardcut", act_order="post"),) * 2,
    "zeros_post": (dict(border="zeros", act_order="post"),) * 2,
    "hardcut_pre": (dict(border="hardcut", act_order="pre"),) * 2,
    "none_act": (dict(rgb_act="none", alpha_act="sigmoid"),) * 2,
    "clamp": (dict(pixel_center=0.5, coord_mode="affine", rgb_act="clamp", alpha_act="clamp", act_order="post", border="hardcut"),) * 2,
    "relu_rgb": (dict(pixel_center=0.5, coord_mode="affine", rgb_act="relu", alpha_act="sigmoid", act_order="post", border="hardcut"),) * 2,
    "abs_rgb": (dict(pixel_center=0.5, coord_mode="affine", rgb_act="abs", alpha_act="sigmoid", act_order="post", border="hardcut"),) * 2,
    "linear": (dict(rgb_act="none", alpha_act="none"),) * 2,
}


@pytest.mark.parametrize("name", list(SPECS))
@pytest.mark.parametrize("shape", [(8, 2, 48, 64, 40, 56), (5, 1, 33, 47, 61, 70), (3, 3, 20, 24, 9, 130)])
def test_fused_vs_oracle(dev, name, shape):
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, Hs, Ws, H, W = shape
    kw_p, kw_o = SPECS[name]
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=11)
    if name in ("clamp", "linear"):
        stack = stack * 0.4 + 0.5
        if name == "linear":            # pre-activated stacks: alphas must already be in (0, 1)
            stack[..., 3] = stack[..., 3].clamp(0.02, 0.9)
    homos = bench_homos(D, H, W, scale=4.0)     # strong parallax: planes partly leave the frame
    # add an in-plane rotation/zoom so taps are not axis aligned
    th = math.radians(3.0)
    Rz = torch.tensor([[math.cos(th) * 1.07, -math.sin(th), 2.0], [math.sin(th), math.cos(th) * 0.93, -1.5], [1e-4, -2e-4, 1.0]])
    homos = homos @ Rz
    # map the (H,W) frame onto the (Hs,Ws) plane extent
    S = torch.tensor([[Ws / W, 0, 0], [0, Hs / H, 0], [0, 0, 1.0]])
    homos = S @ homos
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    g_a = synth.hash_uniform((T, H, W), seed=6) - 0.5
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, _ = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(**kw_o))
    (gs_o,) = torch.autograd.grad([rgb_o, alpha_o], s_cpu, [g_rgb, g_a])
    s_gpu = stack.to(dev).requires_grad_(True)
    rgb, alpha = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec(**kw_p))
    (gs,) = torch.autograd.grad([rgb, alpha], s_gpu, [g_rgb.to(dev), g_a.to(dev)])
    assert maxabs(rgb, rgb_o) <= TOL
    assert maxabs(alpha, alpha_o) <= TOL
    tol_g = TOL * max(1.0, float(gs_o.abs().max()))
    if name in ("clamp", "relu_rgb", "abs_rgb"):
        # activations with a kink: a sample within rounding of the kink has derivative 0 on one side and 1 on the other, so a
        # handful of texels may differ by a whole tap weight; everything else must agree to the tolerance
        bad = (gs.cpu() - gs_o).abs() > tol_g
        assert float(bad.float().mean()) <= 2e-4
    else:
        assert maxabs(gs, gs_o) <= tol_g
    assert float(gs_o.abs().sum()) > 0


def test_uncovered_frame_and_degenerate_homography(dev):
    """all planes outside the frame / Z<=0 -> rgb = alpha = 0 and zero gradient (no NaNs)."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, Hs, Ws, H, W = 3, 1, 16, 16, 8, 8
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=1, device=dev).requires_grad_(True)
    far_away = torch.eye(3).repeat(D, 1, 1)
    far_away[:, 0, 2] = 1e6
    far_away[1, 2, 2] = -1.0      # behind the camera
    far_away[2, 2, 2] = 0.0       # division by zero
    far_away[2, 2, 0] = 0.0
    for spec in (RenderSpec(), RenderSpec.mpv()):
        rgb, alpha = render_planes(stack, far_away.to(dev), H, W, spec)
        (gs,) = torch.autograd.grad(rgb.sum() + alpha.sum(), stack)
        assert torch.isfinite(rgb).all() and torch.isfinite(gs).all()
    # plane 0 only is far away: still finite
    assert float(alpha.abs().max()) <= 1.0


def test_row_band_windows_tile_the_frame_exactly(dev):
    """rendering row bands with window offsets and concatenating == the full render, bit for bit;
    band gradients sum to the full gradient."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, Hs, Ws, H, W = 6, 2, 70, 90, 64, 80
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=3, device=dev).requires_grad_(True)
    homos = (torch.tensor([[Ws / W, 0, 0], [0, Hs / H, 0], [0, 0, 1.0]]) @ bench_homos(D, H, W, scale=2.0)).to(dev)
    spec = RenderSpec.mpv()
    g = (synth.hash_uniform((T, H, W, 3), seed=9) - 0.5).to(dev)
    full, full_a = render_planes(stack, homos, H, W, spec)
    (g_full,) = torch.autograd.grad(full, stack, g)
    bands, g_sum = [], torch.zeros_like(g_full)
    for r in range(4):
        r0 = r * H // 4
        b, _ = render_planes(stack, homos, H // 4, W, spec, window=(r0, 0))
        bands.append(b)
        (gb,) = torch.autograd.grad(b, stack, g[:, r0:r0 + H // 4])
        g_sum += gb
    assert torch.equal(torch.cat(bands, 1), full)
    assert maxabs(g_sum, g_full) <= 1e-5


def test_constant_stack_closed_form_720p(dev):
    """size-independent property at cfg2 scale (720p, D=32): a constant stack under any in-range warp renders
    c*(1-(1-a)^D); the gradient of sum(rgb) sums to the analytic value."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, Hs, Ws, H, W = 32, 1, 792, 1408, 720, 1280      # mpi_h/w_scale = 1.1 (configs/mpv_base.txt:10-11)
    v = torch.tensor([0.3, -0.7, 1.1, -1.5])
    stack = v.to(dev).expand(D, T, Hs, Ws, 4).contiguous().requires_grad_(True)
    homos = bench_homos(D, H, W)
    homos = (torch.tensor([[1.0, 0, 64.0], [0, 1.0, 36.0], [0, 0, 1.0]]) @ homos).to(dev)  # centre the frame in the plane
    rgb, alpha = render_planes(stack, homos, H, W, RenderSpec.mpv())
    c, a = torch.sigmoid(v[:3]), torch.sigmoid(v[3])
    A = 1 - (1 - a) ** D
    assert maxabs(alpha, A.expand_as(alpha)) <= 1e-5
    assert maxabs(rgb, (c * A).expand_as(rgb)) <= 1e-5
    (gs,) = torch.autograd.grad(rgb.sum(), stack)
    # d(sum rgb)/d(v_rgb) summed over all texels & planes = Npix * A * c(1-c) per channel
    npix = H * W
    got = gs.double().sum(dim=(0, 1, 2, 3)).cpu()
    want_rgb = npix * A.double() * (c * (1 - c)).double()
    assert (got[:3] - want_rgb).abs().max() <= 1e-3 * want_rgb.abs().max()
    # alpha channel: d/dv_a of sum_c c*(1-(1-a)^D) over planes = Npix * sum(c) * D (1-a)^(D-1) * a(1-a) / ... summed over k
    want_a = npix * c.double().sum() * D * (1 - a.double()) ** (D - 1) * (a * (1 - a)).double()
    assert abs(got[3] - want_a) <= 1e-3 * abs(want_a)


def test_720p_window_vs_oracle(dev):
    """cfg2-sized frame (720p, D=32): a 24x40 window of the full render and of its gradient vs the CPU oracle."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, Hs, Ws, H, W = 32, 1, 720, 1280, 720, 1280
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=2, device=dev).requires_grad_(True)
    homos = bench_homos(D, H, W)
    r0, c0, h, w = 333, 611, 24, 40
    for spec, ospec in ((RenderSpec(), MO.RenderSpec()),
                        (RenderSpec.mpv(), MO.RenderSpec(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post"))):
        rgb, alpha = render_planes(stack, homos.to(dev), H, W, spec)
        gfull = torch.zeros_like(rgb)
        gwin = (synth.hash_uniform((T, h, w, 3), seed=5) - 0.5)
        gfull[:, r0:r0 + h, c0:c0 + w] = gwin.to(dev)
        (gs,) = torch.autograd.grad(rgb, stack, gfull)
        # oracle on the window: fold the integer window offset into the homography
        shift = torch.tensor([[1.0, 0, c0], [0, 1.0, r0], [0, 0, 1.0]])
        lo_r, hi_r, lo_c, hi_c = r0 - 40, r0 + h + 40, c0 - 60, c0 + w + 60
        s_cpu = stack.detach().cpu().requires_grad_(True)
        rgb_o, alpha_o, _ = MO.render_planes(s_cpu, homos @ shift, h, w, ospec)
        (gs_o,) = torch.autograd.grad(rgb_o, s_cpu, gwin)
        # (pixel_center is added after the shift in both, so the fold is exact for integer offsets)
        assert maxabs(rgb[:, r0:r0 + h, c0:c0 + w], rgb_o) <= TOL
        assert maxabs(alpha[:, r0:r0 + h, c0:c0 + w], alpha_o) <= TOL
        assert maxabs(gs[:, :, lo_r:hi_r, lo_c:hi_c], gs_o[:, :, lo_r:hi_r, lo_c:hi_c]) <= TOL
        assert float(gs_o[:, :, lo_r:hi_r, lo_c:hi_c].abs().sum()) == pytest.approx(float(gs_o.abs().sum()))


# ---- backward kernel variants ---------------------------------------------------------------------------
def _tile_ran():
    from videoloop3d_amd import render
    return int(render.LAST_BWD_SCRATCH.view(torch.int32)[0].item())


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("spec_name", ["mpv", "utils_mpi", "hardcut_pre"])
@pytest.mark.parametrize("shape", [(6, 2, 150, 200, 139, 187), (4, 1, 70, 300, 64, 280), (3, 1, 40, 40, 37, 35), (5, 3, 96, 130, 96, 130), (3, 1, 45, 53, 41, 50)])
def test_bwd_variants_agree_with_oracle(dev, variant, spec_name, shape):
    """variant 0 = default dispatch (frame-pair kernels where they apply: the last shape is a 1.0x stack with T = 3, i.e. the
    pair kernel DIRECTLY against the oracle, incl. the odd tail frame; the T = 1 shapes take the flat 64 x 8 regions under "mpv"),
    1 = global atomics, 2 = the tile kernel in flat 64 x 8 regions at any T, 3 = LDS-staged owner-computes tile
    kernel in 64 x 16 regions (and the one-texel owner-table pass; the last shape: one frame whose rows are no multiple of four texels -- the
    four-texel pass's scalar stores); near-unit-scale geometry with rotation + perspective so the owner-computes plan is feasible, odd sizes so tiles
    are ragged."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, Hs, Ws, H, W = shape
    kw_p, kw_o = SPECS[spec_name]
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=13)
    th = math.radians(2.0)
    Rz = torch.tensor([[math.cos(th) * 1.05, -math.sin(th), 3.0], [math.sin(th), math.cos(th) * 0.96, 2.5], [2e-5, -3e-5, 1.0]])
    homos = bench_homos(D, H, W, scale=1.5) @ Rz
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    g_a = synth.hash_uniform((T, H, W), seed=6) - 0.5
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, _ = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(**kw_o))
    (gs_o,) = torch.autograd.grad([rgb_o, alpha_o], s_cpu, [g_rgb, g_a])
    s_gpu = stack.to(dev).requires_grad_(True)
    rgb, alpha = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec(variant=variant, **kw_p))
    (gs,) = torch.autograd.grad([rgb, alpha], s_gpu, [g_rgb.to(dev), g_a.to(dev)])
    assert _tile_ran() == (0 if variant == 1 else 1)
    assert maxabs(gs, gs_o) <= TOL * max(1.0, float(gs_o.abs().max()))
    assert torch.isfinite(gs).all()


def test_bwd_infeasible_geometry_falls_back_on_device(dev):
    """2x magnification violates the 1-pixel-halo precondition: the on-device plan must route to the atomics kernel
    (no host sync, same result)."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, Hs, Ws, H, W = 3, 1, 30, 40, 60, 80
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=1)
    homos = torch.tensor([[0.5, 0, 0], [0, 0.5, 0], [0, 0, 1.0]]).repeat(D, 1, 1)
    g = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, _, _ = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec())
    (gs_o,) = torch.autograd.grad(rgb_o, s_cpu, g)
    s_gpu = stack.to(dev).requires_grad_(True)
    rgb, _ = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec(variant=3))
    (gs,) = torch.autograd.grad(rgb, s_gpu, g.to(dev))
    assert _tile_ran() == 0
    assert maxabs(gs, gs_o) <= TOL


def test_bwd_tile_720p_matches_atomics(dev):
    """cfg2 scale (720p, D=32, benchmark cameras): the owner-computes kernel equals the atomics kernel to fp32
    summation-order noise, every texel of grad_stack is written (no holes: the buffer starts as NaN)."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, Hs, Ws, H, W = 32, 2, 720, 1280, 720, 1280
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=2, device=dev).requires_grad_(True)
    homos = bench_homos(D, H, W).to(dev)
    g = (synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5)
    out = {}
    from videoloop3d_amd import _lib as L
    from videoloop3d_amd.render import _desc
    rgb, alpha = render_planes(stack.detach(), homos, H, W, RenderSpec.mpv())
    for variant in (1, 0, 3, 2):
        # raw ABI with the gradient buffer pre-filled with NaN: a texel no kernel writes stays NaN (deterministic, unlike hoping
        # that the caching allocator hands out a poisoned block)
        d = _desc(stack, H, W, RenderSpec.mpv(variant=variant), 0, 0)
        gs = torch.full_like(stack, float("nan")).detach()
        nscratch = int(L.lib().vl3d_render_bwd_scratch_bytes(d))
        scratch = torch.zeros((nscratch + 3) // 4, dtype=torch.float32, device=dev)
        L.check(L.lib().vl3d_render_bwd(d, L.ptr(stack), L.ptr(homos), L.ptr(rgb), L.ptr(alpha), L.ptr(g), None, None, None, None, L.ptr(gs),
                                        L.ptr(scratch), nscratch, L.stream_ptr(dev)), "vl3d_render_bwd")
        assert int(scratch.view(torch.int32)[0].item()) == (0 if variant == 1 else 1)
        assert torch.isfinite(gs).all()
        out[variant] = gs
    scale = float(out[1].abs().max())
    # frame pairs (default dispatch) and the one-frame tile kernel: the same bits
    assert torch.equal(out[0], out[3])
    # ... and the flat 64 x 8 regions (the default of a single frame)
    assert torch.equal(out[2], out[3])
    # vs the atomics kernel: same coordinates bit for bit (explicit FMAs in make_taps2), so only the summation order differs
    assert maxabs(out[3], out[1]) <= 2e-6 * max(1.0, scale)


def test_render_band_from_local_rows_matches_full(dev):
    """the multi-GPU building block on one GPU: every band rendered from its band-local stack rows (halo only) equals
    the corresponding rows of the full render bit-for-bit; band gradients land in the right local rows."""
    from videoloop3d_amd.dist import plan_bands, render_band
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, Hs, Ws, H, W = 6, 2, 150, 96, 144, 90
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=3, device=dev)
    homos = (torch.tensor([[1.0, 0, 3.0], [0, 1.0, 3.0], [0, 0, 1.0]]) @ bench_homos(D, H, W, scale=2.0))
    spec = RenderSpec.mpv()
    full_in = stack.clone().requires_grad_(True)
    full, _ = render_planes(full_in, homos.to(dev), H, W, spec)
    g = (synth.hash_uniform((T, H, W, 3), seed=9) - 0.5).to(dev)
    (g_full,) = torch.autograd.grad(full, full_in, g)
    g_acc = torch.zeros_like(g_full)
    for world in (4,):
        for b in plan_bands(homos, H, W, Hs, world, spec):
            local = stack[:, :, b.src0:b.src1].contiguous().requires_grad_(True)
            rgb, _ = render_band(local, homos.to(dev), b, W, Hs, spec)
            assert torch.equal(rgb, full[:, b.row0:b.row0 + b.rows])
            (gl,) = torch.autograd.grad(rgb, local, g[:, b.row0:b.row0 + b.rows])
            g_acc[:, :, b.src0:b.src1] += gl
    assert maxabs(g_acc, g_full) <= 1e-5


@pytest.mark.parametrize("feasible", [True, False])
@pytest.mark.parametrize("spec_name", ["mpv", "utils_mpi"])
def test_fused_smoothness_regularisers(dev, feasible, spec_name):
    """rgb_smooth / a_smooth (MPV.py:517-531) from the fused kernels == finite differences of the materialised layers
    (oracle), value and gradient; `feasible=False` uses 2x magnification so the backward falls back to the atomics kernel."""
    from videoloop3d_amd.render import RenderSpec, render_planes_with_smoothness
    D, T, Hs, Ws, H, W = 5, 2, 90, 130, 83, 121
    kw_p, kw_o = SPECS[spec_name]
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=17) * 0.5
    if feasible:
        th = math.radians(1.5)
        Rz = torch.tensor([[math.cos(th) * 1.03, -math.sin(th), 3.0], [math.sin(th), math.cos(th) * 0.98, 2.0], [1e-5, -2e-5, 1.0]])
        homos = bench_homos(D, H, W, scale=1.5) @ Rz
    else:
        homos = torch.tensor([[0.5, 0, 10.0], [0, 0.5, 8.0], [0, 0, 1.0]]) @ bench_homos(D, H, W, scale=1.0)
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    coef = torch.tensor([0.7, -0.4, 1.3, 0.9])
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, _, L = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(**kw_o), return_layers=True)   # L: T,H,W,D,4
    sums_o = torch.stack([(L[:, :, :-1, :, :3] - L[:, :, 1:, :, :3]).abs().sum(), (L[:, :-1, :, :, :3] - L[:, 1:, :, :, :3]).abs().sum(),
                          (L[:, :, :-1, :, 3] - L[:, :, 1:, :, 3]).abs().sum(), (L[:, :-1, :, :, 3] - L[:, 1:, :, :, 3]).abs().sum()])
    loss_o = (rgb_o * g_rgb).sum() + (sums_o * coef).sum() * 1e-3
    (gs_o,) = torch.autograd.grad(loss_o, s_cpu)
    s_gpu = stack.to(dev).requires_grad_(True)
    rgb, alpha, sums = render_planes_with_smoothness(s_gpu, homos.to(dev), H, W, RenderSpec(**kw_p))
    loss = (rgb * g_rgb.to(dev)).sum() + (sums * coef.to(dev)).sum() * 1e-3
    (gs,) = torch.autograd.grad(loss, s_gpu)
    assert _tile_ran() == (1 if feasible else 0)
    assert maxabs(sums, sums_o) <= 2e-5 * float(sums_o.abs().max())
    assert maxabs(rgb, rgb_o) <= TOL
    # sign() of near-zero layer differences may flip with 1-ulp sampling differences: compare with a robust criterion
    diff = (gs.cpu() - gs_o).abs()
    scale = float(gs_o.abs().max())
    assert float((diff > 1e-4 * max(1.0, scale)).float().mean()) <= 1e-4
    assert float(diff.max()) <= 5e-3 * max(1.0, scale)


@pytest.mark.parametrize("spec_name", ["mpv", "utils_mpi"])
def test_fp16_plane_stack(dev, spec_name):
    """cfg5 of BASELINE.json keeps the plane stack in fp16 (8-byte texels); arithmetic stays fp32, so the result equals the
    fp32 path on the fp16-rounded stack (the reference's own --fp16 is 'do NOT use', config_parser.py:32-33: the oracle on
    the rounded values is the parity definition)."""
    from videoloop3d_amd.render import RenderSpec, render_planes, render_planes_with_smoothness
    D, T, Hs, Ws, H, W = 6, 2, 100, 140, 93, 131
    kw_p, kw_o = SPECS[spec_name]
    stack16 = synth.make_plane_stack(D, T, Hs, Ws, seed=21).half()
    homos = bench_homos(D, H, W, scale=1.5)
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    s_cpu = stack16.float().requires_grad_(True)
    rgb_o, alpha_o, _ = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(**kw_o))
    (gs_o,) = torch.autograd.grad(rgb_o, s_cpu, g_rgb)
    s_gpu = stack16.to(dev).requires_grad_(True)
    rgb, alpha = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec(**kw_p))
    (gs,) = torch.autograd.grad(rgb, s_gpu, g_rgb.to(dev))
    assert _tile_ran() == 1 and gs.dtype == torch.float16
    assert maxabs(rgb, rgb_o) <= TOL and maxabs(alpha, alpha_o) <= TOL
    assert maxabs(gs.float(), gs_o) <= 1e-3 * max(1e-3, float(gs_o.abs().max())) + 1e-6      # fp16 rounding of the returned gradient
    # the atomics fallback accumulates straight into the fp16 gradient (packed-half atomics): a few roundings per texel
    rgb_a, _ = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec(variant=1, **kw_p))
    (gs_a,) = torch.autograd.grad(rgb_a, s_gpu, g_rgb.to(dev))
    assert _tile_ran() == 0 and gs_a.dtype == torch.float16
    assert maxabs(gs_a.float(), gs_o) <= 4e-3 * max(1e-3, float(gs_o.abs().max())) + 1e-5
    # bit-identical to the fp32 kernels on the same (rounded) values, including the fused regulariser sums
    s32 = stack16.float().to(dev).requires_grad_(True)
    rgb32, _, sums32 = render_planes_with_smoothness(s32, homos.to(dev), H, W, RenderSpec(**kw_p))
    rgb16, _, sums16 = render_planes_with_smoothness(s_gpu, homos.to(dev), H, W, RenderSpec(**kw_p))
    assert torch.equal(rgb16, rgb32) and torch.equal(sums16, sums32)


@pytest.mark.parametrize("shape", [(1, 1, 4, 6, 3, 5), (2, 1, 9, 70, 1, 66), (1, 3, 40, 5, 33, 1), (3, 2, 17, 129, 16, 125)])
def test_tiny_and_ragged_frames(dev, shape):
    """degenerate sizes: single plane / frame, frames narrower than a wave, one-pixel rows and columns, tile-size +- 1."""
    from videoloop3d_amd.render import RenderSpec, render_planes_with_smoothness
    D, T, Hs, Ws, H, W = shape
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=4) * 0.5
    homos = torch.tensor([[1.0, 0.01, 0.3], [-0.01, 1.0, 0.2], [0, 0, 1.0]]).repeat(D, 1, 1)
    homos[:, 0, 2] += torch.arange(D) * 0.7
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    for kw in (dict(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post"), dict()):
        s_cpu = stack.clone().requires_grad_(True)
        rgb_o, alpha_o, _, L = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(**kw), return_layers=True)
        sx = (L[:, :, :-1, :, :3] - L[:, :, 1:, :, :3]).abs().sum() if W > 1 else L.sum() * 0
        sy = (L[:, :-1, :, :, :3] - L[:, 1:, :, :, :3]).abs().sum() if H > 1 else L.sum() * 0
        loss_o = (rgb_o * g_rgb).sum() + 1e-2 * (sx - 0.5 * sy)
        (gs_o,) = torch.autograd.grad(loss_o, s_cpu)
        s_gpu = stack.to(dev).requires_grad_(True)
        rgb, alpha, sums = render_planes_with_smoothness(s_gpu, homos.to(dev), H, W, RenderSpec(**kw))
        loss = (rgb * g_rgb.to(dev)).sum() + 1e-2 * (sums[0] - 0.5 * sums[1])
        (gs,) = torch.autograd.grad(loss, s_gpu)
        assert maxabs(rgb, rgb_o) <= TOL and maxabs(alpha, alpha_o) <= TOL
        assert maxabs(sums[0], sx) <= 1e-4 * max(1.0, float(sx)) and maxabs(sums[1], sy) <= 1e-4 * max(1.0, float(sy))
        d = (gs.cpu() - gs_o).abs()
        assert float(d.max()) <= 5e-3 and float((d > 1e-4).float().mean()) <= 2e-3


@pytest.mark.parametrize("logit_hi", [3.0, 6.0])
def test_nearly_opaque_planes_gradient_envelope(dev, logit_hi):
    """The backward uses sum_{j>k} w_j q_j = (G.C + gA.A) - sum_{j<=k} w_j q_j, divided by (1 - a_k): cancellation error is
    amplified by 1/(1-a).  With alpha logits up to +6 (a = 0.9975, amplification 400x) the gradient must still agree with the
    oracle's autograd (which divides by (1-a) as well, through cumprod's backward) to 1e-4 of the gradient scale."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, Hs, Ws, H, W = 12, 1, 70, 90, 64, 80
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=23)
    stack[..., 3] = synth.hash_uniform((D, T, Hs, Ws), seed=24) * (logit_hi + 4.0) - 4.0
    homos = (torch.tensor([[Ws / W, 0, 0], [0, Hs / H, 0], [0, 0, 1.0]]) @ bench_homos(D, H, W, scale=1.0))
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    g_a = synth.hash_uniform((T, H, W), seed=6) - 0.5
    spec_o = MO.RenderSpec(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post")
    s_cpu = stack.clone().double().requires_grad_(True)      # fp64 oracle = ground truth for the envelope
    rgb_o, alpha_o, _ = MO.render_planes(s_cpu, homos.double(), H, W, spec_o)
    (gs_o,) = torch.autograd.grad([rgb_o, alpha_o], s_cpu, [g_rgb.double(), g_a.double()])
    s_gpu = stack.to(dev).requires_grad_(True)
    for variant in (0, 1):
        rgb, alpha = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec.mpv(variant=variant))
        (gs,) = torch.autograd.grad([rgb, alpha], s_gpu, [g_rgb.to(dev), g_a.to(dev)])
        assert maxabs(rgb, rgb_o) <= TOL
        assert maxabs(gs, gs_o) <= TOL * max(1.0, float(gs_o.abs().max()))


@pytest.mark.parametrize("spec_name", ["mpv", "utils_mpi"])
def test_bwd_tile_path_is_deterministic(dev, spec_name):
    """The owner-computes backward writes every texel once in a fixed order: two runs are bitwise identical."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, Hs, Ws, H, W = (6, 2, 150, 200, 139, 187)
    kw_p, _ = SPECS[spec_name]
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=13, device=dev).requires_grad_(True)
    th = math.radians(2.0)
    Rz = torch.tensor([[math.cos(th) * 1.05, -math.sin(th), 3.0], [math.sin(th), math.cos(th) * 0.96, 2.5], [2e-5, -3e-5, 1.0]])
    homos = (bench_homos(D, H, W, scale=1.5) @ Rz).to(dev)
    g_rgb = (synth.hash_uniform((T, H, W, 3), seed=5) - 0.5).to(dev)
    g_a = (synth.hash_uniform((T, H, W), seed=6) - 0.5).to(dev)
    outs = []
    for _ in range(2):
        rgb, alpha = render_planes(stack, homos, H, W, RenderSpec(**kw_p))
        (gs,) = torch.autograd.grad([rgb, alpha], stack, [g_rgb, g_a])
        assert _tile_ran() == 1
        outs.append(gs)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("spec_name", ["mpv", "utils_mpi", "hardcut_pre"])
@pytest.mark.parametrize("keep_frac", [0.0, 0.3, 1.0])
def test_tile_culling_matches_oracle(dev, spec_name, keep_frac):
    """vl3d_render_fwd/bwd_culled (include/vl3d.h; MPI.py:288-442): samples in culled quads are uncovered -- outputs and the stack
    gradient match the oracle's quad coverage, culled texels get exactly zero gradient, and a map that keeps everything is bitwise
    the plain render."""
    from videoloop3d_amd import tiles
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, Hs, Ws, H, W = 7, 2, 150, 200, 139, 187
    QH, QW = 6, 9
    kw_p, kw_o = SPECS[spec_name]
    torch.manual_seed(3)
    keep = torch.rand(D, QH, QW) < keep_frac
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=13)
    th = math.radians(2.0)
    Rz = torch.tensor([[math.cos(th) * 1.05, -math.sin(th), 3.0], [math.sin(th), math.cos(th) * 0.96, 2.5], [2e-5, -3e-5, 1.0]])
    homos = bench_homos(D, H, W, scale=1.5) @ Rz
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    g_a = synth.hash_uniform((T, H, W), seed=6) - 0.5
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, _ = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(**kw_o), quad_keep=keep)
    (gs_o,) = torch.autograd.grad([rgb_o, alpha_o], s_cpu, [g_rgb, g_a])
    s_gpu = stack.to(dev).requires_grad_(True)
    rgb, alpha = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec(**kw_p), quad_keep=keep.to(dev))
    (gs,) = torch.autograd.grad([rgb, alpha], s_gpu, [g_rgb.to(dev), g_a.to(dev)])
    assert _tile_ran() == 1
    assert maxabs(rgb, rgb_o) <= TOL and maxabs(alpha, alpha_o) <= TOL
    assert maxabs(gs, gs_o) <= TOL * max(1.0, float(gs_o.abs().max()))
    # the culled forward composites two frames per thread (render_fwd2x_k<CULL>); forward variant 6 = one frame per thread: the same bits
    rgb1, alpha1 = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec(variant=0x600, **kw_p), quad_keep=keep.to(dev))
    assert torch.equal(rgb, rgb1) and torch.equal(alpha, alpha1)
    dead = ~tiles.quad_to_texel_mask(keep.to(dev), Hs, Ws)
    assert float(gs[dead[:, None].expand(D, T, Hs, Ws)].abs().max() if dead.any() else 0.0) == 0.0
    if keep_frac == 1.0:
        rgb_p, alpha_p = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec(**kw_p))
        (gs_p,) = torch.autograd.grad([rgb_p, alpha_p], s_gpu, [g_rgb.to(dev), g_a.to(dev)])
        assert torch.equal(rgb, rgb_p) and torch.equal(alpha, alpha_p) and torch.equal(gs, gs_p)


@pytest.mark.parametrize("variant", [0, 1])
def test_tile_culling_with_regularisers_matches_oracle(dev, variant):
    """culling + the fused layer regularisers (tile and atomics backward): culled samples have layer value 0 (MPV.py:441)."""
    from videoloop3d_amd.render import RenderSpec, render_planes_with_regularisers
    D, T, Hs, Ws, H, W = 5, 2, 90, 120, 83, 111
    kw_p, kw_o = SPECS["mpv"]
    torch.manual_seed(5)
    keep = torch.rand(D, 4, 6) < 0.5
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=17)
    homos = bench_homos(D, H, W, scale=1.2)
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, _, layers = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(**kw_o), return_layers=True, quad_keep=keep)
    sums_o = torch.stack([(layers[:, :, 1:, :, :3] - layers[:, :, :-1, :, :3]).abs().sum(), (layers[:, 1:, :, :, :3] - layers[:, :-1, :, :, :3]).abs().sum(),
                          (layers[:, :, 1:, :, 3] - layers[:, :, :-1, :, 3]).abs().sum(), (layers[:, 1:, :, :, 3] - layers[:, :-1, :, :, 3]).abs().sum()])
    wts = torch.tensor([1e-4, 2e-4, 3e-4, 4e-4])
    (gs_o,) = torch.autograd.grad((rgb_o * g_rgb).sum() + (sums_o * wts).sum(), s_cpu)
    s_gpu = stack.to(dev).requires_grad_(True)
    rgb, alpha, sums, _ = render_planes_with_regularisers(s_gpu, homos.to(dev), H, W, RenderSpec(variant=variant, **kw_p), quad_keep=keep.to(dev))
    (gs,) = torch.autograd.grad((rgb * g_rgb.to(dev)).sum() + (sums * wts.to(dev)).sum(), s_gpu)
    assert maxabs(rgb, rgb_o) <= TOL
    assert float(((sums.cpu() - sums_o).abs() / sums_o.abs().clamp_min(1.0)).max()) <= 1e-4
    assert maxabs(gs, gs_o) <= TOL * max(1.0, float(gs_o.abs().max()))


@pytest.mark.parametrize("empty", ["frames", "rows", "cols"])
def test_empty_render_is_empty_like_the_reference(dev, empty):
    """an empty `ts` / zero-area crop gives empty outputs and a zero stack gradient (grid_sample + cumprod on empty tensors,
    MPV.py:425-454), on every wrapper; the ABI itself refuses non-positive dims (test below)."""
    from videoloop3d_amd.render import RenderSpec, render_planes, render_planes_with_regularisers
    D, Hs, Ws = 3, 12, 20
    T, H, W = (0, 8, 10) if empty == "frames" else ((2, 0, 10) if empty == "rows" else (2, 8, 0))
    stack = (synth.make_plane_stack(D, max(T, 1), Hs, Ws, seed=3)[:, :T]).to(dev).requires_grad_(True)
    homos = torch.eye(3).repeat(D, 1, 1).to(dev)
    rgb, alpha = render_planes(stack, homos, H, W, RenderSpec.mpv())
    assert rgb.shape == (T, H, W, 3) and alpha.shape == (T, H, W) and rgb.numel() == 0
    rgb, alpha, sums, asum = render_planes_with_regularisers(stack, homos, H, W)
    assert asum.shape == (T, H, W, 2) and float(sums.detach().abs().sum()) == 0.0
    (g,) = torch.autograd.grad(rgb.sum() + alpha.sum() + sums.sum() + asum.sum(), stack)
    assert g.shape == stack.shape and float(g.abs().sum()) == 0.0


def test_abi_refuses_bad_dims_with_a_status_not_a_crash(dev):
    """SURVEY §8b 'Errors': the native side returns an int status that the wrapper turns into RuntimeError; never aborts."""
    from videoloop3d_amd import _lib as L
    from videoloop3d_amd.render import RenderSpec, _desc
    stack = synth.make_plane_stack(2, 1, 8, 8, seed=1).to(dev)
    homos = torch.eye(3).repeat(2, 1, 1).to(dev)
    out = torch.empty((1, 4, 4, 3), device=dev)
    al = torch.empty((1, 4, 4), device=dev)
    d = _desc(stack, 4, 4, RenderSpec(), 0, 0)
    for field in ("D", "T", "Hs", "Ws", "H", "W"):
        keep = getattr(d, field)
        setattr(d, field, 0)
        rc = L.lib().vl3d_render_fwd(d, L.ptr(stack), L.ptr(homos), L.ptr(out), L.ptr(al), None, L.stream_ptr(dev))
        assert rc == 1                                                       # VL3D_EINVAL
        with pytest.raises(RuntimeError, match="non-positive render dims"):
            L.check(rc, "vl3d_render_fwd")
        setattr(d, field, keep)
    assert L.lib().vl3d_render_fwd(d, None, L.ptr(homos), L.ptr(out), L.ptr(al), None, L.stream_ptr(dev)) == 1
    d.stack_dtype = 7
    assert L.lib().vl3d_render_fwd(d, L.ptr(stack), L.ptr(homos), L.ptr(out), L.ptr(al), None, L.stream_ptr(dev)) == 1


def test_cfg3_full_size_frame_independence_and_linearity(dev):
    """BASELINE.json's metric configuration at FULL size (D=32, T=50, 720p: a 23.6 GB stack and as much gradient), through
    size-independent properties: frames are independent (utils_mpi.py:159-176 has no cross-frame term), so a stack whose 50
    frames are copies of one frame renders 50 bit-identical images and receives 50 bit-identical gradient frames, each
    bit-equal to a T=1 call (pinned against the oracle at this size by test_720p_window_vs_oracle); and the backward is
    linear in the incoming gradient.  Catches 32-bit offset overflow and frame-stride mistakes no small case can show."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 120 * 2**30:
        pytest.skip("needs 120 GiB of free HBM")
    D, T, Hs, Ws, H, W = 32, 50, 720, 1280, 720, 1280
    one = synth.make_plane_stack(D, 1, Hs, Ws, seed=2, device=dev)
    homos = bench_homos(D, H, W).to(dev)
    g1 = synth.hash_uniform((1, H, W, 3), seed=5, device=dev) - 0.5
    g2 = synth.hash_uniform((1, H, W, 3), seed=6, device=dev) - 0.5
    s1 = one.clone().requires_grad_(True)
    rgb1, _ = render_planes(s1, homos, H, W, RenderSpec.mpv())
    (gs1,) = torch.autograd.grad(rgb1, s1, g1, retain_graph=True)
    (gs2,) = torch.autograd.grad(rgb1, s1, g2, retain_graph=True)
    (gs12,) = torch.autograd.grad(rgb1, s1, g1 + 0.5 * g2)
    assert _tile_ran() == 1
    lin = (gs12 - (gs1 + 0.5 * gs2)).abs().max()
    assert float(lin) <= 2e-6 * max(1.0, float(gs12.abs().max()))
    del gs2, gs12
    full = one.expand(D, T, Hs, Ws, 4).contiguous().requires_grad_(True)
    assert full.numel() * 4 > 2**34                                       # really beyond 32-bit byte offsets
    rgb, alpha = render_planes(full, homos, H, W, RenderSpec.mpv())
    assert torch.equal(rgb, rgb1.expand(T, H, W, 3))
    (gs,) = torch.autograd.grad(rgb, full, g1.expand(T, H, W, 3).contiguous())
    assert _tile_ran() == 1
    for t0 in range(0, T, 10):                                            # compare in slabs to bound the temporaries
        assert torch.equal(gs[:, t0:t0 + 10], gs1.expand(D, 10, Hs, Ws, 4))


@pytest.mark.parametrize("stack_scale", [1.0, 1.1, 1.35])
@pytest.mark.parametrize("spec_name", ["mpv", "utils_mpi"])
def test_bwd_2x2_gather_equals_3x3_gather_bitwise(dev, spec_name, stack_scale):
    """tiles on which pixels are >= 1 texel apart (no minification: the reference stores its stacks at 1.1x the frame,
    configs/mpv_base.txt:10-11) gather from the 2x2 staged pixels towards the texel instead of 3x3: the five pixels left out
    have tent weight exactly 0, so the gradient must equal the 3x3 gather's (variant 4) bit for bit -- at 1.0 (benchmark
    cameras: about half of the tiles qualify), 1.1 and 1.35 (all qualify), with plane borders inside the frame."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, T, H, W = 6, 2, 300, 500
    Hs, Ws = int(H * stack_scale) - 7, int(W * stack_scale) - 11          # a little smaller than the footprint: borders in view
    kw_p, _ = SPECS[spec_name]
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=21, device=dev).requires_grad_(True)
    homos = bench_homos(D, H, W).to(dev)
    if spec_name == "mpv":
        kw_p = dict(kw_p, scale=(stack_scale, stack_scale), offset=(-2.0, -3.0))
    else:
        homos = torch.diag(torch.tensor([Ws / W, Hs / H, 1.0])).to(dev) @ homos
    g = synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5
    out = {}
    for variant in (0, 4):
        rgb, _ = render_planes(stack, homos, H, W, RenderSpec(variant=variant, **kw_p))
        (gs,) = torch.autograd.grad(rgb, stack, g)
        assert _tile_ran() == 1
        out[variant] = gs
    assert torch.equal(out[0], out[4])
    assert float(out[0].abs().max()) > 1e-3


@pytest.mark.parametrize("T", [2, 5])
@pytest.mark.parametrize("spec_name", ["mpv", "utils_mpi", "hardcut_pre"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_fwd_frame_pair_kernel_equals_single_frame_kernel_bitwise(dev, spec_name, T, dtype):
    """render_fwd2x_k composites frames t and t+1 per thread (the coordinate half of the instruction stream is paid once): same
    per-frame arithmetic as render_fwd2_k (forward variant 6) -> same bits, for even and odd T (last pair = one frame), with the
    sparsity sums, fp32 and fp16 stacks."""
    from videoloop3d_amd.render import RenderSpec, render_planes_with_regularisers
    D, Hs, Ws, H, W = 5, 70, 150, 61, 139
    kw_p, _ = SPECS[spec_name]
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=31, device=dev, dtype=dtype)
    homos = bench_homos(D, H, W, scale=2.0).to(dev)
    outs = []
    for variant in (0, 0x600):
        rgb, alpha, sums, asum = render_planes_with_regularisers(stack, homos, H, W, RenderSpec(variant=variant, **kw_p))
        outs.append((rgb, alpha, asum))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert float(outs[0][0].abs().max()) > 0.01


@pytest.mark.parametrize("T", [2, 5])
@pytest.mark.parametrize("spec_name,stack_scale", [("mpv", 1.0), ("mpv", 1.1), ("utils_mpi", 1.0), ("hardcut_pre", 1.35)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_bwd_frame_pair_kernel_equals_tile_kernel_bitwise(dev, spec_name, stack_scale, T, dtype):
    """render_bwd_pair_k sweeps and gathers frames t and t+1 per thread (coordinates, owner decode and tent weights are paid
    once): per frame the arithmetic of render_bwd_tile_k (variant 3) in the same order -> the same gradient bits, for even and
    odd T, fp32 and fp16 stacks, windows wider than the 32-texel region (strip pass) and plane borders inside the frame."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    D, H, W = 5, 150, 260
    Hs, Ws = int(H * stack_scale) - 5, int(W * stack_scale) - 9
    kw_p, _ = SPECS[spec_name]
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=41, device=dev, dtype=dtype).requires_grad_(True)
    homos = bench_homos(D, H, W).to(dev)
    if spec_name == "mpv":
        kw_p = dict(kw_p, scale=(stack_scale, stack_scale), offset=(-1.5, -2.5))
    else:
        homos = torch.diag(torch.tensor([Ws / W, Hs / H, 1.0])).to(dev) @ homos
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5, device=dev) - 0.5
    g_a = synth.hash_uniform((T, H, W), seed=6, device=dev) - 0.5
    out = {}
    for variant in (0, 3, 5):          # 0: frame pairs, 3: one frame per thread in 64-wide regions, 5: in 32-wide regions (fp32 planar convention; else = 3)
        rgb, alpha = render_planes(stack, homos, H, W, RenderSpec(variant=variant, **kw_p))
        (gs,) = torch.autograd.grad([rgb, alpha], stack, [g_rgb, g_a])
        assert _tile_ran() == 1
        out[variant] = gs
    assert all(torch.equal(out[v], out[3]) for v in (0, 5))
    assert float(out[0].float().abs().max()) > 1e-3


def test_abi_is_reentrant_across_host_threads_and_streams(dev):
    """SURVEY §8b 'Threading' (the reference drives its module through nn.DataParallel threads): two host threads, each on its
    own HIP stream, call the render ABI concurrently with DIFFERENT stack dtypes, conventions, regulariser / pair / atomics
    variants and sizes -- every dispatch option travels in the call's own argument block, so each thread gets bit for bit what
    the same calls give when issued serially."""
    import threading
    from videoloop3d_amd.render import RenderSpec, render_planes, render_planes_with_regularisers

    def job(idx):
        D, T, Hs, Ws, H, W = [(6, 4, 150, 200, 139, 187), (5, 3, 96, 130, 96, 130)][idx]
        dtype = [torch.float32, torch.float16][idx]
        spec = [RenderSpec(), RenderSpec.mpv()][idx]
        stack = synth.make_plane_stack(D, T, Hs, Ws, seed=40 + idx, device=dev, dtype=dtype).requires_grad_(True)
        homos = bench_homos(D, H, W, scale=1.5).to(dev)
        g = synth.hash_uniform((T, H, W, 3), seed=5 + idx, device=dev) - 0.5
        outs = []
        for rep in range(6):
            if (rep + idx) % 2:
                rgb, alpha, sums, asum = render_planes_with_regularisers(stack, homos, H, W, spec)
                obj = (rgb * g).sum() + 1e-4 * sums.sum() + 1e-3 * asum.sum()
            else:
                v = dataclasses.replace(spec, variant=1) if rep == 4 else spec
                rgb, alpha = render_planes(stack, homos, H, W, v)
                obj = (rgb * g).sum()
            (gs,) = torch.autograd.grad(obj, stack)
            outs.append((rgb.detach().clone(), gs.clone()))
        return outs

    import dataclasses
    serial = [job(0), job(1)]
    torch.cuda.synchronize()
    res, errs = [None, None], []

    def run(idx):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                res[idx] = job(idx)
            st.synchronize()
        except Exception as e:       # surfaced below: an assertion inside a thread would otherwise be lost
            errs.append(e)

    for _ in range(3):
        th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        for idx in range(2):
            for rep, ((rgb_s, gs_s), (rgb_c, gs_c)) in enumerate(zip(serial[idx], res[idx])):
                assert torch.equal(rgb_s, rgb_c), (idx, rep)
                if rep == 4 and idx == 1:      # atomics into an fp16 gradient: order-dependent rounding
                    assert maxabs(gs_s.float(), gs_c.float()) <= 4e-3 * max(1e-3, float(gs_s.float().abs().max())) + 1e-5
                elif rep == 4:
                    assert maxabs(gs_s, gs_c) <= 2e-6 * max(1.0, float(gs_s.abs().max()))
                else:
                    assert torch.equal(gs_s, gs_c), (idx, rep)


@pytest.mark.parametrize("T", [2, 3])
@pytest.mark.parametrize("stack_scale", [1.0, 1.1])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_bwd_pair_reg_kernel_vs_oracle_and_tile_kernel(dev, T, stack_scale, dtype):
    """render_bwd_pair_k<REG> (frame pairs WITH the layer regularisers, decoded from the forward's sign words) -- what a shipped stage-2
    iteration runs (rgb_smooth / a_smooth 0.2 on a 1.1x stack, configs/mpv_base.txt:10-11,33-34):
    gradient vs the oracle's materialised layers (smoothness sums + sparsity sums + composite), and per frame the same bits as the
    one-frame REG tile kernel (variant 3), even and odd T, multi-tile frames with ragged borders."""
    from videoloop3d_amd.render import RenderSpec, render_planes_with_regularisers
    D, H, W = 5, 150, 260
    Hs, Ws = int(H * stack_scale) - 3, int(W * stack_scale) - 5
    kw = dict(MPV_KW, scale=(stack_scale, stack_scale), offset=(-1.0, -2.0))
    stack = (synth.make_plane_stack(D, T, Hs, Ws, seed=43) * 0.6).to(dtype)
    homos = bench_homos(D, H, W, scale=1.3)
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    wts = torch.tensor([1e-3, 2e-3, 3e-3, 1.5e-3])
    g_as = (synth.hash_uniform((T, H, W, 2), seed=8) - 0.5) * 1e-2
    s_cpu = stack.float().requires_grad_(True)
    rgb_o, alpha_o, bw_o, layers = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(**kw), return_layers=True)
    sums_o = torch.stack([(layers[:, :, 1:, :, :3] - layers[:, :, :-1, :, :3]).abs().sum(), (layers[:, 1:, :, :, :3] - layers[:, :-1, :, :, :3]).abs().sum(),
                          (layers[:, :, 1:, :, 3] - layers[:, :, :-1, :, 3]).abs().sum(), (layers[:, 1:, :, :, 3] - layers[:, :-1, :, :, 3]).abs().sum()])
    asum_o = torch.stack([layers[..., 3].sum(-1), (layers[..., 3] ** 2).sum(-1)], -1)
    (gs_o,) = torch.autograd.grad((rgb_o * g_rgb).sum() + (sums_o * wts).sum() + (asum_o * g_as).sum(), s_cpu)
    out = {}
    for variant in (0, 3):
        s_gpu = stack.to(dev).requires_grad_(True)
        rgb, alpha, sums, asum = render_planes_with_regularisers(s_gpu, homos.to(dev), H, W, RenderSpec(variant=variant, **kw))
        (gs,) = torch.autograd.grad((rgb * g_rgb.to(dev)).sum() + (sums * wts.to(dev)).sum() + (asum * g_as.to(dev)).sum(), s_gpu)
        assert _tile_ran() == 1
        out[variant] = gs
    tol = TOL * max(1.0, float(gs_o.abs().max())) if dtype == torch.float32 else 1e-3 * max(1e-3, float(gs_o.abs().max())) + 1e-6
    # |o - o_neighbour| has a kink: where two neighbouring layer values agree to rounding the sign of their difference is decided by
    # the last bit, and a flipped sign moves a texel's gradient by up to 2 * weight * tap weight.  Those few texels aside (the
    # weights here are 10x the other tests' so that a wrong smoothness term could not hide under the tolerance), everything agrees.
    err = (out[0].float().cpu() - gs_o).abs()
    assert float((err > tol).float().mean()) <= 2e-4
    assert float(err.max()) <= 8 * float(wts.max()) + tol
    assert torch.equal(out[0], out[3])


@pytest.mark.parametrize("spec_name", ["mpv", "utils_mpi"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(5, 3, 150, 260, 139, 251), (3, 1, 40, 70, 9, 130), (4, 2, 64, 64, 63, 64)])
def test_fused_forward_with_regularisers_equals_the_two_pass_forward(dev, spec_name, dtype, shape):
    """vl3d_render_fwd_reg (render + smoothness sums in ONE sweep over the stack) against vl3d_render_fwd followed by
    vl3d_render_reg_fwd (variant 0x1000): image, alpha and alpha sums bit for bit, the four sums to double-summation order."""
    from videoloop3d_amd.render import RenderSpec, render_planes_with_regularisers
    D, T, Hs, Ws, H, W = shape
    kw_p, _ = SPECS[spec_name]
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=29, device=dev, dtype=dtype)
    homos = (torch.diag(torch.tensor([Ws / W, Hs / H, 1.0])) @ bench_homos(D, H, W, scale=1.5)).to(dev)
    out = {}
    for variant in (0, 0x1000):
        out[variant] = render_planes_with_regularisers(stack, homos, H, W, RenderSpec(variant=variant, **kw_p))
    for k in (0, 1, 3):
        assert torch.equal(out[0][k], out[0x1000][k]), k
    assert float(((out[0][2] - out[0x1000][2]).abs() / out[0x1000][2].abs().clamp_min(1.0)).max()) <= 1e-6
    assert float(out[0][2].abs().min()) > 0


@pytest.mark.parametrize("T,window", [(1, False), (3, False), (4, True)])
def test_one_pass_culled_forward_with_regularisers_equals_the_two_passes(dev, T, window):
    """vl3d_render_fwd_reg_culled (a tile-culled model's render AND its regulariser sums from one walk over every pixel's covered planes:
    the slot kernel composites as it goes) against vl3d_render_fwd_culled followed by vl3d_render_reg_fwd_culled (variant 0x1000): image,
    alpha and alpha sums bit for bit, the four sums and -- through the sign words both forwards leave for the backward -- the stack
    gradient too.  window: the stack is a texel window of a larger plane the quad grid lies over (crop-aware training)."""
    from videoloop3d_amd.render import RenderSpec, render_planes_with_regularisers
    D, Hs, Ws, H, W = 7, 66, 88, 60, 80
    kw_p, _ = SPECS["mpv"]
    torch.manual_seed(11)
    keep = (torch.rand(D, 6, 8) < 0.45).to(dev)
    keep[2] = False
    shift = torch.tensor([[1.0, 0, -8.0], [0, 1.0, -6.0], [0, 0, 1.0]])
    homos = (bench_homos(D, H, W, scale=2.5) @ shift).to(dev)
    cull_window = (8, 16, Hs + 24, Ws + 40) if window else None
    g_rgb = (synth.hash_uniform((T, H, W, 3), seed=5) - 0.5).to(dev)
    wts = torch.tensor([1.1e-3, 0.7e-3, 1.6e-3, 0.9e-3], device=dev)
    out, grads = {}, {}
    for variant in (0, 0x1000):
        stack = synth.make_plane_stack(D, T, Hs, Ws, seed=29, device=dev).requires_grad_(True)
        rgb, alpha, sums, asum = render_planes_with_regularisers(stack, homos, H, W, RenderSpec(variant=variant, **kw_p), quad_keep=keep,
                                                                 cull_window=cull_window)
        out[variant] = (rgb.detach(), alpha.detach(), sums.detach(), asum.detach())
        ((rgb * g_rgb).sum() + (sums * wts).sum() + 1e-3 * asum.sum()).backward()
        grads[variant] = stack.grad
    for k in (0, 1, 3):
        assert torch.equal(out[0][k], out[0x1000][k]), k
    assert float(((out[0][2] - out[0x1000][2]).abs() / out[0x1000][2].abs().clamp_min(1.0)).max()) <= 1e-6
    assert float(out[0][2].abs().min()) > 0 and float(out[0][1].max()) > 0.1
    assert torch.equal(grads[0], grads[0x1000])


@pytest.mark.parametrize("T", [1, 2, 3])
@pytest.mark.parametrize("mode", ["dense_plane_edges", "sparsified", "sparsified_atomics", "sparsified_f16"])
def test_smoothness_regularisers_are_hit_slot_indexed(dev, mode, T):
    """The reference's layer tensor is indexed by HIT SLOT (MPV.py:386-392, 441-449; utils.py:64-69): slot k of a pixel is its k-th
    nearest COVERED plane, and rgb_smooth / a_smooth difference neighbours per slot (MPV.py:517-531).  Where neighbours are covered
    by different planes -- a plane's edge inside the view (stack 1.1x the frame, off-centre crop) or the quad borders of a sparsified
    model -- that is NOT the per-plane difference.  Sums and stack gradient against the slot oracle; the plane-indexed reading of the
    same layers differs by far more than the tolerance, so the test tells the two apart.  T = 1: one-frame tile kernel, T = 2 / 3:
    frame pairs (even / odd)."""
    from videoloop3d_amd.render import RenderSpec, render_planes_with_regularisers
    D, Hs, Ws, H, W = 7, 66, 88, 60, 80
    kw_p, kw_o = SPECS["mpv"]
    stack = synth.make_plane_stack(D, T, Hs, Ws, seed=31)
    if mode == "sparsified_f16":
        stack = stack.half().float()
    # the crop looks past the stack's upper left corner: the NEAR planes end inside the view while the far ones still cover it, so
    # behind a near plane's edge every slot holds another plane than next to it
    shift = torch.tensor([[1.0, 0, -8.0], [0, 1.0, -6.0], [0, 0, 1.0]])
    homos = bench_homos(D, H, W, scale=2.5) @ shift
    keep = None
    if mode != "dense_plane_edges":
        torch.manual_seed(11)
        keep = torch.rand(D, 6, 8) < 0.45
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    wts = torch.tensor([1.1e-3, 0.7e-3, 1.6e-3, 0.9e-3])

    def sums_of(L):
        return torch.stack([(L[:, :, 1:, :, :3] - L[:, :, :-1, :, :3]).abs().sum(), (L[:, 1:, :, :, :3] - L[:, :-1, :, :, :3]).abs().sum(),
                            (L[:, :, 1:, :, 3] - L[:, :, :-1, :, 3]).abs().sum(), (L[:, 1:, :, :, 3] - L[:, :-1, :, :, 3]).abs().sum()])
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, _, slots = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(**kw_o), return_layers=True, quad_keep=keep)
    sums_o = sums_of(slots)
    (gs_o,) = torch.autograd.grad((rgb_o * g_rgb).sum() + (sums_o * wts).sum(), s_cpu)
    with torch.no_grad():
        planes = MO.render_planes(stack, homos, H, W, MO.RenderSpec(**kw_o), return_layers=True, quad_keep=keep, layer_order="plane")[3]
        # the two readings are far apart here (plane edges are lines, quad borders are everywhere)
        assert float(((sums_of(planes) - sums_o).abs() / sums_o).max()) > (0.003 if keep is None else 0.02)
    variant = 1 if mode == "sparsified_atomics" else 0
    s_gpu = (stack.half() if mode == "sparsified_f16" else stack).to(dev).requires_grad_(True)
    rgb, alpha, sums, _ = render_planes_with_regularisers(s_gpu, homos.to(dev), H, W, RenderSpec(variant=variant, **kw_p),
                                                          quad_keep=None if keep is None else keep.to(dev))
    (gs,) = torch.autograd.grad((rgb * g_rgb.to(dev)).sum() + (sums * wts.to(dev)).sum(), s_gpu)
    assert _tile_ran() == (0 if variant == 1 else 1)
    assert maxabs(rgb, rgb_o) <= TOL and maxabs(alpha, alpha_o) <= TOL
    assert float(((sums.cpu() - sums_o).abs() / sums_o.abs().clamp_min(1.0)).max()) <= 1e-4
    diff = (gs.float().cpu() - gs_o).abs()
    scale = max(1.0, float(gs_o.abs().max()))
    if mode == "sparsified_f16":
        assert float(diff.max()) <= 4e-3 * scale
    else:
        # the sign of a near-zero layer difference may flip with a 1-ulp sampling difference: robust criterion
        assert float((diff > TOL * scale).float().mean()) <= 1e-4 and float(diff.max()) <= 5e-3 * scale
