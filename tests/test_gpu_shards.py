"""BASELINE.json configs[3] and [4] on one MI355X: one rank's shard of the 8-GPU configurations at FULL per-GPU size
(cfg4 = 1080p, D=64, T=80, fp32; cfg5 = 4K, D=96, T=120, fp16 stack AND fp16 gradient resident in the 288 GB), plus the
oracle-parity cases those configurations need that no 720p / D=32 test reaches: D in {64, 65, 96, 128} (the second 64-bit
word of the culling plane masks, csrc/vl3d_render.hip render_fwd2_k CULL), fp16 stacks at D = 96, windows of a 1080p and a
4K frame against the CPU oracle, and the looping loss at the clip lengths of cfg4 / cfg5 at 1080p row width.

The full-size shards are checked through size-independent properties: frames are independent
(utils_mpi.py:159-176 has no cross-frame term), so a shard whose T frames are copies of one frame renders T bit-identical
images and receives T bit-identical gradient frames (no index of the > 100 GB shard wraps), equal to a T=1 call that the
window tests pin against the oracle at the same frame size; the backward is linear in the incoming gradient; nothing is NaN.
"""
import math

import pytest
import torch

from oracle import mpi_oracle as MO
from oracle import vid_oracle as VO
from videoloop3d_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4   # north star: <= 1e-4 max-abs in fp32
MPV_O = dict(pixel_center=0.5, coord_mode="affine", border="hardcut", act_order="post")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import __graft_entry__ as g
    g.build()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _release_hbm():
    yield
    torch.cuda.empty_cache()


def maxabs(a, b):
    return float((a.detach().double().cpu() - torch.as_tensor(b).double().cpu()).abs().max())


def _tile_ran():
    from videoloop3d_amd import render
    return int(render.LAST_BWD_SCRATCH.view(torch.int32)[0].item())


def bench_homos(D, H, W, scale=1.0):
    from videoloop3d_amd.utils_mpi import compute_homography, make_depths
    ref_e, Kr, tar_e, Kt = synth.make_cameras(H, W)
    tar_e = tar_e.clone()
    tar_e[:3, 3] *= scale
    return compute_homography(ref_e[None], Kr[None], tar_e[None], Kt[None], torch.tensor([0., 0., 1.]).expand(1, D, 3),
                              make_depths(D, 1.0, 100.0).flip(0)[None])[0]


def deep_stack(D, T, Hs, Ws, seed, dtype=torch.float32):
    """hash stack with the alpha logits lowered so the transmittance is still ~0.1 behind 128 planes: the planes >= 64
    carry visible weight and gradient (with the init bias of -2 alone T_64 = 3e-4)."""
    s = synth.make_plane_stack(D, T, Hs, Ws, seed=seed)
    s[..., 3] -= 2.0
    return s.to(dtype)


# ---- many planes at small spatial size vs the oracle ---------------------------------------------------------------------
@pytest.mark.parametrize("D", [64, 65, 96, 128])
@pytest.mark.parametrize("spec_name", ["mpv", "utils_mpi"])
def test_many_planes_dense_vs_oracle(dev, D, spec_name):
    """cfg4 / cfg5 plane counts (and the 64 / 65 word boundary), dense stack, both conventions, T odd (frame pairs + tail)."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    T, Hs, Ws, H, W = 3, 44, 70, 40, 66
    kw = MPV_O if spec_name == "mpv" else {}
    stack = deep_stack(D, T, Hs, Ws, seed=31)
    S = torch.tensor([[Ws / W, 0, 0], [0, Hs / H, 0], [0, 0, 1.0]])
    homos = S @ bench_homos(D, H, W, scale=2.0)
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    g_a = synth.hash_uniform((T, H, W), seed=6) - 0.5
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, bw_o = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(**kw))
    (gs_o,) = torch.autograd.grad([rgb_o, alpha_o], s_cpu, [g_rgb, g_a])
    assert float(bw_o[..., 64:].sum()) > 1.0 if D > 64 else True          # the planes behind the word boundary are visible
    s_gpu = stack.to(dev).requires_grad_(True)
    rgb, alpha = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec(**kw))
    (gs,) = torch.autograd.grad([rgb, alpha], s_gpu, [g_rgb.to(dev), g_a.to(dev)])
    assert _tile_ran() == 1
    assert maxabs(rgb, rgb_o) <= TOL and maxabs(alpha, alpha_o) <= TOL
    assert maxabs(gs, gs_o) <= TOL * max(1.0, float(gs_o.abs().max()))
    assert float(gs_o[D - 1].abs().max()) > 0


@pytest.mark.parametrize("D", [65, 96, 128])
@pytest.mark.parametrize("with_reg", [False, True])
def test_many_planes_culled_walks_second_mask_word(dev, D, with_reg):
    """tile culling with kept quads on planes >= 64 only on part of the frame: the forward walks bits of BOTH 64-bit plane
    masks (m0 and m1), the backward skips culled (tile, plane) pairs; with_reg adds the fused layer regularisers (their own
    in-workgroup plane mask).  Against the oracle's quad coverage."""
    from videoloop3d_amd.render import RenderSpec, render_planes, render_planes_with_regularisers
    T, Hs, Ws, H, W = 2, 150, 200, 139, 187
    QH, QW = 6, 9
    torch.manual_seed(D)
    keep = torch.rand(D, QH, QW) < 0.25
    keep[:60:2] = False                      # whole planes culled in the first word
    keep[64:, :, :4] |= torch.rand(D - 64, QH, 4) < 0.5
    keep[D - 1, 2:4, 3:6] = True
    stack = deep_stack(D, T, Hs, Ws, seed=17)
    th = math.radians(2.0)
    Rz = torch.tensor([[math.cos(th) * 1.05, -math.sin(th), 3.0], [math.sin(th), math.cos(th) * 0.96, 2.5], [2e-5, -3e-5, 1.0]])
    homos = bench_homos(D, H, W, scale=1.5) @ Rz
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    s_cpu = stack.clone().requires_grad_(True)
    rgb_o, alpha_o, bw_o, layers = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(**MPV_O), return_layers=True, quad_keep=keep)
    assert float(bw_o[..., 64:].sum()) > 1.0
    sums_o = torch.stack([(layers[:, :, 1:, :, :3] - layers[:, :, :-1, :, :3]).abs().sum(), (layers[:, 1:, :, :, :3] - layers[:, :-1, :, :, :3]).abs().sum(),
                          (layers[:, :, 1:, :, 3] - layers[:, :, :-1, :, 3]).abs().sum(), (layers[:, 1:, :, :, 3] - layers[:, :-1, :, :, 3]).abs().sum()])
    wts = torch.tensor([1e-4, 2e-4, 3e-4, 1.5e-4])
    obj_o = (rgb_o * g_rgb).sum() + ((sums_o * wts).sum() if with_reg else 0.0)
    (gs_o,) = torch.autograd.grad(obj_o, s_cpu)
    s_gpu = stack.to(dev).requires_grad_(True)
    if with_reg:
        rgb, alpha, sums, _ = render_planes_with_regularisers(s_gpu, homos.to(dev), H, W, RenderSpec.mpv(), quad_keep=keep.to(dev))
        assert float(((sums.cpu() - sums_o).abs() / sums_o.abs().clamp_min(1.0)).max()) <= 1e-4
        obj = (rgb * g_rgb.to(dev)).sum() + (sums * wts.to(dev)).sum()
    else:
        rgb, alpha = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec.mpv(), quad_keep=keep.to(dev))
        obj = (rgb * g_rgb.to(dev)).sum()
    (gs,) = torch.autograd.grad(obj, s_gpu)
    assert _tile_ran() == 1
    assert maxabs(rgb, rgb_o) <= TOL and maxabs(alpha, alpha_o) <= TOL
    assert maxabs(gs, gs_o) <= TOL * max(1.0, float(gs_o.abs().max()))


@pytest.mark.parametrize("D", [96, 128])
def test_many_planes_fp16_stack_vs_oracle(dev, D):
    """cfg5's storage (fp16 stack, fp16 gradient) at its plane count: equals the oracle on the rounded values."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    T, Hs, Ws, H, W = 3, 60, 100, 56, 93
    stack16 = deep_stack(D, T, Hs, Ws, seed=21, dtype=torch.float16)
    S = torch.tensor([[Ws / W, 0, 0], [0, Hs / H, 0], [0, 0, 1.0]])
    homos = S @ bench_homos(D, H, W, scale=1.5)
    g_rgb = synth.hash_uniform((T, H, W, 3), seed=5) - 0.5
    s_cpu = stack16.float().requires_grad_(True)
    rgb_o, alpha_o, _ = MO.render_planes(s_cpu, homos, H, W, MO.RenderSpec(**MPV_O))
    (gs_o,) = torch.autograd.grad(rgb_o, s_cpu, g_rgb)
    s_gpu = stack16.to(dev).requires_grad_(True)
    rgb, alpha = render_planes(s_gpu, homos.to(dev), H, W, RenderSpec.mpv())
    (gs,) = torch.autograd.grad(rgb, s_gpu, g_rgb.to(dev))
    assert _tile_ran() == 1 and gs.dtype == torch.float16
    assert maxabs(rgb, rgb_o) <= TOL and maxabs(alpha, alpha_o) <= TOL
    assert maxabs(gs.float(), gs_o) <= 1e-3 * max(1e-3, float(gs_o.abs().max())) + 1e-6      # fp16 rounding of the returned gradient


# ---- a window of a full-size frame vs the oracle -------------------------------------------------------------------------
def _window_vs_oracle(dev, D, H, W, dtype, r0, c0):
    """24x40 window of the full-frame render and of its gradient vs the CPU oracle run on the stack crop the window can reach
    (the crop's texel offset is folded into the affine texel transform; integer window offset into the homography)."""
    from videoloop3d_amd.render import RenderSpec, render_planes
    h, w = 24, 40
    stack = synth.make_plane_stack(D, 1, H, W, seed=2, device=dev, dtype=dtype).requires_grad_(True)
    homos = bench_homos(D, H, W)
    rgb, alpha = render_planes(stack, homos.to(dev), H, W, RenderSpec.mpv())
    gfull = torch.zeros_like(rgb)
    gwin = synth.hash_uniform((1, h, w, 3), seed=5) - 0.5
    gfull[:, r0:r0 + h, c0:c0 + w] = gwin.to(dev)
    (gs,) = torch.autograd.grad(rgb, stack, gfull)
    assert _tile_ran() == 1
    # parallax of the benchmark cameras (synth.make_cameras): <= 0.9 W (0.03 + tan 0.5 deg) ~ 0.035 W along x, 0.009 W along y
    my, mx = int(0.03 * H) + 16, int(0.05 * W) + 16
    lo_r, hi_r, lo_c, hi_c = max(r0 - my, 0), min(r0 + h + my, H), max(c0 - mx, 0), min(c0 + w + mx, W)
    # The oracle runs in fp64 here: at 1080p / 4K coordinates one fp32 ulp of a texel coordinate is 1.2e-4 / 2.4e-4 texels, so two
    # fp32 evaluation orders of the same homography (the kernel's, and the oracle's with the window offset folded into the
    # matrix) differ by about the north star's 1e-4 already; against the exact value the kernel's own error is what is bounded,
    # with the tolerance scaled by the coordinate magnitude relative to the 720p frame the 1e-4 is quoted on.
    crop = stack.detach()[:, :, lo_r:hi_r, lo_c:hi_c].double().cpu().requires_grad_(True)
    shift = torch.tensor([[1.0, 0, c0], [0, 1.0, r0], [0, 0, 1.0]], dtype=torch.float64)
    ospec = MO.RenderSpec(offset=(-float(lo_c), -float(lo_r)), **MPV_O)
    rgb_o, alpha_o, _ = MO.render_planes(crop, homos.double() @ shift, h, w, ospec)
    (gs_o,) = torch.autograd.grad(rgb_o, crop, gwin.double())
    TOLW = TOL * max(1.0, W / 1280)
    # the window never reaches a border of the crop that is not a border of the frame (there the crop's hard cut would differ)
    assert lo_r == 0 or float(gs_o[:, :, :2].abs().max()) == 0
    assert hi_r == H or float(gs_o[:, :, -2:].abs().max()) == 0
    assert lo_c == 0 or float(gs_o[:, :, :, :2].abs().max()) == 0
    assert hi_c == W or float(gs_o[:, :, :, -2:].abs().max()) == 0
    assert maxabs(rgb[:, r0:r0 + h, c0:c0 + w], rgb_o) <= TOLW
    assert maxabs(alpha[:, r0:r0 + h, c0:c0 + w], alpha_o) <= TOLW
    gwin_gpu = gs[:, :, lo_r:hi_r, lo_c:hi_c].float()
    tol_g = TOLW if dtype == torch.float32 else 1e-3 * max(1e-3, float(gs_o.abs().max())) + TOLW
    assert maxabs(gwin_gpu, gs_o) <= tol_g
    # nothing outside the crop received gradient
    total = float(gs.float().abs().sum())
    assert float(gwin_gpu.abs().sum()) == pytest.approx(total, rel=1e-6)
    return rgb.detach(), gs.detach()


def test_1080p_d64_window_vs_oracle(dev):
    """cfg4's frame (1080p, D=64, fp32): windows in the interior and at the lower right corner."""
    _window_vs_oracle(dev, 64, 1080, 1920, torch.float32, 517, 903)
    _window_vs_oracle(dev, 64, 1080, 1920, torch.float32, 1080 - 24, 1920 - 40)


def test_4k_d96_fp16_window_vs_oracle(dev):
    """cfg5's frame (4K, D=96, fp16 stack and gradient): a 6.4 GB single-frame stack, window vs the oracle on the rounded values."""
    _window_vs_oracle(dev, 96, 2160, 3840, torch.float16, 1201, 2377)


# ---- one rank's shard at full per-GPU size ----------------------------------------------------------------------------------
def _shard_properties(dev, D, T, H, W, dtype, N, r, need_gib):
    from videoloop3d_amd.dist import plan_bands, render_band
    from videoloop3d_amd.render import RenderSpec
    free, _ = torch.cuda.mem_get_info(dev)
    if free < need_gib * 2**30:
        pytest.skip(f"needs {need_gib} GiB of free HBM")
    spec = RenderSpec.mpv()
    homos = bench_homos(D, H, W)
    band = plan_bands(homos, H, W, H, N, spec)[r]
    rows = band.src1 - band.src0
    hd = homos.to(dev)
    one = synth.make_plane_stack(D, 1, rows, W, seed=2, device=dev, dtype=dtype)
    g1 = synth.hash_uniform((1, band.rows, W, 3), seed=5, device=dev) - 0.5
    g2 = synth.hash_uniform((1, band.rows, W, 3), seed=6, device=dev) - 0.5
    # T = 1 reference of the same band + linearity of the backward
    s1 = one.clone().requires_grad_(True)
    rgb1, _ = render_band(s1, hd, band, W, H, spec)
    (gs1,) = torch.autograd.grad(rgb1, s1, g1, retain_graph=True)
    assert _tile_ran() == 1
    (gs2,) = torch.autograd.grad(rgb1, s1, g2, retain_graph=True)
    (gs12,) = torch.autograd.grad(rgb1, s1, g1 + 0.5 * g2)
    lin = float((gs12.float() - (gs1.float() + 0.5 * gs2.float())).abs().max())
    scale = max(1.0, float(gs12.float().abs().max()))
    assert lin <= (2e-6 if dtype == torch.float32 else 2e-3) * scale
    assert bool(torch.isfinite(gs1.float()).all()) and float(gs1.float().abs().max()) > 0
    del gs2, gs12, s1
    # the shard: T copies of the frame
    full = one.expand(D, T, rows, W, 4).contiguous().requires_grad_(True)
    nbytes = full.numel() * full.element_size()
    assert nbytes > 2**34                                          # really beyond 32-bit byte offsets
    del one
    rgb, alpha = render_band(full, hd, band, W, H, spec)
    assert torch.equal(rgb, rgb1.expand(T, band.rows, W, 3))
    (gs,) = torch.autograd.grad(rgb, full, g1.expand(T, band.rows, W, 3).contiguous())
    assert _tile_ran() == 1
    step = 8
    for t0 in range(0, T, step):                                   # compare in slabs to bound the temporaries
        n = min(step, T - t0)
        assert torch.equal(gs[:, t0:t0 + n], gs1.expand(D, n, rows, W, 4))
    return nbytes, float(torch.cuda.max_memory_allocated(dev)) / 2**30


def test_cfg4_shard_full_size(dev):
    """BASELINE.json configs[3]: stage-2 MPV 1080p, D=64, T=80, fp32 -- band 3 of 8 (135 frame rows from ~158 stack rows,
    23 GiB of stack + 23 GiB of gradient)."""
    nbytes, _ = _shard_properties(dev, 64, 80, 1080, 1920, torch.float32, 8, 3, need_gib=70)
    assert nbytes > 20 * 2**30


def test_cfg4_shard_border_band(dev):
    """the last band of cfg4 (frame border at the bottom: clamped owner pixels, zero-filled texels beyond the frame)."""
    _shard_properties(dev, 64, 80, 1080, 1920, torch.float32, 8, 7, need_gib=70)


def test_cfg5_shard_full_size(dev):
    """BASELINE.json configs[4]: 4K MPV, D=96, T=120, fp16 plane stack resident in the 288 GB -- band 3 of 8 (270 frame rows
    from ~311 stack rows: 102 GiB of stack + 102 GiB of fp16 gradient, peak ~210 GiB)."""
    torch.cuda.reset_peak_memory_stats(dev)
    nbytes, peak = _shard_properties(dev, 96, 120, 2160, 3840, torch.float16, 8, 3, need_gib=230)
    assert nbytes > 95 * 2**30 and peak < 260


# ---- looping loss at the clip lengths of cfg4 / cfg5, 1080p row width ----------------------------------------------------
def _fold_with_indices(y, nn, Tx, h, w, ps, pt, s, st):
    """the oracle's gather + fold (utils_vid.py:217-229) applied to given NN indices [h_o,w_o,n1]."""
    py = VO.extract_3Dpatches(y, ps, pt, s, st)
    h_, w_, d_ = nn.shape
    Yl = VO._to_location_major(py, h_ * w_, pt, ps)
    picked = Yl[torch.arange(h_ * w_)[:, None], nn.long().reshape(h_ * w_, -1)].reshape(1, h_, w_, d_, 3, pt, ps, ps)
    acc = torch.zeros(1, 4, Tx, h, w)
    for kt in range(pt):
        for kh in range(ps):
            for kw in range(ps):
                v = picked[..., kt, kh, kw].permute(0, 4, 3, 1, 2)
                acc[:, :3, kt:kt + st * d_:st, kh:kh + s * h_:s, kw:kw + s * w_:s] += v
                acc[:, 3:, kt:kt + st * d_:st, kh:kh + s * h_:s, kw:kw + s * w_:s] += 1
    return acc[:, :3] / acc[:, 3:].clamp_min(1e-10), acc[:, 3:].clamp_min(1e-10)


@pytest.mark.parametrize("T,Ty", [(80, 120), (120, 180)])
@pytest.mark.parametrize("cfg_name", ["ref", "other"])
def test_loss_cfg4_cfg5_clip_lengths_1080p_rows(dev, T, Ty, cfg_name):
    """cfg4 (T=80) and cfg5 (T=120) clips (x has T+2 frames after loop padding, MPV.py:490-492; captured clip 1.5x as long, as
    in the cfg3 bench) on a strip of full 1080p width: the long-clip instantiations of the NN kernel (512 / 1024 threads) and the
    narrow fold tiles, both shipped loss configurations, against the oracle."""
    from test_gpu_loss import nn_mismatch_is_near_tie
    from videoloop3d_amd.utils_vid import Patch3DGPNNLowMemLoss, find_nn_indices
    ps, s, al = (11, 4, 0.5) if cfg_name == "ref" else (3, 2, None)
    pt, st = 3, 1
    h, w = ps + 2 * s, (1920 - ps) // s * s + ps
    Tx = T + 2
    x = synth.make_video(Tx, h, w, seed=3)
    y = synth.make_video(Ty, h, w, seed=4)
    xg = x.to(dev).requires_grad_(True)
    L = Patch3DGPNNLowMemLoss()
    loss = L(xg, y.to(dev), patch_size=ps, stride=s, patcht_size=pt, stridet=st, rou="-2", scaling=0.1,
             alpha=1e10 if al is None else al, macro_block=65)
    (gx,) = torch.autograd.grad(loss, xg)
    nn_gpu = find_nn_indices(x.to(dev), y.to(dev), ps, pt, s, st, al)[0]
    nbad, unexplained = nn_mismatch_is_near_tie(x, y, ps, pt, s, st, al, nn_gpu)
    assert unexplained == 0 and nbad <= 0.01 * nn_gpu.numel()
    y2x_o, w_o = _fold_with_indices(y, nn_gpu.cpu(), Tx, h, w, ps, pt, s, st)
    assert float((L.last_weight.cpu() - w_o).abs().max()) == 0.0
    assert float((L.last_y2x.cpu() - y2x_o).abs().max()) <= 1e-5
    xr = x.clone().requires_grad_(True)
    loss_r = VO.robust_lossfun(xr - L.last_y2x.cpu(), "-2", 0.1).mean()
    (gx_r,) = torch.autograd.grad(loss_r, xr)
    assert abs(float(loss.detach()) - float(loss_r)) <= 1e-5 * max(1.0, abs(float(loss_r)))
    assert float((gx.cpu() - gx_r).abs().max()) <= 1e-5 * max(1.0, float(gx_r.abs().max()))
