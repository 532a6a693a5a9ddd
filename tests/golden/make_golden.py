#!/usr/bin/env python3
"""Generate the golden vectors G1..G9 by IMPORTING the reference operators.

Runs only in the build container (needs /root/reference); the GPU box receives the
resulting .npz files, never the reference. Usage:  python tests/golden/make_golden.py

The three missing third-party modules of utils_vid.py are replaced by stand-ins with the
documented im2col / col2im semantics: channel order (C,kt,kh,kw), location
order (d,h,w), fold = exact adjoint of unfold.  They exist in this script only.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def _install_standins():
    m = types.ModuleType("unfoldNd")

    class UnfoldNd:
        def __init__(self, kernel_size, dilation=1, padding=0, stride=1):
            self.k, self.s = tuple(kernel_size), tuple(stride)

        def __call__(self, x):
            (kt, kh, kw), (st, sh, sw) = self.k, self.s
            n, c = x.shape[:2]
            p = x.unfold(2, kt, st).unfold(3, kh, sh).unfold(4, kw, sw)  # n,c,dT,dH,dW,kt,kh,kw
            p = p.permute(0, 1, 5, 6, 7, 2, 3, 4)
            return p.reshape(n, c * kt * kh * kw, -1)

    class FoldNd:
        def __init__(self, output_size, kernel_size, dilation=1, padding=0, stride=1):
            self.o, self.k, self.s = tuple(output_size), tuple(kernel_size), tuple(stride)

        def __call__(self, cols):
            (T, H, W), (kt, kh, kw), (st, sh, sw) = self.o, self.k, self.s
            n = cols.shape[0]
            c = cols.shape[1] // (kt * kh * kw)
            dT, dH, dW = (T - kt) // st + 1, (H - kh) // sh + 1, (W - kw) // sw + 1
            cols = cols.reshape(n, c, kt, kh, kw, dT, dH, dW)
            out = torch.zeros(n, c, T, H, W, dtype=cols.dtype)
            for a in range(kt):
                for b in range(kh):
                    for d in range(kw):
                        out[:, :, a:a + st * dT:st, b:b + sh * dH:sh, d:d + sw * dW:sw] += cols[:, :, a, b, d]
            return out

    m.UnfoldNd, m.FoldNd = UnfoldNd, FoldNd
    sys.modules["unfoldNd"] = m
    ms = types.ModuleType("pytorch_msssim")
    ms.ssim = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("ssim off-path"))
    sys.modules["pytorch_msssim"] = ms
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvt.Resize = object
    tvt.InterpolationMode = object
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt


def small_pose(rng, scale=0.05):
    """random small-motion 4x4 extrinsic (float32)."""
    w = rng.normal(size=3) * scale
    th = np.linalg.norm(w) + 1e-12
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = rng.normal(size=3) * scale
    return E.astype(np.float32)


def main():
    warnings.simplefilter("ignore")
    torch.set_num_threads(8)
    _install_standins()
    sys.path.insert(0, REF)
    import utils_mpi as R  # noqa  (reference, as-is)
    import utils_vid as V  # noqa  (reference, with stand-ins above)
    from videoloop3d_amd import synth

    rng = np.random.default_rng(20260928)
    sv = lambda name, **kw: np.savez_compressed(os.path.join(HERE, name), **kw)
    T = lambda a: torch.from_numpy(np.asarray(a))

    # ---- G1 compute_homography: two B=1 cases, D=8 ---------------------------------------
    # (the reference broadcasts translation[B,3,1] @ normal[B,D,1,3], which only works for B == 1:
    #  utils_mpi.py:270-271 — so goldens are B=1, normal [1,D,3])
    D = 8
    depths = R.make_depths(D, 1.0, 100.0).numpy()
    g1 = {"depths": depths}
    for ci in range(2):
        src_ext = small_pose(rng)[None]
        tar_ext = small_pose(rng)[None]
        Ks = np.array([[[30., 0, 20.], [0, 31., 12.], [0, 0, 1]]], np.float32)
        Kt = np.array([[[28., 0, 18.], [0, 28., 10.], [0, 0, 1]]], np.float32)
        nrm = np.array([0, 0, 1.], np.float32) if ci == 0 else np.array([0.05, -0.02, 1.], np.float32)
        nrm = nrm / np.linalg.norm(nrm)
        normal = np.tile(nrm.astype(np.float32), (1, D, 1))
        dist = depths[None].astype(np.float32)
        homo = R.compute_homography(T(src_ext), T(Ks), T(tar_ext), T(Kt), T(normal), T(dist))
        g1.update({f"c{ci}_src_ext": src_ext, f"c{ci}_tar_ext": tar_ext, f"c{ci}_src_K": Ks, f"c{ci}_tar_K": Kt,
                   f"c{ci}_normal": normal, f"c{ci}_dist": dist, f"c{ci}_homo": homo.numpy()})
    sv("g1_homography.npz", **g1)

    # ---- G2 warp_homography: B=2, D=4, C=4, src 24x40 -> out 20x36 (+ grad wrt images) -------
    B, D, C, Hs, Ws, h, w = 2, 4, 4, 24, 40, 20, 36
    Ks2 = np.array([[[36., 0, 20.], [0, 36., 12.], [0, 0, 1]]], np.float32)
    Kt2 = np.array([[[34., 0, 18.], [0, 34., 10.], [0, 0, 1]]], np.float32)
    se = np.eye(4, dtype=np.float32)[None]
    normal = np.tile(np.array([0, 0, 1.], np.float32), (1, D, 1))
    d4 = R.make_depths(D, 1.0, 20.0).numpy()[None].astype(np.float32)
    homos = torch.cat([R.compute_homography(T(se), T(Ks2), T(small_pose(rng, 0.03)[None]), T(Kt2), T(normal), T(d4))
                       for _ in range(B)], 0)
    images = torch.from_numpy(rng.normal(size=(B, D, C, Hs, Ws)).astype(np.float32)).requires_grad_(True)
    out = R.warp_homography(h, w, homos, images)
    gout = torch.from_numpy(rng.normal(size=tuple(out.shape)).astype(np.float32))
    (gi,) = torch.autograd.grad(out, images, gout)
    sv("g2_warp.npz", homos=homos.numpy(), images=images.detach().numpy(), out=out.detach().numpy(),
       grad_out=gout.numpy(), grad_images=gi.numpy(), h=h, w=w)

    # ---- G3 overcompose & overcomposeNto0: [2,16,20,D=8] ----------------------------------
    B, Hh, Ww, D = 2, 16, 20, 8
    alpha = torch.from_numpy(rng.uniform(0.02, 0.95, size=(B, Hh, Ww, D)).astype(np.float32)).requires_grad_(True)
    content = torch.from_numpy(rng.normal(size=(B, Hh, Ww, D, 3)).astype(np.float32)).requires_grad_(True)
    rgb, bw = R.overcompose(alpha, content)
    g_rgb = torch.from_numpy(rng.normal(size=tuple(rgb.shape)).astype(np.float32))
    g_bw = torch.from_numpy(rng.normal(size=tuple(bw.shape)).astype(np.float32))
    ga, gc = torch.autograd.grad([rgb, bw], [alpha, content], [g_rgb, g_bw])
    mpi = torch.from_numpy(np.concatenate([rng.normal(size=(B, D, 3, Hh, Ww)),
                                           rng.uniform(0.02, 0.95, size=(B, D, 1, Hh, Ww))], 2).astype(np.float32))
    mpi.requires_grad_(True)
    rgbN, bwN = R.overcomposeNto0(mpi, ret_mask=True)
    g_rgbN = torch.from_numpy(rng.normal(size=tuple(rgbN.shape)).astype(np.float32))
    (gm,) = torch.autograd.grad(rgbN, mpi, g_rgbN)
    sv("g3_overcompose.npz", alpha=alpha.detach().numpy(), content=content.detach().numpy(),
       rgb=rgb.detach().numpy(), blendweight=bw.detach().numpy(), g_rgb=g_rgb.numpy(), g_bw=g_bw.numpy(),
       grad_alpha=ga.numpy(), grad_content=gc.numpy(),
       mpi=mpi.detach().numpy(), rgbN=rgbN.detach().numpy(), bwN=bwN.detach().numpy(),
       g_rgbN=g_rgbN.numpy(), grad_mpi=gm.numpy())

    # ---- G4 end-to-end cfg1: sigmoid -> warp -> compositeNto0, D=8, 256x256 ----------------
    # inputs come from the integer-hash generator (rebuilt in the tests), only outputs are stored.
    D, Hh, Ww = 8, 256, 256
    stack = synth.make_plane_stack(D, 1, Hh, Ww, seed=2)            # (D,1,H,W,4) pre-activation
    ref_e, Kr, tar_e, Kt4 = synth.make_cameras(Hh, Ww)
    dep = R.make_depths(D, 1.0, 100.0).flip(0)                      # plane 0 = nearest (MPV.py:51)
    homos4 = R.compute_homography(ref_e[None], Kr[None], tar_e[None], Kt4[None],
                                  torch.tensor([0., 0., 1.]).expand(1, D, 3), dep[None])  # [1,D,3,3]
    act = torch.sigmoid(stack[:, 0].permute(0, 3, 1, 2))[None].clone().requires_grad_(True)  # [1,D,4,H,W]
    warped = R.warp_homography(Hh, Ww, homos4, act)
    # overcomposeNto0 has front = LAST index -> feed planes far..near
    rgb4 = R.overcomposeNto0(warped.flip(1))                        # [1,3,H,W]
    g4 = synth.hash_uniform(tuple(rgb4.shape), seed=7) - 0.5
    (gact,) = torch.autograd.grad(rgb4, act, g4)
    # chain through sigmoid to the pre-activation stack: d/ds = g * a(1-a)
    a_ = act.detach()
    gstack = (gact * a_ * (1 - a_))[0].permute(0, 2, 3, 1)           # (D,H,W,4)
    sv("g4_cfg1_render.npz", homos=homos4[0].numpy(), rgb=rgb4[0].detach().numpy(),
       grad_stack_crop=gstack[:, 96:160, 96:160].numpy().copy(),
       grad_stack_sum=np.array([float(gstack.double().sum()), float(gstack.double().abs().sum())]))

    # ---- G5 extract_3Dpatches ordering on a 1x3x5x9x9 ramp ----------------------------------
    ramp = torch.arange(3 * 5 * 9 * 9, dtype=torch.float32).reshape(1, 3, 5, 9, 9)
    p5 = V.extract_3Dpatches(ramp, 3, 3, 2, 1)
    p5b = V.extract_3Dpatches(ramp, 5, 2, 4, 2)
    sv("g5_patches.npz", p_3_3_2_1=p5.numpy(), p_5_2_4_2=p5b.numpy())

    # ---- G6 get_NN_indices_low_memory ------------------------------------------------------
    X = torch.from_numpy(rng.uniform(size=(6, 7, 3, 3, 5, 5)).astype(np.float32))
    Y = torch.from_numpy(rng.uniform(size=(6, 9, 3, 3, 5, 5)).astype(np.float32))
    nn_none = V.get_NN_indices_low_memory(X, Y, None, 1024)
    nn_a = V.get_NN_indices_low_memory(X, Y, 0.5, 1024)
    nn_a2 = V.get_NN_indices_low_memory(X, Y, 0.005, 4)
    distXY = V.efficient_compute_distances(X, Y)
    sv("g6_nn.npz", X=X.numpy(), Y=Y.numpy(), nn_none=nn_none.numpy(), nn_alpha05=nn_a.numpy(),
       nn_alpha0005=nn_a2.numpy(), dist=distXY.numpy())

    # ---- G7 FindNNpatchAndMerge ------------------------------------------------------------
    x7 = torch.from_numpy(rng.uniform(size=(1, 3, 8, 17, 17)).astype(np.float32))
    y7 = torch.from_numpy(rng.uniform(size=(1, 3, 12, 17, 17)).astype(np.float32))
    g7 = {"x": x7.numpy(), "y": y7.numpy()}
    for (ps, pt, s, st, al) in [(5, 3, 2, 1, 1e10), (3, 3, 2, 1, 1e10), (5, 3, 2, 1, 0.5), (3, 2, 1, 2, 0.05)]:
        s_, w_ = V.FindNNpatchAndMerge(x7, y7, patch_size=ps, patcht_size=pt, stride=s, stridet=st, alpha=al)
        key = f"ps{ps}_pt{pt}_s{s}_st{st}_a{al:g}"
        g7[key + "_sum"] = s_.numpy()
        g7[key + "_weight"] = w_.numpy()
    sv("g7_merge.npz", **g7)

    # ---- G8 Patch3DGPNNLowMemLoss value+grad for both shipped cfgs -----------------------------
    x8 = torch.from_numpy(rng.uniform(size=(1, 3, 12, 35, 35)).astype(np.float32))
    y8 = torch.from_numpy(rng.uniform(size=(1, 3, 20, 35, 35)).astype(np.float32))
    g8 = {"x": x8.numpy(), "y": y8.numpy()}
    cfgs = {
        "ref": dict(macro_block=19, patch_size=11, stride=4, patcht_size=3, stridet=1, rou='-2', scaling=0.1,
                    alpha=0.5, dist_fn='mse'),
        "other": dict(macro_block=17, patch_size=3, stride=2, patcht_size=3, stridet=1, rou='-2', scaling=0.1,
                      alpha=10000, dist_fn='mse'),
        "trim": dict(macro_block=16, patch_size=5, stride=3, patcht_size=3, stridet=2, rou=0, scaling=0.2,
                     alpha=10000, dist_fn='mse'),
    }
    for name, cfg in cfgs.items():
        xx = x8.clone().requires_grad_(True)
        lm = V.Patch3DGPNNLowMemLoss()
        loss = lm(xx, y8, **cfg)
        (gx,) = torch.autograd.grad(loss, xx)
        g8[name + "_loss"] = np.float32(loss.item())
        g8[name + "_grad"] = gx.numpy()
        g8[name + "_y2x"] = lm.last_y2x.numpy()
        g8[name + "_weight"] = lm.last_weight.numpy()
        if name != "trim":
            dl = V.Patch3DGPNNDirectLoss()
            dcfg = {k: v for k, v in cfg.items() if k != "macro_block"}
            g8[name + "_direct_loss"] = np.float32(dl(x8, y8, **dcfg).item())
    sv("g8_loss.npz", **g8)

    # ---- G9 robust_lossfun -----------------------------------------------------------------
    xs = torch.linspace(-2, 2, 41, dtype=torch.float32)
    g9 = {"x": xs.numpy()}
    for rou in ['mse', 'abs', '0', '2', '-2', '1']:
        for sc in [0.1, 0.2]:
            xr = xs.clone().requires_grad_(True)
            val = V.robust_lossfun(xr, rou, sc)
            (gr,) = torch.autograd.grad(val.sum(), xr)
            g9[f"rou{rou}_s{sc}"] = val.detach().numpy()
            g9[f"rou{rou}_s{sc}_grad"] = gr.numpy()
    sv("g9_robust.npz", **g9)

    # ---- G10 evaluations/NNMSE.compute_nnerr -----------------------------------------
    from evaluations.NNMSE import compute_nnerr
    x10 = torch.from_numpy(rng.uniform(size=(1, 3, 9, 29, 37)).astype(np.float32))
    y10 = torch.from_numpy(rng.uniform(size=(1, 3, 13, 29, 37)).astype(np.float32))
    g10 = {"x": x10.numpy(), "y": y10.numpy()}
    for (ps, s_, pt, st, mb) in [(5, 2, 3, 1, 13), (7, 2, 3, 2, 65), (3, 1, 3, 1, 9), (11, 4, 3, 1, 19)]:
        g10[f"ps{ps}_s{s_}_pt{pt}_st{st}_mb{mb}"] = np.float64(compute_nnerr(x10, y10, ps, s_, pt, st, mb))
    sv("g10_nnerr.npz", **g10)

    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print("golden fixtures written, total bytes:", tot)


if __name__ == "__main__":
    main()
