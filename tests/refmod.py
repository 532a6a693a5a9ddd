"""Shared helpers of the module-level parity tests (goldens G14-G17, tests/golden/make_golden_r04.py).  Pure data handling: nothing here
imports the reference; the scene / argument helpers are the generator's own (so the tests rebuild exactly the cases it recorded)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
if GOLD not in sys.path:
    sys.path.insert(0, GOLD)
import make_golden_r04 as R4  # noqa: E402  (helpers only: scene, make_args, SHAPES, collate -- the reference is imported by its main())


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def state_dict_of(g, prefix, **replace):
    """the reference state_dict a golden holds under `prefix` ("self_*" arrays back to python scalars under "self.*")."""
    sd = {}
    for k in g.files:
        if not k.startswith(prefix):
            continue
        name, v = k[len(prefix):], g[k]
        if name.startswith("self_"):
            sd["self." + name[5:]] = v.item()
        else:
            sd[name] = torch.from_numpy(np.array(v))
    for k, v in replace.items():
        sd[k] = v
    return sd


def case_A():
    """(H, W, args overrides, K, ref_extrin, tar) of the grid shape every G15-G17 case uses."""
    H, W, over = R4.SHAPES["A"]
    K, ref_extrin, tar = R4.scene(H, W)
    return H, W, over, K, ref_extrin, tar


def crop_view(g):
    """the training crop of G17: (h, w, tar_extrin [1,4,4], crop intrinsics [1,3,3], full-frame intrinsics [1,3,3]) float32."""
    h, w = (int(v) for v in g["hw_crop"])
    return h, w, torch.from_numpy(g["tar_extrin"]), torch.from_numpy(g["tar_intrin_crop"]), torch.from_numpy(g["tar_intrin_full"])


REG = dict(sparsity_loss_weight=0.004, rgb_smooth_loss_weight=0.2, a_smooth_loss_weight=0.5, density_loss_weight=0.02)
LOSS_CFGS = {"other": dict(loss_name="gpnn_lm", loss_gain=1.0, patch_size=3, patcht_size=3, stride=2, stridet=1, rou="-2", scaling=0.1,
                           alpha=10000.0, macro_block=9),
             "ref": dict(loss_name="gpnn_lm", loss_gain=3.5, patch_size=7, patcht_size=3, stride=2, stridet=1, rou="-2", scaling=0.1,
                         alpha=0.0, macro_block=11)}
MPV_WEIGHTS = dict(swd=1.0, sparsity=REG["sparsity_loss_weight"], rgb_smooth=REG["rgb_smooth_loss_weight"],
                   a_smooth=REG["a_smooth_loss_weight"], density=REG["density_loss_weight"])


def mpv_args(T, bg="", regs=REG, **kw):
    _, _, over, *_ = case_A()
    return R4.make_args(mpv_frm_num=T, mpv_isloop=True, init_std=0.5, scale_invariant=True, swd_patch_size=3, swd_patcht_size=3,
                        swd_stride=2, swd_stridet=1, bg_color=bg, **over, **regs, **kw)


def mpi_args(**kw):
    _, _, over, *_ = case_A()
    return R4.make_args(**over, **kw)


def lattice_grad_from_reference(sd, hv, wv, D, T, grad_atlas, grad_atlas_dyn):
    """What the reference's gradients w.r.t. its tile atlases say about the gradient of the tile-LATTICE stack (tiles.stack_from_reference_state):
    a lattice texel is every tile texel that sits on it (neighbouring tiles duplicate their border samples), so its gradient is the SUM of
    theirs.  -> (per-frame part from the dynamic atlas [D,T,Hl,Wl,4], frame-summed part from the static atlas [D,Hl,Wl,4])."""
    from videoloop3d_amd import tiles
    parts = [("static", sd["faces"], sd["uvfaces"], sd["uvs"], grad_atlas), ("dyn", sd["faces_dyn"], sd["uvfaces_dyn"], sd["uvs_dyn"], grad_atlas_dyn)]
    (th, tw), lists = tiles._aligned_tiles(parts, hv, wv)
    QH, QW = hv - 1, wv - 1
    Hl, Wl = QH * (th - 1) + 1, QW * (tw - 1) + 1
    out = {"static": torch.zeros((D, Hl, Wl, 1, 4)), "dyn": torch.zeros((D, Hl, Wl, T, 4))}
    iy, ix = torch.arange(th), torch.arange(tw)
    for kind, d, vy, vx, y0, x0, atlas in lists:
        ay = (y0[:, None] + iy[None])[:, :, None].expand(-1, th, tw)
        ax = (x0[:, None] + ix[None])[:, None, :].expand(-1, th, tw)
        tl = atlas[:, :, ay, ax].permute(2, 3, 4, 0, 1)                                              # n,th,tw,A,4
        ly = (vy[:, None] * (th - 1) + iy[None])[:, :, None].expand(-1, th, tw)
        lx = (vx[:, None] * (tw - 1) + ix[None])[:, None, :].expand(-1, th, tw)
        out[kind].index_put_((d[:, None, None].expand(-1, th, tw), ly, lx), tl, accumulate=True)
    return out["dyn"].permute(0, 3, 1, 2, 4).contiguous(), out["static"][:, :, :, 0].contiguous()


def own_grad_from_reference(sd, hv, wv, D, T, grad_atlas, grad_atlas_dyn):
    """the reference's gradients w.r.t. its tile atlases, tile for tile on the TILE-EXACT stack (every tile texel is its own parameter: no sums):
    -> (per-frame gradient of the dynamic tiles [D,T,H,W,4], gradient of the static tiles -- ONE texture for all frames -- [D,H,W,4])."""
    from videoloop3d_amd import tiles
    parts = [("static", sd["faces"], sd["uvfaces"], sd["uvs"], grad_atlas), ("dyn", sd["faces_dyn"], sd["uvfaces_dyn"], sd["uvs_dyn"], grad_atlas_dyn)]
    (th, tw), lists = tiles._aligned_tiles(parts, hv, wv)
    QH, QW = hv - 1, wv - 1
    g_dyn, g_static = torch.zeros((D, T, QH * th, QW * tw, 4)), torch.zeros((D, QH * th, QW * tw, 4))
    for kind, d, vy, vx, y0, x0, atlas in lists:
        for i in range(len(d)):
            tl = atlas[:, :, int(y0[i]):int(y0[i]) + th, int(x0[i]):int(x0[i]) + tw].permute(0, 2, 3, 1)
            ys, xs = slice(int(vy[i]) * th, (int(vy[i]) + 1) * th), slice(int(vx[i]) * tw, (int(vx[i]) + 1) * tw)
            if kind == "dyn":
                g_dyn[int(d[i]), :, ys, xs] = tl
            else:
                g_static[int(d[i]), ys, xs] = tl[0]
    return g_dyn, g_static


OTHER_LOSSES = {"gpnn": dict(loss_name="gpnn", loss_gain=1.5, patch_size=5, patcht_size=3, stride=2, stridet=2, rou="0", scaling=0.2, alpha=0.5),
                "mse": dict(loss_name="mse"), "avg": dict(loss_name="avg", loss_gain=2.0)}
