#!/usr/bin/env python3
"""Round-3 golden vectors, generated like G1..G12 by IMPORTING the reference (tests/golden/make_golden.py holds the stand-ins for the
three absent third-party modules and the recipe).  Runs only in the build container.

G13  the looping loss at the SHIPPED ref-view normaliser alpha = 0 (configs/mpv_base.txt:52 `swd_alpha_ref = 0`; utils_vid.py:122-142):
     a  get_NN_indices_low_memory(X, Y, 0, chunk) on the G6 patches
     b  FindNNpatchAndMerge(alpha=0) on the G7 videos
     c  Patch3DGPNNLowMemLoss ref-view cfg with alpha=0 on the G8 videos: loss / grad / y2x / weight, and == DirectLoss
     d  the same on clips with MORE y frames than x patches (n2 > n1: by pigeonhole some x patch is the column minimum of several y
        patches, i.e. the score 1.0 = (m/d)/(0 + m/d) is tied EXACTLY and torch.argmin's first-minimum rule decides -- systematic at
        alpha = 0, not a measure-zero event) with the per-location NN indices recovered from the reference's own functions
     e  a deliberately DEGENERATE case (y holds exact copies of x frames: column minima are ~0, the reference's normalised distances
        hold NaN / +-inf / negative values): what the reference returns is RECORDED (indices, finiteness) so the HIP
        path's behaviour there is a documented decision, not a parity target.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def ref_nn_indices(V, x, y, ps, pt, s, st, alpha):
    """per-location NN indices [h_o, w_o, n1] through the reference's own extract_3Dpatches + get_NN_indices_low_memory
    (the first lines of FindNNpatchAndMerge, utils_vid.py:209-216)."""
    px = V.extract_3Dpatches(x, ps, pt, s, st)
    b, c, d, h, w = px.shape
    B = b * h * w
    px = px.permute(0, 3, 4, 2, 1).reshape(B, -1, 3, pt, ps, ps)
    py = V.extract_3Dpatches(y, ps, pt, s, st).permute(0, 3, 4, 2, 1).reshape(B, -1, 3, pt, ps, ps)
    nn = V.get_NN_indices_low_memory(px, py, alpha, 1024)
    dist = V.efficient_compute_distances(px, py)
    return nn.reshape(h, w, d), dist.reshape(h, w, d, -1)


def main():
    warnings.simplefilter("ignore")
    torch.set_num_threads(8)
    MG._install_standins()
    sys.path.insert(0, MG.REF)
    import utils_vid as V  # noqa  (reference, with the stand-ins)
    from videoloop3d_amd import synth

    out = {}
    # ---- a: G6 patches ------------------------------------------------------------------------------------------------
    g6 = np.load(os.path.join(HERE, "g6_nn.npz"))
    X, Y = torch.from_numpy(g6["X"]), torch.from_numpy(g6["Y"])
    out["a_nn_alpha0"] = V.get_NN_indices_low_memory(X, Y, 0, 1024).numpy()
    out["a_nn_alpha0_chunk4"] = V.get_NN_indices_low_memory(X, Y, 0, 4).numpy()
    # ---- b: G7 videos -------------------------------------------------------------------------------------------------
    g7 = np.load(os.path.join(HERE, "g7_merge.npz"))
    x7, y7 = torch.from_numpy(g7["x"]), torch.from_numpy(g7["y"])
    for (ps, pt, s, st) in [(5, 3, 2, 1), (3, 3, 2, 1), (3, 2, 1, 2)]:
        s_, w_ = V.FindNNpatchAndMerge(x7, y7, patch_size=ps, patcht_size=pt, stride=s, stridet=st, alpha=0)
        out[f"b_ps{ps}_pt{pt}_s{s}_st{st}_sum"], out[f"b_ps{ps}_pt{pt}_s{s}_st{st}_weight"] = s_.numpy(), w_.numpy()
    # ---- c: G8 videos, the shipped ref-view configuration (alpha = 0) ---------------------------------------------------
    g8 = np.load(os.path.join(HERE, "g8_loss.npz"))
    x8, y8 = torch.from_numpy(g8["x"]), torch.from_numpy(g8["y"])
    cfg = dict(macro_block=19, patch_size=11, stride=4, patcht_size=3, stridet=1, rou='-2', scaling=0.1, alpha=0, dist_fn='mse')
    xx = x8.clone().requires_grad_(True)
    lm = V.Patch3DGPNNLowMemLoss()
    loss = lm(xx, y8, **cfg)
    (gx,) = torch.autograd.grad(loss, xx)
    out.update(c_loss=np.float32(loss.item()), c_grad=gx.numpy(), c_y2x=lm.last_y2x.numpy(), c_weight=lm.last_weight.numpy())
    out["c_direct_loss"] = np.float32(V.Patch3DGPNNDirectLoss()(x8, y8, **{k: v for k, v in cfg.items() if k != "macro_block"}).item())
    nn, dist = ref_nn_indices(V, x8, y8, 11, 3, 4, 1, 0)
    out["c_nn"], out["c_dist"] = nn.numpy().astype(np.int32), dist.numpy()
    # ---- d: more y patches than x patches, both shipped patch shapes, alpha = 0 -----------------------------------------
    xd = synth.make_video(9, 23, 27, seed=51)       # n1 = 7
    yd = synth.make_video(21, 23, 27, seed=52)      # n2 = 19
    out["d_x"], out["d_y"] = xd.numpy(), yd.numpy()
    for (ps, s) in [(11, 4), (3, 2)]:
        nn, dist = ref_nn_indices(V, xd, yd, ps, 3, s, 1, 0)
        out[f"d_ps{ps}_nn"], out[f"d_ps{ps}_dist"] = nn.numpy().astype(np.int32), dist.numpy()
        xr = xd.clone().requires_grad_(True)
        lm = V.Patch3DGPNNLowMemLoss()
        loss = lm(xr, yd, macro_block=ps + 2 * s, patch_size=ps, stride=s, patcht_size=3, stridet=1, rou='-2', scaling=0.1, alpha=0)
        (g,) = torch.autograd.grad(loss, xr)
        out[f"d_ps{ps}_loss"], out[f"d_ps{ps}_grad"], out[f"d_ps{ps}_y2x"] = np.float32(loss.item()), g.numpy(), lm.last_y2x.numpy()
        # how many rows are decided by an EXACT tie at the minimum score (the first-minimum rule at work)
        sc = dist / (0 + dist.min(2, keepdim=True)[0])
        ties = (sc == sc.min(3, keepdim=True)[0]).sum(3) > 1
        out[f"d_ps{ps}_tied_rows"] = np.int64(ties.sum().item())
    # ---- e: degenerate -- y contains exact copies of x frames -----------------------------------------------------------
    xe = synth.make_video(8, 15, 15, seed=53)
    ye = torch.cat([synth.make_video(4, 15, 15, seed=54), xe[:, :, 1:7], synth.make_video(3, 15, 15, seed=55)], dim=2)
    out["e_x"], out["e_y"] = xe.numpy(), ye.numpy()
    nn, dist = ref_nn_indices(V, xe, ye, 5, 3, 2, 1, 0)
    colmin = dist.min(2)[0]
    sc = dist / (0 + colmin[:, :, None])
    out.update(e_nn=nn.numpy().astype(np.int32), e_dist=dist.numpy(), e_colmin_min=np.float32(colmin.min().item()),
               e_colmin_neg=np.int64((colmin < 0).sum().item()), e_colmin_zero=np.int64((colmin == 0).sum().item()),
               e_score_nan=np.int64(torch.isnan(sc).sum().item()), e_score_inf=np.int64(torch.isinf(sc).sum().item()),
               e_score_neg=np.int64((sc < 0).sum().item()))
    lm = V.Patch3DGPNNLowMemLoss()
    le = lm(xe, ye, macro_block=9, patch_size=5, stride=2, patcht_size=3, stridet=1, rou='-2', scaling=0.1, alpha=0)
    out.update(e_loss=np.float32(le.item()), e_y2x=lm.last_y2x.numpy())
    np.savez_compressed(os.path.join(HERE, "g13_alpha0.npz"), **out)
    print("wrote g13_alpha0.npz:", {k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items()
                                    if k.endswith(("loss", "rows", "nan", "inf", "neg", "zero", "min"))})


if __name__ == "__main__":
    main()
