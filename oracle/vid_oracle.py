"""CPU oracle for the looping-loss path (utils_vid.py).  TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32 restatement of the reference's per-location temporal patch nearest-neighbour
looping loss.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it.

Pinning: checked against golden vectors G5-G9 produced by importing the reference's utils_vid.py
in the build container (tests/golden/make_golden.py).  The reference's im2col/col2im lives in the
un-vendored, un-pinned third-party package `unfoldNd` (requirements.txt:7; call sites
utils_vid.py:66-68, 218-227): the goldens were generated with a stand-in implementing its documented
semantics (channel order (C,kt,kh,kw), location order (d,h,w), fold = adjoint of unfold), so parity is
"unpinned" w.r.t. the real unfoldNd binary and pinned w.r.t. those semantics.

Reference citations are into /root/reference.
"""
import warnings

import torch


def robust_lossfun(x, rou, scale, eps=1e-6):
    """utils_vid.py:10-26 (Barron general robust loss; `rou` may arrive as a string)."""
    if rou == "mse":
        return x ** 2
    if rou == "abs":
        return x.abs()
    rou = float(rou)
    s = (x / scale) ** 2
    if rou == 0:
        return torch.log1p(s * 0.5)
    if rou == 2:
        return 0.5 * s
    b = abs(rou - 2) + eps
    d = rou + eps if rou >= 0 else rou - eps
    return (b / d) * (torch.pow(s / b + 1.0, 0.5 * d) - 1.0) * (scale * 10)


def extract_3Dpatches(x, ps, pt, stride, tstride):
    """utils_vid.py:60-69: 3-D im2col.  [b,3,T,h,w] -> [b, 3*pt*ps*ps, d_out, h_out, w_out],
    channel order (c,kt,kh,kw)."""
    b, c = x.shape[:2]
    p = x.unfold(2, pt, tstride).unfold(3, ps, stride).unfold(4, ps, stride)   # b,c,dT,dH,dW,kt,kh,kw
    dT, dH, dW = p.shape[2:5]
    return p.permute(0, 1, 5, 6, 7, 2, 3, 4).reshape(b, c * pt * ps * ps, dT, dH, dW)


def patch_distances(X, Y):
    """utils_vid.py:72-86: (|x|^2 + |y|^2 - 2 x.y) / d, X [B,n1,...], Y [B,n2,...] -> [B,n1,n2]."""
    X = X.reshape(*X.shape[:2], -1)
    Y = Y.reshape(*Y.shape[:2], -1)
    dist = (X * X).sum(-1)[:, :, None] + (Y * Y).sum(-1)[:, None, :] - 2.0 * (X @ Y.permute(0, 2, 1))
    return dist / X.shape[-1]


def patch_distances_exact(X, Y):
    """Cancellation-free fp64 distances (for the top-2 gap criterion of the parity tests)."""
    X = X.reshape(*X.shape[:2], -1).double()
    Y = Y.reshape(*Y.shape[:2], -1).double()
    return ((X[:, :, None] - Y[:, None]) ** 2).mean(-1)


def nn_indices(X, Y, alpha):
    """utils_vid.py:109-142: optional column-min normalisation, then row argmin (first minimum)."""
    dist = patch_distances(X, Y)
    if alpha is not None:
        dist = dist / (alpha + dist.min(1)[0][:, None])
    return torch.argmin(dist, dim=2)


def _to_location_major(p, B, pt, ps):
    """utils_vid.py:213: [b,C,d,h,w] -> [B=b*h*w, d, 3, pt, ps, ps]."""
    return p.permute(0, 3, 4, 2, 1).reshape(B, -1, 3, pt, ps, ps)


def find_nn_and_merge(x, y, patch_size=7, patcht_size=7, stride=1, stridet=1, alpha=1e10, return_nn=False):
    """utils_vid.py:206-229: NN search per spatial location, gather, vote-fold.
    Returns (y2x_sum [1,3,T,h,w], weight [1,1,T,h,w] clamped at 1e-10)."""
    alpha = None if alpha > 100 else alpha
    ps, pt = patch_size, patcht_size
    px = extract_3Dpatches(x, ps, pt, stride, stridet)
    b, c, d, h, w = px.shape
    B = b * h * w
    X = _to_location_major(px, B, pt, ps)
    Y = _to_location_major(extract_3Dpatches(y, ps, pt, stride, stridet), B, pt, ps)
    nns = nn_indices(X, Y, alpha)                                           # [B, d]
    picked = Y[torch.arange(B)[:, None], nns]                              # [B, d, 3, pt, ps, ps]
    picked = picked.reshape(b, h, w, d, 3, pt, ps, ps)
    T, Hh, Ww = x.shape[-3:]
    acc = torch.zeros(b, 4, T, Hh, Ww, dtype=x.dtype)
    for kt in range(pt):
        for kh in range(ps):
            for kw in range(ps):
                v = picked[..., kt, kh, kw].permute(0, 4, 3, 1, 2)         # b,3,d,h,w
                sl = (slice(None), slice(0, 3), slice(kt, kt + stridet * d, stridet),
                      slice(kh, kh + stride * h, stride), slice(kw, kw + stride * w, stride))
                acc[sl] += v
                acc[:, 3:, kt:kt + stridet * d:stridet, kh:kh + stride * h:stride, kw:kw + stride * w:stride] += 1
    out = (acc[:, :3], acc[:, 3:].clamp_min(1e-10))
    return out + (nns.reshape(h, w, d),) if return_nn else out


def fit_patch(s, name, p, st, warn=True):
    """utils_vid.py:307-313."""
    if (s - p) % st != 0:
        new_s = (s - p) // st * st + p
        if warn:
            warnings.warn(f"{name} doesnot satisfy ({name} - patch_size) % stride == 0. changing {name} from {s} to {new_s}")
        return new_s
    return s


def gpnn_loss(x, y, macro_block=64, patch_size=7, stride=2, patcht_size=7, stridet=2, rou=0, scaling=0.2,
              alpha=1e10, **_):
    """utils_vid.py:289-349 without the macro-block loop (which only bounds unfold memory and does not
    change the result -- verified against golden G8 that was generated WITH macro blocks).
    Returns (loss, y2x, weight)."""
    t, h, w = x.shape[-3:]
    h = fit_patch(h, "patch_height", patch_size, stride, False)
    w = fit_patch(w, "patch_width", patch_size, stride, False)
    t = fit_patch(t, "frame_num", patcht_size, stridet, False)
    x = x[..., :t, :h, :w]
    y = y[..., :h, :w]
    with torch.no_grad():
        s, wgt = find_nn_and_merge(x, y, patch_size, patcht_size, stride, stridet, alpha)
        y2x = s / wgt
    return robust_lossfun(x - y2x, rou, scaling).mean(), y2x, wgt


def gpnn_direct_loss(x, y, rou=0, scaling=0.2, patch_size=7, patcht_size=7, stride=1, stridet=1, alpha=1e10, **_):
    """utils_vid.py:265-286 Patch3DGPNNDirectLoss: NO trimming -- FindNNpatchAndMerge on x as it is (UnfoldNd floors the patch grid, FoldNd fills the
    whole x.shape: voxels no patch covers get sum 0 / weight 1e-10), the loss mean runs over all of x.  Defaults = FindNNpatchAndMerge's (utils_vid.py:206)."""
    with torch.no_grad():
        s, wgt = find_nn_and_merge(x, y, patch_size, patcht_size, stride, stridet, alpha)
        y2x = s / wgt
    return robust_lossfun(x - y2x, rou, scaling).mean(), y2x, wgt


def compute_nnerr(src, tar, patch_size=7, stride=2, patcht_size=7, stridet=2, macro_block=65):
    """evaluations/NNMSE.py:7-58: mean over macro blocks of mean |NN patch of tar - patch of src| (plain NN, alpha None)."""
    import numpy as np
    t, h, w = src.shape[-3:]
    macro_block = fit_patch(macro_block, "macro_block", patch_size, stride, False)
    h = fit_patch(h, "patch_height", patch_size, stride, False)
    w = fit_patch(w, "patch_width", patch_size, stride, False)
    t = fit_patch(t, "frame_num", patcht_size, stridet, False)
    src, tar = src[..., :t, :h, :w], tar[..., :h, :w]
    ms = macro_block - patch_size + stride
    errs = []
    for hs in np.arange(0, h - macro_block + ms, ms):
        for ws in np.arange(0, w - macro_block + ms, ms):
            sc = src[..., hs:hs + macro_block, ws:ws + macro_block]
            tc = tar[..., hs:hs + macro_block, ws:ws + macro_block]
            ps_ = extract_3Dpatches(sc, patch_size, patcht_size, stride, stridet)
            b, c, d, hh, ww = ps_.shape
            B = b * hh * ww
            X = _to_location_major(ps_, B, patcht_size, patch_size)
            Y = _to_location_major(extract_3Dpatches(tc, patch_size, patcht_size, stride, stridet), B, patcht_size, patch_size)
            nns = nn_indices(X, Y, None)
            errs.append((Y[torch.arange(B)[:, None], nns] - X).abs().mean().item())
    return float(np.array(errs).mean())
