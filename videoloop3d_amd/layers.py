"""The materialised per-layer tensors of the reference's render(), on request (slow path).

The fused HIP render never builds the reference's `mpi` [T,H,W,K,4] layer tensor (MPV.py:441-449, MPI.py:538-548): the shipped
regularisers (rgb_smooth / a_smooth / sparsity / density) are fused into the kernels.  The terms NO shipped configuration switches on --
`d_smooth` (MPV.py:463-466, 539-551; MPI.py:558-566, 622-637: needs the rasteriser's depth buffer), `l_smooth` (MPI.py:639-645: the
loop-mask layers), `normalize_blendweight_fordepth` -- and `variables['mpi' | 'blend_weight' | 'disp_norm' | 'loopmask3d']` read it, so
it is built here from the unfused operators (warp_homography on the HIP kernel + torch), differentiable, for those callers only.
"""
import torch

from .utils_mpi import warp_homography


def plane_texel_coords(homos, H, W, spec, dev):
    """texel coordinates (tx, ty) [D,H,W] of every output pixel's sample on every plane + the plane-pixel coordinates (xm, ym)."""
    c = spec.pixel_center
    y, x = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32) + c, torch.arange(W, device=dev, dtype=torch.float32) + c, indexing="ij")
    p = homos.to(dev)[:, None, None] @ torch.stack([x, y, torch.ones_like(x)], -1)[None, ..., None]   # D,H,W,3,1
    xm, ym = p[..., 0, 0] / p[..., 2, 0], p[..., 1, 0] / p[..., 2, 0]
    return xm * spec.scale[0] + spec.offset[0], ym * spec.scale[1] + spec.offset[1], xm, ym


def coverage(tx, ty, Hs, Ws, quad_keep=None, cull_window=None):
    """hard-cut coverage of the plane quads (MPV.py:389) [D,H,W] float; a sample inside a culled quad is not covered (no face there)."""
    cov = ((tx >= 0) & (tx <= Ws - 1) & (ty >= 0) & (ty <= Hs - 1)).to(torch.float32)
    if quad_keep is not None:
        D = tx.shape[0]
        QH, QW = quad_keep.shape[1:]
        y0, x0, Hp, Wp = (0, 0, Hs, Ws) if cull_window is None else cull_window
        qx = torch.floor((tx + x0) * (QW / max(Wp - 1, 1))).clamp(0, QW - 1).long()
        qy = torch.floor((ty + y0) * (QH / max(Hp - 1, 1))).clamp(0, QH - 1).long()
        cov = cov * quad_keep.to(tx.device)[torch.arange(D, device=tx.device)[:, None, None], qy, qx].to(cov.dtype)
    return cov


def sample_planes(tex, homos, H, W, spec):
    """tex (D,T,Hs,Ws,C) -> bilinear samples [T,D,C,H,W] at the spec's sampling positions through the unfused warp kernel.
    warp_homography samples at texel = p*(S-1)/S from integer pixels, so the MPV convention (pixel centre c, texel = p*s + o) is folded
    into the homography:  H' = diag(Ws/(Ws-1), Hs/(Hs-1), 1) * A * H * shift(c)."""
    D, T, Hs, Ws, _ = tex.shape
    dev = tex.device
    c = spec.pixel_center
    (sx, sy), (ox, oy) = spec.scale, spec.offset
    A = torch.tensor([[sx * Ws / (Ws - 1), 0, ox * Ws / (Ws - 1)], [0, sy * Hs / (Hs - 1), oy * Hs / (Hs - 1)], [0, 0, 1.]], device=dev)
    C = torch.tensor([[1., 0, c], [0, 1., c], [0, 0, 1.]], device=dev)
    hm = A @ homos.to(dev) @ C
    return warp_homography(H, W, hm[None].expand(T, D, 3, 3), tex.permute(1, 0, 4, 2, 3))


def to_slots(x, cov):
    """plane-indexed [T,H,W,D,C] (0 where uncovered) -> the reference's HIT-SLOT order [T,H,W,K,C]: slot k of a pixel = its k-th nearest
    covered plane (masked_scatter over the z-sorted pix_to_face, MPV.py:441-449), K = the deepest pixel (utils.py:64-69).  cov [D,H,W]."""
    T, H, W, D, C = x.shape
    covp = cov.permute(1, 2, 0) > 0                                                      # H,W,D
    K = max(int(covp.sum(-1).max()), 1)
    slot = (torch.cumsum(covp.long(), -1) - 1).clamp(min=0)[None, ..., None].expand(T, H, W, D, C)
    out = torch.zeros((T, H, W, max(K, D), C), dtype=x.dtype, device=x.device).scatter_add(3, slot, x)
    return out[:, :, :, :K]


def inverse_depth(xm, ym, intrin_mpi, planedepth, extrin):
    """1 / zbuf of the planar mesh under every pixel, [H,W,D] (MPV.py:385): the plane point under the pixel, P_ref = depth * K_mpi^-1
    (xm, ym, 1), moved into the target camera: z = R[2] . P_ref + t[2].  extrin [4,4] ref -> target."""
    dev = xm.device
    Ki = torch.inverse(intrin_mpi.to(dev).double())
    E = extrin.to(dev).double()
    ray = Ki @ torch.stack([xm, ym, torch.ones_like(xm)], -1)[..., None].double()
    P = ray[..., 0] * planedepth.to(dev).double()[:, None, None, None]                   # D,H,W,3
    z = (P * E[2, :3]).sum(-1) + E[2, 3]
    return (1.0 / z).float().permute(1, 2, 0)


def materialise(stack, homos, H, W, spec, rgb_activate, alpha_activate, quad_keep=None, cull_window=None):
    """-> (slot-ordered [T,H,W,K,4] = the reference's `mpi`, plane-indexed [T,H,W,D,4], cov [D,H,W], (tx, ty, xm, ym))."""
    Hs, Ws = stack.shape[2:4]
    samp = sample_planes(stack, homos, H, W, spec)                                       # T,D,4,H,W
    tx, ty, xm, ym = plane_texel_coords(homos, H, W, spec, stack.device)
    cov = coverage(tx, ty, Hs, Ws, quad_keep, cull_window)
    rgba = torch.cat([rgb_activate(samp[:, :, :3]), alpha_activate(samp[:, :, 3:])], dim=2)
    rgba = (rgba * cov[None, :, None]).permute(0, 3, 4, 1, 2)                            # T,H,W,D,4, plane-indexed
    return to_slots(rgba, cov), rgba, cov, (tx, ty, xm, ym)
