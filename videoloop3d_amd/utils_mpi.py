"""Drop-in mirror of the reference's utils_mpi.py operators that sit on the hot path.

Same names, argument meaning and return shapes as /root/reference/utils_mpi.py; the arithmetic of
warp_homography / overcompose / overcomposeNto0 runs in the HIP library (csrc/vl3d_ops.hip), the tiny
host-side geometry (make_depths, compute_homography, gen_mpi_vertices) is plain torch exactly as in the
reference (it stays on the host/device in PyTorch there too).
"""
from typing import Tuple, Union

import torch

from . import _lib as L


# ---- geometry (host side) -------------------------------------------------------------------------
def make_depths(num_plane, min_depth, max_depth):
    """utils_mpi.py:210-211: D depths uniform in disparity, far -> near."""
    return torch.reciprocal(torch.linspace(1. / max_depth, 1. / min_depth, num_plane, dtype=torch.float32))


def gen_mpi_vertices(H, W, intrin, num_vert_h, num_vert_w, planedepth):
    """utils_mpi.py:80-89: back-project a num_vert_h x num_vert_w grid of plane pixels to 3-D for each depth."""
    ys = torch.linspace(0, H - 1, num_vert_h)
    xs = torch.linspace(0, W - 1, num_vert_w)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    xy = torch.stack([gx, gy], dim=-1).reshape(1, -1, 2)
    depth = planedepth[:, None, None].type_as(xy)
    xy = (xy - intrin[None, None, :2, 2]) * depth
    xy = xy / intrin[None, None, [0, 1], [0, 1]]
    z = depth.expand_as(xy[..., :1])
    return torch.cat([xy.reshape(-1, 2), z.reshape(-1, 1)], dim=-1)


def compute_homography(src_extrin_4x4, src_intrin, tar_extrin_4x4, tar_intrin, normal, distances):
    """utils_mpi.py:240-273.  H_d = K_src (R + t n^T R/(d - n^T t)) K_tar^-1 with [R|t] = E_src E_tar^-1.

    src/tar_extrin [B,4,4], intrinsics [B,3,3], normal [B,D,3], distances [B,D] -> [B,D,3,3], mapping
    target pixels to source (plane) pixels.  Like the reference this needs B == 1 (or B == D)."""
    pose = src_extrin_4x4 @ torch.inverse(tar_extrin_4x4)
    rotation, translation = pose[..., :3, :3], pose[..., :3, 3]
    distances_tar = distances - (normal @ translation.unsqueeze(-1)).squeeze(-1)
    plane_term = translation.unsqueeze(-1) @ normal.unsqueeze(-2) @ rotation.unsqueeze(-3)
    homo = rotation.unsqueeze(-3) + plane_term / distances_tar.unsqueeze(-1).unsqueeze(-1)
    return src_intrin.unsqueeze(-3) @ homo @ torch.inverse(tar_intrin.unsqueeze(-3))


def plane_homographies_host(src_intrin, planedepth, tar_extrin, tar_intrin):
    """compute_homography for the case the modules need every iteration -- source camera at the identity, plane normal (0, 0, 1), poses on
    the HOST -- in numpy: H_d = K_src (R + t R[2] / (d - t_z)) K_tar^-1 with [R|t] = E_tar^-1 (the same operations in the same order, in the
    poses' dtype; (t n^T) R = t R[2] exactly).  The torch spelling is ~30 CPU operators of a few microseconds each, in iterations that are
    bound by the host (stage 1: 0.9 ms of GPU time in 1.24 ms).  numpy arrays: src_intrin [3,3], planedepth [D], tar_extrin [4,4],
    tar_intrin [3,3] -> float32 tensor [D,3,3]."""
    import numpy as np
    dt = tar_extrin.dtype
    pose = np.linalg.inv(tar_extrin)
    R, t = pose[:3, :3], pose[:3, 3]
    dist = planedepth.astype(dt) - t[2]
    homo = R[None] + (t[:, None] * R[2][None, :])[None] / dist[:, None, None]
    H = src_intrin.astype(dt)[None] @ homo @ np.linalg.inv(tar_intrin.astype(dt))[None]
    return torch.from_numpy(np.ascontiguousarray(H, dtype=np.float32))


# ---- warp_homography ------------------------------------------------------------------------------
class _Warp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, homos, images, h, w):
        L.check_cuda(homos, images)
        B, D, C, Hs, Ws = images.shape
        hm = homos.detach().to(torch.float32).reshape(B * D, 3, 3).contiguous()
        img = images.to(torch.float32).contiguous()
        out = torch.empty((B, D, C, h, w), dtype=torch.float32, device=images.device)
        with torch.cuda.device(images.device):
            L.check(L.lib().vl3d_warp_fwd(B * D, C, Hs, Ws, h, w, L.ptr(hm), L.ptr(img), L.ptr(out),
                                          L.stream_ptr(images.device)), "vl3d_warp_fwd")
        ctx.save_for_backward(hm)
        ctx.dims = (B, D, C, Hs, Ws, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        (hm,) = ctx.saved_tensors
        B, D, C, Hs, Ws, h, w = ctx.dims
        g = g.contiguous()
        gi = torch.empty((B, D, C, Hs, Ws), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            L.check(L.lib().vl3d_warp_bwd(B * D, C, Hs, Ws, h, w, L.ptr(hm), L.ptr(g), L.ptr(gi),
                                          L.stream_ptr(g.device)), "vl3d_warp_bwd")
        return None, gi, None, None


def warp_homography(h, w, homos: torch.Tensor, images: torch.Tensor) -> torch.Tensor:
    """utils_mpi.py:159-176: homos [B,D,3,3], images [B,D,C,Hs,Ws] -> [B,D,C,h,w]
    (bilinear, zeros padding, texel = p*(size-1)/size).  Differentiable w.r.t. `images` only, like the
    geometry-free use in the reference (utils_mpi.py:292, MPV.py:354)."""
    return _Warp.apply(homos, images, int(h), int(w))


# ---- overcompose ----------------------------------------------------------------------------------
class _Overcompose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, content):
        L.check_cuda(alpha, content)
        D = alpha.shape[-1]
        C = content.shape[-1]
        a = alpha.to(torch.float32).contiguous()
        c = content.to(torch.float32).contiguous()
        P = a.numel() // D
        rgb = torch.empty(alpha.shape[:-1] + (C,), dtype=torch.float32, device=a.device)
        bw = torch.empty_like(a)
        with torch.cuda.device(a.device):
            L.check(L.lib().vl3d_overcompose_fwd(P, D, C, L.ptr(a), L.ptr(c), L.ptr(rgb), L.ptr(bw),
                                                 L.stream_ptr(a.device)), "vl3d_overcompose_fwd")
        ctx.save_for_backward(a, c)
        return rgb, bw

    @staticmethod
    def backward(ctx, g_rgb, g_bw):
        a, c = ctx.saved_tensors
        D, C = a.shape[-1], c.shape[-1]
        P = a.numel() // D
        g_rgb = g_rgb.contiguous() if g_rgb is not None else torch.zeros(a.shape[:-1] + (C,), device=a.device)
        g_bw = g_bw.contiguous() if g_bw is not None else None
        ga, gc = torch.empty_like(a), torch.empty_like(c)
        with torch.cuda.device(a.device):
            L.check(L.lib().vl3d_overcompose_bwd(P, D, C, L.ptr(a), L.ptr(c), L.ptr(g_rgb), L.ptr(g_bw),
                                                 L.ptr(ga), L.ptr(gc), L.stream_ptr(a.device)), "vl3d_overcompose_bwd")
        return ga, gc


def overcompose(alpha, content):
    """utils_mpi.py:92-107: front = index 0.  alpha [B,H,W,D], content [B,H,W,D,C] -> (rgb [B,H,W,C], blendweight [B,H,W,D])."""
    if content.shape[:-1] != alpha.shape:
        raise RuntimeError(f"overcompose: content {tuple(content.shape)} does not match alpha {tuple(alpha.shape)}")
    return _Overcompose.apply(alpha, content)


class _OvercomposeNto0(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mpi, blend_content):
        L.check_cuda(mpi, blend_content)
        B, D, _, H, W = mpi.shape
        m = mpi.to(torch.float32).contiguous()
        HW = H * W
        alpha = m[:, :, -1]
        content = m[:, :, :3] if blend_content is None else blend_content.to(torch.float32).contiguous()
        C = content.shape[2]
        rgb = torch.empty((B, C, H, W), dtype=torch.float32, device=m.device)
        trans = torch.empty((B, D, H, W), dtype=torch.float32, device=m.device)
        with torch.cuda.device(m.device):
            L.check(L.lib().vl3d_overcompose_nto0_fwd(
                B, D, C, HW, L.ptr(alpha), alpha.stride(0), alpha.stride(1),
                L.ptr(content), content.stride(0), content.stride(1), content.stride(2),
                L.ptr(rgb), L.ptr(trans), L.stream_ptr(m.device)), "vl3d_overcompose_nto0_fwd")
        ctx.save_for_backward(m, content if blend_content is not None else None)
        ctx.mark_non_differentiable(trans)
        return rgb, trans

    @staticmethod
    def backward(ctx, g_rgb, _g_trans):
        m, bc = ctx.saved_tensors
        B, D, _, H, W = m.shape
        alpha = m[:, :, -1]
        content = m[:, :, :3] if bc is None else bc
        C = content.shape[2]
        g_rgb = g_rgb.contiguous()
        ga = torch.empty((B, D, H, W), dtype=torch.float32, device=m.device)
        gc = torch.empty((B, D, C, H, W), dtype=torch.float32, device=m.device)
        with torch.cuda.device(m.device):
            L.check(L.lib().vl3d_overcompose_nto0_bwd(
                B, D, C, H * W, L.ptr(alpha), alpha.stride(0), alpha.stride(1),
                L.ptr(content), content.stride(0), content.stride(1), content.stride(2),
                L.ptr(g_rgb), L.ptr(ga), L.ptr(gc), L.stream_ptr(m.device)), "vl3d_overcompose_nto0_bwd")
        gm = torch.zeros_like(m)
        gm[:, :, -1] = ga
        if bc is None:
            gm[:, :, :3] = gc
            return gm, None
        return gm, gc


def overcomposeNto0(mpi: torch.Tensor, blendweight=None, ret_mask=False, blend_content=None) \
        -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
    """utils_mpi.py:110-132: front = LAST plane index.  mpi [B,D,4,H,W] -> rgb [B,3,H,W]
    (+ the transmittance [B,D,H,W] the reference calls `blendweight` when ret_mask=True; it is returned
    detached).  A caller-supplied `blendweight` short-circuits the scan exactly as in the reference."""
    if blendweight is not None:
        renderw = mpi[:, :, -1] * blendweight
        content = mpi[:, :, :3] if blend_content is None else blend_content
        rgb = (content * renderw.unsqueeze(2)).sum(dim=1)
        return (rgb, blendweight) if ret_mask else rgb
    rgb, trans = _OvercomposeNto0.apply(mpi, blend_content)
    return (rgb, trans) if ret_mask else rgb
