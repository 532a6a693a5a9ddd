"""Counter-based synthetic inputs (bit-identical on CPU and GPU, no RNG stream).

SURVEY.md §8(d): the plane stack / videos of the benchmark are generated from an
integer hash of the flat element index so that the CPU oracle, the golden
fixtures and the GPU path all see exactly the same bytes without shipping data.
Only int64 torch ops are used, so the result does not depend on the device.
"""
import math

import torch

_M32 = 0xFFFFFFFF


def hash32(idx: torch.Tensor, seed: int) -> torch.Tensor:
    """lowbias32-style integer mix of a (possibly > 2^32) int64 counter."""
    lo = idx & _M32
    hi = (idx >> 32) & _M32
    x = (lo ^ ((hi * 0x9E3779B1) & _M32) ^ (seed & _M32)) & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def hash_uniform(shape, seed: int, device="cpu", offset: int = 0, chunk: int = 1 << 24) -> torch.Tensor:
    """float32 tensor of `shape`, u in [0,1) with 24 random bits, element i = f(i + offset)."""
    n = int(math.prod(shape))
    out = torch.empty(n, dtype=torch.float32, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        idx = torch.arange(s + offset, e + offset, dtype=torch.int64, device=device)
        out[s:e] = (hash32(idx, seed) >> 8).to(torch.float32) * (1.0 / (1 << 24))
    return out.reshape(shape)


def make_plane_stack(D, T, Hs, Ws, seed=2, device="cpu", alpha_bias=-2.0, dtype=torch.float32):
    """Pre-activation plane stack S[d,t,y,x,c], layout (D,T,Hs,Ws,4).

    values = 4u-2, alpha channel additionally biased by -2 (mimics MPV.py:109-110 init).
    Generated plane by plane so the int64 temporaries stay small at 23.6 GB scale.
    """
    out = torch.empty((D, T, Hs, Ws, 4), dtype=dtype, device=device)
    per_plane = T * Hs * Ws * 4
    for d in range(D):
        u = hash_uniform((T, Hs, Ws, 4), seed, device=device, offset=d * per_plane)
        u = u * 4.0 - 2.0
        u[..., 3] += alpha_bias
        out[d] = u.to(dtype)
    return out


def make_cameras(H, W, device="cpu", dtype=torch.float32):
    """Benchmark cameras of SURVEY.md §8(d): K=[[.9W,0,W/2],[0,.9W,H/2],[0,0,1]], ref extrinsic = I,
    target = translation (0.03,0.01,0), 0.5 degree rotation about y (near=1, far=100)."""
    K = torch.tensor([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], dtype=torch.float64)
    a = math.radians(0.5)
    R = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]],
                     dtype=torch.float64)
    E = torch.eye(4, dtype=torch.float64)
    E[:3, :3] = R
    E[:3, 3] = torch.tensor([0.03, 0.01, 0.0], dtype=torch.float64)
    ref = torch.eye(4, dtype=torch.float64)
    return (ref.to(dtype).to(device), K.to(dtype).to(device), E.to(dtype).to(device), K.to(dtype).to(device))


def make_video(T, H, W, seed, device="cpu"):
    """[1,3,T,H,W] hash-uniform video in [0,1)."""
    return hash_uniform((1, 3, T, H, W), seed, device=device)
