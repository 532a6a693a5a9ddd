"""CPU oracle for the reference's atlas-cell sampling (MPMeshVid, planar geometry).  TEST INFRASTRUCTURE ONLY.

Restates, in plain PyTorch on the (T,4,Ah,Aw) ATLAS itself, what MPV.py does with it for un-deformed planes:
  MPV.py:37-44    atlas of grid_h x grid_w cells, plane p <-> cell (p // grid_w, p % grid_w)
  MPV.py:75-81    per-vertex UVs: cell origin (j/gw*2-1, i/gh*2-1) + linspace(0,1,verts) * (2/gw, 2/gh)
  MPV.py:394-405  barycentric interpolation of the vertex UVs inside a face -- affine in the plane coordinates, so for the planar
                  quads of a plane: uv = origin + (xm/(mpi_w-1), ym/(mpi_h-1)) * cell size
  MPV.py:425-427  grid_sample(atlas_dyn[ts], uv, bilinear, padding zeros, align_corners=True)
  MPV.py:435      sigmoid activations of the sampled rgba  (feat2rgba = first 4 channels, :108)
  MPV.py:441-453  zero canvas + masked_scatter (uncovered -> 0 after activation), overcompose front-first (utils_mpi.py:92-107)
The plane coordinates (xm, ym) of a target pixel come from the same homography the other oracles use (ray through the pixel centre
x + 0.5, ref_intrin_mpi), coverage is the quad extent 0 <= xm <= mpi_w-1, 0 <= ym <= mpi_h-1.

Pinning (round 4): goldens G14 / G17 (tests/golden/make_golden_r04.py) hold what the reference's OWN MPI.py / MPV.py produce -- the
constructor's `uvs` tensors (G14 == reference_vertex_uvs) and full renders / forwards of dense models through a harness-side analytic
rasteriser (G17 == render_atlas to 1e-6, tests/test_reference_modules_cpu.py).  Still unpinned, by necessity: the un-vendored
rasteriser's pixel-centre (+0.5) and edge rule, which the harness takes from this file's reading of pytorch3d.
Only tests/ import this module.
"""
import torch
import torch.nn.functional as F

from . import mpi_oracle as MO


def reference_vertex_uvs(grid_h, grid_w, h_verts, w_verts):
    """MPV.py:75-81 as written there (meshgrid default 'ij' indexing) -> uvs [grid_h*grid_w, h_verts*w_verts, 2]."""
    uvs_plane = torch.meshgrid([torch.arange(grid_h) / grid_h, torch.arange(grid_w) / grid_w], indexing="ij")
    uvs_plane = torch.stack(uvs_plane[::-1], dim=-1) * 2 - 1
    uvs_voxel_size = (-uvs_plane[-1, -1] + 1).reshape(1, 1, 2)
    uvs_voxel = torch.meshgrid([torch.linspace(0, 1, h_verts), torch.linspace(0, 1, w_verts)], indexing="ij")
    uvs_voxel = torch.stack(uvs_voxel[::-1], dim=-1).reshape(1, -1, 2) * uvs_voxel_size
    return uvs_plane.reshape(-1, 1, 2) + uvs_voxel.reshape(1, -1, 2)


def plane_uv(xm, ym, p, grid_h, grid_w, mpi_h, mpi_w):
    """closed form of the barycentric UV of plane pixel (xm, ym) on plane p (affine interpolation of reference_vertex_uvs)."""
    i, j = p // grid_w, p % grid_w
    u = (j / grid_w) * 2 - 1 + xm / (mpi_w - 1) * (2 / grid_w)
    v = (i / grid_h) * 2 - 1 + ym / (mpi_h - 1) * (2 / grid_h)
    return u, v


def sample_atlas_layers(atlas, homos, H, W, grid_h, mpi_h, mpi_w, pixel_center=0.5, acts=(torch.sigmoid, torch.sigmoid)):
    """atlas (T,C,Ah,Aw) pre-activation (C = 4: rgba; C = 1: the loop-mask texture, MPI.py:568-571), homos [D,3,3] target pixel -> plane
    pixel -> (plane-indexed activated layers [T,H,W,D,C], zero where uncovered; covered [H,W,D] bool)."""
    T, C = atlas.shape[:2]
    D = homos.shape[0]
    grid_w = D // grid_h
    xs, ys = MO._homography_source_coords(H, W, homos, pixel_center)          # [D,H,W] plane pixels
    layers, covs = [], []
    for p in range(D):
        xm, ym = xs[p], ys[p]
        u, v = plane_uv(xm, ym, p, grid_h, grid_w, mpi_h, mpi_w)
        grid = torch.stack([u, v], dim=-1)[None].expand(T, -1, -1, -1).to(atlas.dtype)
        samp = F.grid_sample(atlas, grid, mode="bilinear", padding_mode="zeros", align_corners=True)      # T,C,H,W
        act = torch.cat([acts[0](samp[:, :-1]), acts[1](samp[:, -1:])], 1) if C > 1 else acts[0](samp)
        cov = (xm >= 0) & (xm <= mpi_w - 1) & (ym >= 0) & (ym <= mpi_h - 1)
        layers.append(act * cov[None, None].to(atlas.dtype))
        covs.append(cov)
    return torch.stack(layers, dim=-1).permute(0, 2, 3, 4, 1), torch.stack(covs, dim=-1)


def render_atlas(atlas, homos, H, W, grid_h, mpi_h, mpi_w, pixel_center=0.5):
    """atlas (T,4,Ah,Aw) pre-activation, homos [D,3,3] target pixel -> plane pixel -> rgb [T,H,W,3], alpha [T,H,W], layers."""
    layers, _ = sample_atlas_layers(atlas, homos, H, W, grid_h, mpi_h, mpi_w, pixel_center)       # T,H,W,D,4
    rgb, bw = MO.overcompose(layers[..., 3], layers[..., :3])
    return rgb, bw.sum(-1), layers
