"""Exact atlas-cell sampling of the reference's MPMeshVid on the dense plane stack.

The reference keeps the D planes as a grid of grid_h x grid_w cells in ONE texture atlas (T,4,Ah,Aw), Ah = grid_h*mpi_h,
Aw = grid_w*mpi_w (MPV.py:37-44), and addresses it with normalised UVs (MPV.py:75-81): plane p <-> cell (i, j) = (p // grid_w,
p % grid_w), cell origin (j/gw*2-1, i/gh*2-1), cell extent 2/gw x 2/gh, sampled with grid_sample(align_corners=True)
(MPV.py:425-427).  A plane pixel (xm, ym), 0 <= xm <= mpi_w-1, therefore reads atlas texel
    ax = (j + xm/(mpi_w-1)) * (Aw-1)/gw         (ay alike)
i.e. in the plane's own texel frame  local_x = xm * pitch_x - j/gw,  pitch_x = (Aw-1)/(gw*(mpi_w-1))  (1.00124 at mpi_w = 704,
gw = 8): the pitch is not 1, every cell column has its own sub-texel origin, and at a cell edge the bilinear taps reach the first /
last texel column of the NEIGHBOURING cell (local_x in (-1, 0) for j > 0, in (mpi_w-1, mpi_w) for j < gw-1).

Here the same sampling runs on the dense (D,T,mpi_h,mpi_w,4) stack through the render kernels' per-plane convention
(VL3D_COORD_AFFINE_PLANES, include/vl3d.h): `plane_records` folds every plane's affine texel transform into its homography and
gives it its own hard-cut box, and `stack_with_aprons` adds a one-texel apron around every plane that holds the neighbouring
cells' edge texels -- a differentiable copy, so the aprons' gradient flows back to the texels they mirror.  Identical MPV weights
(`MPV.atlas_to_stack(atlas_dyn)`) then render the reference's image (tests/test_gpu_atlas.py vs oracle/atlas_oracle.py, <= 1e-4;
unpinned at the pytorch3d boundary like the rest of the MPV convention).  It is a parity mode: the apron copy costs a pass over
the stack per call; training at speed uses pitch 1 (`MPMeshVid(texel_scale=...)` for a global pitch).
"""
import torch

from .render import RenderSpec, render_planes


def cell_of(p, grid_w):
    return p // grid_w, p % grid_w


def plane_records(homos, grid_h, mpi_h, mpi_w, apron=1):
    """homos [D,3,3] (target pixel -> plane pixel (xm, ym)) -> [D,16] float32 records of the per-plane convention for the
    apron-padded stack: matrix A_p @ H_p with A_p = the cell's affine texel transform (+apron), then the coverage box."""
    D = homos.shape[0]
    assert D % grid_h == 0, "mpi_d and atlas_grid_h should match (MPV.py:38)"
    grid_w = D // grid_h
    Ah, Aw = grid_h * mpi_h, grid_w * mpi_w
    px = (Aw - 1) / (grid_w * (mpi_w - 1))
    py = (Ah - 1) / (grid_h * (mpi_h - 1))
    Hd = homos.detach().double().cpu()
    rec = torch.zeros((D, 16), dtype=torch.float64)
    for p in range(D):
        i, j = cell_of(p, grid_w)
        ox, oy = -j / grid_w + apron, -i / grid_h + apron
        A = torch.tensor([[px, 0.0, ox], [0.0, py, oy], [0.0, 0.0, 1.0]], dtype=torch.float64)
        rec[p, :9] = (A @ Hd[p]).reshape(9)
        rec[p, 9], rec[p, 10] = ox, ox + px * (mpi_w - 1)        # 0 <= xm <= mpi_w - 1  (hard cut, MPV.py:389 / utils_mpi.py:81-82)
        rec[p, 11], rec[p, 12] = oy, oy + py * (mpi_h - 1)
    return rec.float().to(homos.device)


def stack_with_aprons(stack, grid_h):
    """(D,T,mh,mw,4) -> (D,T,mh+2,mw+2,4): every plane framed by the edge texels of its atlas neighbours (zeros where the atlas
    ends).  Built from differentiable copies: the gradient of an apron texel is added to the texel it mirrors."""
    D, T, mh, mw, C = stack.shape
    gw = D // grid_h
    cells = stack.reshape(grid_h, gw, T, mh, mw, C)
    out = stack.new_zeros((grid_h, gw, T, mh + 2, mw + 2, C))
    out[:, :, :, 1:-1, 1:-1] = cells
    out[:, :-1, :, 1:-1, -1] = cells[:, 1:, :, :, 0]           # right apron  = first column of the cell to the right
    out[:, 1:, :, 1:-1, 0] = cells[:, :-1, :, :, -1]           # left apron   = last column of the cell to the left
    out[:-1, :, :, -1, 1:-1] = cells[1:, :, :, 0, :]           # bottom apron = first row of the cell below
    out[1:, :, :, 0, 1:-1] = cells[:-1, :, :, -1, :]           # top apron    = last row of the cell above
    out[:-1, :-1, :, -1, -1] = cells[1:, 1:, :, 0, 0]          # corners: the diagonal neighbours' corner texels
    out[:-1, 1:, :, -1, 0] = cells[1:, :-1, :, 0, -1]
    out[1:, :-1, :, 0, -1] = cells[:-1, 1:, :, -1, 0]
    out[1:, 1:, :, 0, 0] = cells[:-1, :-1, :, -1, -1]
    return out.reshape(D, T, mh + 2, mw + 2, C)


def render_atlas_exact(stack, homos, H, W, grid_h, pixel_center=0.5, rgb_act="sigmoid", alpha_act="sigmoid"):
    """The reference's render of MPV weights laid out as (D,T,mpi_h,mpi_w,4) = MPV.atlas_to_stack(atlas_dyn): rgb [T,H,W,3],
    alpha [T,H,W], differentiable w.r.t. `stack`.  homos [D,3,3]: target pixel -> plane pixel (MPMeshVid.plane_homographies)."""
    if (rgb_act, alpha_act) != ("sigmoid", "sigmoid"):
        raise RuntimeError("the per-plane convention is built for the shipped (sigmoid, sigmoid) activations (configs/mpv_base.txt:29-30)")
    D, T, mh, mw, _ = stack.shape
    rec = plane_records(homos, grid_h, mh, mw)
    spec = RenderSpec(pixel_center=float(pixel_center), coord_mode="affine_planes", border="hardcut", act_order="post",
                      rgb_act=rgb_act, alpha_act=alpha_act)
    return render_planes(stack_with_aprons(stack, grid_h), rec, H, W, spec)
