"""Restatement of how the reference SPARSIFIES a stage-1 model and PACKS it into its checkpoint format.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference MPI.py:66-81 (vertex grid, faces, vertex UVs), MPI.py:288-442 (`sparsify_faces`: morphology on the atlas, per-quad
tiles cut with grid_sample(align_corners=True), classification by the largest tile sample, static / dynamic tiles packed row-major into
two atlases by `get_hw`, the last tile repeated into the residual slots, `gen_quad_uvs`) and MPI.py:207-221 (state_dict scalars).

Pinning (round 4): `sparsify_atlas` == the state_dict the reference's OWN `MPMesh.sparsify_faces()` produced on the same atlas (golden G15,
tests/golden/make_golden_r04.py imports MPI.py; tests/test_reference_modules_cpu.py: integer tensors and scalars exactly, float tensors to
1e-6).  `pack_reference_state` runs the same packing on a pitch-1 plane stack with given quad maps (tiles of round(quad extent)+1 texels
sit exactly on the stack's texels), for the reader / exporter tests on other shapes.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import atlas_oracle as AO


def _quad_faces(D, hv, wv):
    vid = torch.arange(D * hv * wv).reshape(D, hv, wv)                # MPI.py:66-71
    f013 = torch.stack([vid[:, :-1, :-1], vid[:, :-1, 1:], vid[:, 1:, 1:]], -1)
    f320 = torch.stack([vid[:, 1:, 1:], vid[:, 1:, :-1], vid[:, :-1, :-1]], -1)
    return torch.cat([f013.reshape(-1, 1, 3), f320.reshape(-1, 1, 3)], dim=1)     # [n_quad, 2, 3], quad order (d, vy, vx)


def atlas_grid(n, max_ratio=4):
    """MPI.py:366-377 `get_hw`: rows = the candidate in [sqrt(n/4), sqrt(n)) with the smallest (rows - n % rows), cols = n // rows + 1.
    (The reference's candidate list is empty or starts at 0 below n = 4; one row then -- no shipped model is that small.)"""
    if n == 0:
        return 0, 0, 0
    cand = np.arange(int(np.sqrt(n / max_ratio)), int(np.sqrt(n)))
    if len(cand) == 0 or cand[0] == 0:
        return 1, n + 1, 1
    h = int(cand[np.argmin(cand - n % cand)])
    w = n // h + 1
    return h, w, h * w - n


def _pack(tl):
    """tiles [frames,n,4,ih,iw] -> (atlas [frames,4,gh*ih,gw*iw], corner uvs [4n,2], uvfaces [2n,3], gh, gw)   (MPI.py:380-418)."""
    frames, n, _, ih, iw = tl.shape
    if n == 0:
        return torch.zeros(frames, 4, 0, 0), torch.zeros(0, 2), torch.zeros(0, 3, dtype=torch.long), 0, 0
    gh, gw, pad = atlas_grid(n)
    tl = torch.cat([tl, tl[:, -1:].expand(-1, pad, -1, -1, -1)], 1)
    atlas = tl.reshape(frames, gh, gw, 4, ih, iw).permute(0, 3, 1, 4, 2, 5).reshape(frames, 4, gh * ih, gw * iw)
    Ah, Aw = atlas.shape[-2:]
    k = torch.arange(n)
    # tile k at grid (k // gw, k % gw); corners on its first / last texel centre
    u0 = ((k % gw) * iw).float() / (Aw - 1) * 2 - 1
    v0 = ((k // gw) * ih).float() / (Ah - 1) * 2 - 1
    du, dv = 2 / (Aw - 1) * (iw - 1), 2 / (Ah - 1) * (ih - 1)
    uvs = torch.stack([torch.stack([u0, v0], -1), torch.stack([u0 + du, v0], -1), torch.stack([u0, v0 + dv], -1),
                       torch.stack([u0 + du, v0 + dv], -1)], 1).reshape(-1, 2).float()
    uvfaces = (k * 4)[:, None, None] + torch.tensor([[0, 1, 3], [3, 2, 0]])[None]
    return atlas, uvs, uvfaces.reshape(-1, 3), gh, gw


def _state(faces, idx_s, idx_d, packed_s, packed_d, extra):
    atlas_s, uvs_s, uvf_s, gh_s, gw_s = packed_s
    atlas_d, uvs_d, uvf_d, gh_d, gw_d = packed_d
    return {**extra,
            "faces": faces[idx_s].reshape(-1, 3), "uvfaces": uvf_s, "uvs": uvs_s, "atlas": atlas_s,
            "faces_dyn": faces[idx_d].reshape(-1, 3), "uvfaces_dyn": uvf_d, "uvs_dyn": uvs_d, "atlas_dyn": atlas_d,
            "self.is_sparse": True, "self.has_dyn": True,
            "self.atlas_grid_h": gh_s, "self.atlas_grid_w": gw_s, "self.atlas_full_h": atlas_s.shape[-2], "self.atlas_full_w": atlas_s.shape[-1],
            "self.atlas_grid_dyn_h": gh_d, "self.atlas_grid_dyn_w": gw_d, "self.atlas_full_dyn_h": atlas_d.shape[-2],
            "self.atlas_full_dyn_w": atlas_d.shape[-1]}


def _morph(img, kind):
    """3x3 max (dilate) / min (erode) filter with ZERO padding (utils.py:298-317: torch.nn.Unfold(padding=1) pads with 0)."""
    p = F.pad(img if kind == "max" else -img, (1, 1, 1, 1), value=0.0)
    out = F.max_pool2d(p, 3, stride=1)
    return out if kind == "max" else -out


def sparsify_atlas(atlas, atlas_mask, grid_h, D, hv, wv, erode_num=2, alpha_thresh=0.03, loop_thresh=0.5, rmfirstlayer=0, alpha_init=-3.0):
    """atlas [1,4,Ah,Aw] (+ loop-mask logits [1,1,Ah,Aw]) of a dense stage-1 MPMesh (sigmoid activations) -> the tensors / scalars
    `sparsify_faces()` leaves in `state_dict()` (without the camera buffers and `_verts`, which it does not touch)."""
    gw = D // grid_h
    Ah, Aw = atlas.shape[-2:]
    uvs = AO.reference_vertex_uvs(grid_h, gw, hv, wv)                               # [D, hv*wv, 2]
    uv = uvs.reshape(D, hv, wv, 2)
    ext_w = float(uv[0, 0, 1, 0] - uv[0, 0, 0, 0])
    ext_h = float(uv[0, 1, 0, 1] - uv[0, 0, 0, 1])
    iw, ih = int(np.round(ext_w / 2 * (Aw - 1))), int(np.round(ext_h / 2 * (Ah - 1)))  # MPI.py:303-307
    off = torch.stack(torch.meshgrid(torch.linspace(0, ext_w, iw), torch.linspace(0, ext_h, ih), indexing="xy"), -1)   # ih,iw,2 (u, v)
    v0 = uv[:, :-1, :-1].reshape(-1, 2)                                              # first vertex of every quad, order (d, vy, vx)
    grid = v0[:, None, None, :] + off[None]                                          # n,ih,iw,2
    n = grid.shape[0]

    def cut(img):                                                                     # img [1,C,Ah,Aw] -> [n,C,ih,iw]
        return F.grid_sample(img.expand(n, -1, -1, -1), grid.to(img.dtype), mode="bilinear", padding_mode="zeros", align_corners=True)

    a = atlas[:, 3:].detach().clone()
    a[a == alpha_init] = -10                                                         # MPI.py:318
    a = torch.sigmoid(a)
    m = atlas_mask.detach().clone()
    m[m == alpha_init] = -10
    m = torch.sigmoid(m)
    for _ in range(erode_num):
        m = _morph(m, "min")
    for _ in range(erode_num):
        m = _morph(m, "max")
    for _ in range(erode_num):
        a = _morph(a, "min")
    for _ in range(erode_num + 2):
        a = _morph(a, "max")
    ta = cut(a).reshape(n, -1)
    if rmfirstlayer > 0:
        ta[:hv * wv * rmfirstlayer] = 0                                              # MPI.py:343-346 (vertices per plane, as written)
    keep = ta.max(-1)[0] > alpha_thresh
    dyn = keep & (cut(m).reshape(n, -1).max(-1)[0] > loop_thresh)
    tl = cut(atlas.detach())                                                         # n,4,ih,iw
    idx_s, idx_d = (keep & ~dyn).nonzero()[:, 0], dyn.nonzero()[:, 0]
    return _state(_quad_faces(D, hv, wv), idx_s, idx_d, _pack(tl[idx_s][None]), _pack(tl[idx_d][None]), {})


def pack_reference_state(stack, keep, dyn, hv, wv, planedepth):
    """stack (D,T,H,W,4) at pitch 1 (static quads identical over T), keep/dyn [D,hv-1,wv-1] bool -> reference-style sparse state_dict
    with tiles of round(quad extent)+1 texels (one sample per texel of the quad)."""
    D, T, H, W, _ = stack.shape
    QH, QW = hv - 1, wv - 1
    ch, cw = (H - 1) / QH, (W - 1) / QW
    imsz_h, imsz_w = int(round(ch)) + 1, int(round(cw)) + 1

    def tiles_of(mask, frames):
        idx = mask.reshape(-1).nonzero()[:, 0]
        n = len(idx)
        if n == 0:
            return idx, (torch.zeros(frames, 4, 1, 1), torch.zeros(0, 2), torch.zeros(0, 3, dtype=torch.long), 0, 0)
        d, rem = idx // (QH * QW), idx % (QH * QW)
        vy, vx = rem // QW, rem % QW
        ys = vy[:, None].double() * ch + torch.linspace(0, ch, imsz_h, dtype=torch.float64)[None]      # n, imsz_h plane rows
        xs = vx[:, None].double() * cw + torch.linspace(0, cw, imsz_w, dtype=torch.float64)[None]
        gy = (ys / (H - 1) * 2 - 1).float()
        gx = (xs / (W - 1) * 2 - 1).float()
        grid = torch.stack([gx[:, None, :].expand(n, imsz_h, imsz_w), gy[:, :, None].expand(n, imsz_h, imsz_w)], -1)
        out = []
        for t in range(frames):
            img = stack[d, t].permute(0, 3, 1, 2)                                              # n,4,H,W
            out.append(F.grid_sample(img, grid, mode="bilinear", align_corners=True))          # n,4,ih,iw   (MPI.py:340)
        return idx, _pack(torch.stack(out, 0))

    idx_s, packed_s = tiles_of(keep & ~dyn, 1)
    idx_d, packed_d = tiles_of(dyn, T)
    return _state(_quad_faces(D, hv, wv), idx_s, idx_d, packed_s, packed_d,
                  {"planedepth": planedepth.clone(), "ref_extrin": torch.eye(4, dtype=torch.float64), "ref_intrin": torch.eye(3),
                   "_verts": torch.zeros(D * hv * wv, 3)})
