"""Export to the REFERENCE's checkpoint / asset layout: what `MPMeshVid.state_dict()`, `save_mesh` and `save_texture`
of the reference produce (MPV.py:290-341), from the dense plane stack + quad maps of this build.

  reference_state_dict(model)   -> the state_dict the reference's MPMeshVid.init_from_mpi / load_state_dict reads (MPV.py:235-304):
                                   `_verts` (vertex grid, utils_mpi.py:80-89), `faces` / `faces_dyn` (two triangles per quad, vertex
                                   order of MPV.py:66-71), `uvs*` / `uvfaces*` (4 corners per tile, MPI.py:403-418), `atlas` [1,4,..]
                                   (static tiles, stored ONCE) and `atlas_dyn` [T,4,..] (dynamic tiles), camera buffers and the
                                   python scalars under "self.*" keys.  Tiles are packed like MPI.py:364-400 (get_hw grid, row major,
                                   the last tile repeated into the residual slots).
  save_mesh(model, prefix)      -> prefix.obj / prefix_dyn.obj with the reference's OBJ writer conventions (utils.py:403-435:
                                   unused vertices culled, UVs flipped and moved to texel centres by normalize_uv).
  save_texture(model, prefix)   -> prefix_static.png (activated rgba) and prefix_dyn_%04d.png (activated rgb * alpha per frame).
                                   The reference writes the dynamic frames as one .mov through imageio/ffmpeg (MPV.py:341), which this
                                   image does not have: the same frames go out as PNG files instead.
The reader of this layout is videoloop3d_amd.tiles.stack_from_reference_state (used by MPMeshVid.init_from_mpi): export -> read
round-trips (tests/test_export_cpu.py).  Unpinned like every statement about the reference's packed format: no checkpoint ships
with the reference and its MPI.py / MPV.py cannot be imported here (oracle/ckpt_oracle.py restates the packing for the tests).
"""
import os
import struct
import zlib

import numpy as np
import torch
import torch.nn.functional as F

from .utils_mpi import gen_mpi_vertices


def atlas_grid(n, max_ratio=4):
    """MPI.py:366-377 get_hw: tile grid (rows, cols, residual slots) for n tiles."""
    if n == 0:
        return 0, 0, 0
    n_min, n_max = int(np.sqrt(n / max_ratio)), int(np.sqrt(n))
    n_try = np.arange(n_min, n_max)
    if len(n_try) == 0 or n_try[0] == 0:          # the reference's arange is empty / starts at 0 for tiny n: one row
        return 1, n + 1, 1
    h = int(n_try[np.argmin(n_try - n % n_try)])
    w = n // h + 1
    return h, w, h * w - n


def quad_faces(D, hv, wv):
    """MPV.py:66-71: [D*(hv-1)*(wv-1), 2, 3] vertex ids, quad order (d, vy, vx), triangles (v0,v1,v3), (v3,v2,v0)."""
    vid = torch.arange(D * hv * wv).reshape(D, hv, wv)
    f013 = torch.stack([vid[:, :-1, :-1], vid[:, :-1, 1:], vid[:, 1:, 1:]], -1)
    f320 = torch.stack([vid[:, 1:, 1:], vid[:, 1:, :-1], vid[:, :-1, :-1]], -1)
    return torch.cat([f013.reshape(-1, 1, 3), f320.reshape(-1, 1, 3)], dim=1)


def _pack_tiles(stack, mask, frames, tile_hw, own=False):
    """tiles of the quads in `mask` [D,QH,QW], sampled from stack (D,T,H,W,4) like MPI.py:306-340 (grid_sample, align_corners=True,
    from the quad's first to its last corner), packed row-major into an atlas [frames,4,Ah,Aw] + per-tile corner UVs / uv faces.
    own: the stack is in the TILE-EXACT layout (tile (vy, vx) = rows [vy ih, (vy+1) ih), columns [vx iw, (vx+1) iw)): its tiles are copied
    texel for texel -- the checkpoint the model was read from comes back bit for bit."""
    D, T, H, W, _ = stack.shape
    QH, QW = mask.shape[1:]
    ch, cw = (H - 1) / QH, (W - 1) / QW
    ih, iw = tile_hw
    idx = mask.reshape(-1).nonzero()[:, 0]
    n = len(idx)
    if n == 0:
        return idx, torch.zeros((frames, 4, 1, 1), device=stack.device), torch.zeros((0, 2), device=stack.device), torch.zeros((0, 3), dtype=torch.long), (0, 0)
    d, rem = idx // (QH * QW), idx % (QH * QW)
    vy, vx = rem // QW, rem % QW
    ys = vy[:, None].double() * ch + torch.linspace(0, ch, ih, dtype=torch.float64)[None]
    xs = vx[:, None].double() * cw + torch.linspace(0, cw, iw, dtype=torch.float64)[None]
    gy, gx = (ys / (H - 1) * 2 - 1).float(), (xs / (W - 1) * 2 - 1).float()
    grid = torch.stack([gx[:, None, :].expand(n, ih, iw), gy[:, :, None].expand(n, ih, iw)], -1).to(stack.device)
    # plane by plane, all frames and all of the plane's tiles in one call (the tiles' grids stacked along the rows): indexing
    # stack[d, t] with one plane index per TILE materialised a whole (H,W,4) plane per tile -- 174 GB at the shipped size
    tiles = torch.empty((frames, n, 4, ih, iw), dtype=torch.float32, device=stack.device)
    dd = d.to(stack.device)
    for plane in torch.unique(d).tolist():
        sel = (dd == plane).nonzero()[:, 0]
        if own:
            img = stack.plane(plane, frames, raw=True).float()                                  # frames,H,W,4 = frames,QH,ih,QW,iw,4
            tl = img.reshape(frames, QH, ih, QW, iw, 4)[:, vy[sel.cpu()].to(img.device), :, vx[sel.cpu()].to(img.device)]    # n_sel,frames,ih,iw,4
            tiles[:, sel] = tl.permute(1, 0, 4, 2, 3)
            continue
        g = grid[sel].reshape(1, len(sel) * ih, iw, 2).expand(frames, -1, -1, -1)
        img = stack.plane(plane, frames).permute(0, 3, 1, 2).float()                            # frames,4,H,W
        out = F.grid_sample(img, g, mode="bilinear", align_corners=True)                        # frames,4,len(sel)*ih,iw
        tiles[:, sel] = out.reshape(frames, 4, len(sel), ih, iw).permute(0, 2, 1, 3, 4)
    gh, gw, pad = atlas_grid(n)
    tiles = torch.cat([tiles, tiles[:, -1:].expand(-1, pad, -1, -1, -1)], 1)                    # MPI.py:392
    atlas = tiles.reshape(frames, gh, gw, 4, ih, iw).permute(0, 3, 1, 4, 2, 5).reshape(frames, 4, gh * ih, gw * iw)
    Ah, Aw = atlas.shape[-2:]
    k = torch.arange(n)
    u0 = (k % gw).double() * iw / (Aw - 1) * 2 - 1                                              # gen_quad_uvs, MPI.py:403-418
    v0 = (k // gw).double() * ih / (Ah - 1) * 2 - 1
    du, dv = 2 / (Aw - 1) * (iw - 1), 2 / (Ah - 1) * (ih - 1)
    uvs = torch.stack([torch.stack([u0, v0], -1), torch.stack([u0 + du, v0], -1), torch.stack([u0, v0 + dv], -1),
                       torch.stack([u0 + du, v0 + dv], -1)], 1).reshape(-1, 2).float()
    uvfaces = ((k * 4)[:, None, None] + torch.tensor([[0, 1, 3], [3, 2, 0]])[None]).reshape(-1, 3)
    return idx, atlas, uvs, uvfaces, (gh, gw)


def reference_state_dict(model, tile_texels=None):
    """model: videoloop3d_amd MPMeshVid / MPMesh (dense or sparsified).  tile_texels=(ih, iw) overrides the tile size (default:
    one sample per texel of the quad, round(quad extent) + 1, the choice oracle/ckpt_oracle.py pins the reader with)."""
    class _Planes:
        """plane accessor: stack[d, :frames] of the dense model without ever holding more than one plane of a packed one."""
        def __init__(self, m):
            self.m = m
            self.shape = tuple(m.stack_dims()) + (4,) if hasattr(m, "stack_dims") else tuple(m.stack.shape)
            self.device = (m._param() if hasattr(m, "_param") else m.stack).device

        def plane(self, d, frames, raw=False):
            pl = self.m.stack_plane(d, range(frames)) if hasattr(self.m, "stack_plane") else self.m.stack.detach()[d, :frames]
            pl = pl.detach()
            if bool(getattr(self.m, "is_sparse", False)) and not raw:
                # texels no kept quad can read hold the alpha logit CULLED_ALPHA (-1e4, tiles.py); a tile's border samples sit exactly on
                # texel centres, but their fp32 coordinates carry ~1e-6 texels of rounding, which would pull 1e-6 * (-1e4) of a culled
                # neighbour into an exported kept texel.  -30 is as transparent (sigmoid = 1e-13) and bleeds nothing.
                pl = torch.cat([pl[..., :3], pl[..., 3:].clamp_min(-30.0)], dim=-1)
            return pl
    stack = _Planes(model)
    D, T, H, W, _ = stack.shape
    hv, wv = int(model.args.mpi_h_verts), int(model.args.mpi_w_verts)
    QH, QW = hv - 1, wv - 1
    own = getattr(model, "tile_own", None) is not None
    if own:                                         # tile-exact layout: the model's tiles ARE the reference's (texel for texel)
        if tile_texels is not None and tuple(tile_texels) != tuple(model.tile_own):
            raise RuntimeError(f"a tile-exact model exports its own tiles of {model.tile_own} texels")
        tile_texels = tuple(model.tile_own)
    if tile_texels is None:
        tile_texels = (int(round((H - 1) / QH)) + 1, int(round((W - 1) / QW)) + 1)
    sparse = bool(getattr(model, "is_sparse", False)) and getattr(model, "quad_keep", None) is not None
    if sparse:
        keep, dyn = model.quad_keep.cpu().bool(), model.quad_dyn.cpu().bool()
    else:                                           # a dense model: every quad exists and is dynamic ("load static as dynamic", MPV.py:266)
        keep = torch.ones((D, QH, QW), dtype=torch.bool)
        dyn = keep.clone()
        if not hasattr(model, "frm_num"):           # ... a dense stage-1 MPI is one static atlas (MPI.py:95-117)
            dyn = torch.zeros_like(keep)
    faces = quad_faces(D, hv, wv)
    idx_s, atlas_s, uvs_s, uvf_s, (gh_s, gw_s) = _pack_tiles(stack, keep & ~dyn, 1, tile_texels, own)
    idx_d, atlas_d, uvs_d, uvf_d, (gh_d, gw_d) = _pack_tiles(stack, dyn, T, tile_texels, own)
    # atlas_full_*: the atlas size at FULL resolution -- what MPV.lod scales its tiles from (MPV.py:149-151); a model exported at a pyramid
    # level keeps the full tile size there (tile-exact models know it: tile_full), everything else is exported at the size it has
    full_hw = tuple(model.tile_full) if (own and getattr(model, "tile_full", None) is not None) else tuple(tile_texels)
    intrin_mpi = model.ref_intrin_mpi.detach().cpu().float()
    mh, mw = model.mpi_h, model.mpi_w
    verts = gen_mpi_vertices(mh, mw, intrin_mpi, hv, wv, model.planedepth.detach().cpu().float())
    if bool(getattr(model.args, "normalize_verts", False)):
        # the reference stores `_verts` divided by the plane depth under this flag and multiplies it back in its `verts` property (MPV.py:62-64)
        verts = (verts.reshape(D, -1, 3) / model.planedepth.detach().cpu().float().reshape(D, 1, 1)).reshape(verts.shape)
    if not hasattr(model, "frm_num") and not sparse:
        # a dense stage-1 MPI: the reference's MPMesh.state_dict() before sparsify_faces has no dynamic lists (MPI.py:207-221)
        return {"_verts": verts, "planedepth": model.planedepth.detach().cpu().clone(), "ref_extrin": model.ref_extrin.detach().cpu().clone(),
                "ref_intrin": model.ref_intrin.detach().cpu().clone(), "faces": faces[idx_s].reshape(-1, 3), "uvfaces": uvf_s, "uvs": uvs_s.cpu(),
                "atlas": atlas_s.cpu(), "self.is_sparse": False, "self.atlas_grid_h": gh_s, "self.atlas_grid_w": gw_s,
                "self.atlas_full_h": int(atlas_s.shape[-2]), "self.atlas_full_w": int(atlas_s.shape[-1])}
    return {
        "_verts": verts, "planedepth": model.planedepth.detach().cpu().clone(), "ref_extrin": model.ref_extrin.detach().cpu().clone(),
        "ref_intrin": model.ref_intrin.detach().cpu().clone(),
        "faces": faces[idx_s].reshape(-1, 3), "uvfaces": uvf_s, "uvs": uvs_s.cpu(), "atlas": atlas_s.cpu(),
        "faces_dyn": faces[idx_d].reshape(-1, 3), "uvfaces_dyn": uvf_d, "uvs_dyn": uvs_d.cpu(), "atlas_dyn": atlas_d.cpu(),
        "self.is_sparse": sparse, "self.has_dyn": True,
        "self.atlas_grid_h": gh_s, "self.atlas_grid_w": gw_s, "self.atlas_full_h": gh_s * full_hw[0] if gh_s else int(atlas_s.shape[-2]), "self.atlas_full_w": gw_s * full_hw[1] if gw_s else int(atlas_s.shape[-1]),
        "self.atlas_grid_dyn_h": gh_d, "self.atlas_grid_dyn_w": gw_d, "self.atlas_full_dyn_h": gh_d * full_hw[0] if gh_d else int(atlas_d.shape[-2]),
        "self.atlas_full_dyn_w": gw_d * full_hw[1] if gw_d else int(atlas_d.shape[-1]),
    }


# ---- assets -----------------------------------------------------------------------------------------------------------------
def normalize_uv(uv, h, w):
    """utils.py:403-407: flip v, [-1,1] -> [0,1], then to texel centres of an h x w texture."""
    uv = np.array(uv, dtype=np.float64, copy=True)
    uv[:, 1] = -uv[:, 1]
    uv = uv * 0.5 + 0.5
    return uv * np.array([w - 1, h - 1]) / np.array([w, h]) + 0.5 / np.array([w, h])


def _cull_unused(v, f):
    """utils.py:410-416."""
    ids = np.unique(f)
    old2new = -np.ones(len(v), dtype=np.int64)
    old2new[ids] = np.arange(len(ids))
    return v[ids], old2new[f]


def save_obj(path, verts, faces, uvs, uvfaces, rm_unused=True):
    """utils.py:419-435."""
    if rm_unused:
        verts, faces = _cull_unused(verts, faces)
        uvs, uvfaces = _cull_unused(uvs, uvfaces)
    with open(path, "w") as f:
        for p in verts:
            f.write(f"v {p[0]} {p[1]} {p[2]}\n")
        for uv in uvs:
            f.write(f"vt {uv[0]} {uv[1]}\n")
        for face, uvface in zip(faces + 1, uvfaces + 1):
            f.write(f"f {face[0]}/{uvface[0]} {face[1]}/{uvface[1]} {face[2]}/{uvface[2]}\n")
        f.write("\n")


def save_mesh(model, prefix, state=None):
    """MPV.py:306-323: the static and the dynamic mesh as OBJ files (returns the paths written)."""
    sd = reference_state_dict(model) if state is None else state
    out = []
    for faces_k, uvs_k, uvf_k, atlas_k, suffix in (("faces", "uvs", "uvfaces", "atlas", ".obj"), ("faces_dyn", "uvs_dyn", "uvfaces_dyn", "atlas_dyn", "_dyn.obj")):
        faces = sd[faces_k].numpy()
        if len(faces) == 0:
            continue
        uvs = normalize_uv(sd[uvs_k].numpy(), sd[atlas_k].shape[2], sd[atlas_k].shape[3])
        print(f"Saving to {prefix + suffix}: # v = {len(sd['_verts'])}, # f = {len(faces)}")
        save_obj(prefix + suffix, sd["_verts"].numpy(), faces, uvs, sd[uvf_k].numpy())
        out.append(prefix + suffix)
    return out


def write_png(path, img):
    """uint8 [H,W,3|4] -> PNG (zlib + CRC from the standard library; imageio is not in this image)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    assert c in (3, 4)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * c)], axis=1).tobytes()      # filter type 0 in front of every row

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6 if c == 4 else 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


@torch.no_grad()
def save_texture(model, prefix, state=None):
    """MPV.py:325-341: activated textures, static as one RGBA image, dynamic as rgb * alpha frames."""
    sd = reference_state_dict(model) if state is None else state
    out = []
    if len(sd["faces"]) > 0:
        t = sd["atlas"][0].permute(1, 2, 0)
        rgba = torch.cat([model.rgb_activate(t[..., :-1]), model.alpha_activate(t[..., -1:])], dim=-1)
        write_png(prefix + "_static.png", (rgba * 255).type(torch.uint8).numpy())
        out.append(prefix + "_static.png")
    if len(sd["faces_dyn"]) > 0:
        t = sd["atlas_dyn"].permute(0, 2, 3, 1)
        rgb = model.rgb_activate(t[..., :-1]) * model.alpha_activate(t[..., -1:])
        frames = (rgb * 255).type(torch.uint8).numpy()
        names = [f"{prefix}_dyn_{i:04d}.png" for i in range(len(frames))]
        # (zlib releases the interpreter lock: the T frames of a 2K x 4K dynamic atlas compress side by side -- 25 s -> a few for T = 50)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1, len(frames)))) as ex:
            list(ex.map(write_png, names, frames))
        out.extend(names)
    return out
