"""Export to the reference's checkpoint / asset layout (videoloop3d_amd/export.py; MPV.py:290-341, MPI.py:364-436) and back through
the product's reader of that layout (tiles.stack_from_reference_state): pack -> init_from_mpi -> same stack and quad maps; the
packer agrees with the restated reference packing of oracle/ckpt_oracle.py; OBJ / PNG assets are well formed.  CPU only."""
import struct
import types
import zlib

import numpy as np
import torch

from oracle import ckpt_oracle as CO
from videoloop3d_amd import export as EX
from videoloop3d_amd import tiles


def _args(**kw):
    a = dict(mpi_h_scale=1.0, mpi_w_scale=1.0, mpi_d=3, rgb_mlp_type="direct", rgb_activate="sigmoid", alpha_activate="sigmoid",
             bg_color="", learn_loop_mask=True, mpi_h_verts=5, mpi_w_verts=7, sparsify_rmfirstlayer=0, atlas_grid_h=1,
             mpv_frm_num=3, mpv_isloop=True, init_std=0.5, scale_invariant=True, fp16=False,
             swd_patch_size=3, swd_patcht_size=3, swd_stride=2, swd_stridet=1)
    a.update(kw)
    return types.SimpleNamespace(**a)


def _sparse_model(H=41, W=61):
    """(H-1) % QH == (W-1) % QW == 0: quads are 10 x 10 texels, so tiles sample the texels themselves (lossless round trip)."""
    from videoloop3d_amd.MPV import MPMeshVid
    K = np.array([[50., 0, 30], [0, 50., 20], [0, 0, 1]])
    torch.manual_seed(2)
    m = MPMeshVid(_args(), H, W, np.eye(4), K, 1.0, 100.0)
    keep = torch.rand(3, 4, 6) < 0.6
    keep[1] = False
    dyn = keep & (torch.rand(3, 4, 6) < 0.5)
    def closed(mask):                                   # closed plane-pixel rectangles of the quads of a map
        out = torch.zeros((3, H, W), dtype=torch.bool)
        for d, qy, qx in mask.nonzero().tolist():
            out[d, qy * 10:qy * 10 + 11, qx * 10:qx * 10 + 11] = True
        return out
    with torch.no_grad():
        m.stack.uniform_(-2.0, 2.0)
        # the reference's layout is per QUAD: a static quad's tile is one texture for all frames, a texel on the border line it
        # shares with a dynamic quad lives in both tiles.  (The dense model classifies per TEXEL and leaves one more texel row next
        # to a dynamic quad free per frame, tiles.tie_static_grad; an export keeps frame 0 of those.)
        static_t = (closed(keep & ~dyn) & ~closed(dyn))[:, None, :, :, None]
        m.stack.data = torch.where(static_t, m.stack.data[:, :1], m.stack.data)
        tiles.cull_stack_(m.stack.data, keep)
    m.register_buffer("quad_keep", keep)
    m.register_buffer("quad_dyn", dyn)
    m.is_sparse = m.has_dyn = True
    return m, keep, dyn


def test_export_then_read_roundtrip():
    from videoloop3d_amd.MPV import MPMeshVid
    m, keep, dyn = _sparse_model()
    sd = m.reference_state_dict()
    # the layout the reference's init_from_mpi consumes (MPV.py:235-265)
    for k in ("_verts", "ref_extrin", "ref_intrin", "planedepth", "uvs", "atlas", "uvfaces", "faces", "uvs_dyn", "atlas_dyn", "uvfaces_dyn",
              "faces_dyn", "self.is_sparse", "self.has_dyn", "self.atlas_full_w", "self.atlas_full_h", "self.atlas_grid_h", "self.atlas_grid_w",
              "self.atlas_full_dyn_w", "self.atlas_full_dyn_h", "self.atlas_grid_dyn_h", "self.atlas_grid_dyn_w"):
        assert k in sd, k
    n_s, n_d = int((keep & ~dyn).sum()), int(dyn.sum())
    assert sd["faces"].shape == (2 * n_s, 3) and sd["faces_dyn"].shape == (2 * n_d, 3) and sd["uvs"].shape == (4 * n_s, 2)
    assert sd["atlas"].shape[0] == 1 and sd["atlas_dyn"].shape[0] == 3                      # static tiles are stored ONCE
    assert sd["atlas"].shape[-2] == sd["self.atlas_grid_h"] * 11 and sd["atlas_dyn"].shape[-1] == sd["self.atlas_grid_dyn_w"] * 11
    assert sd["_verts"].shape == (3 * 5 * 7, 3)
    # back through the reader: the same quad maps and, on every texel a kept quad can read, the same values
    b = MPMeshVid(_args(), 41, 61, np.eye(4), np.array([[50., 0, 30], [0, 50., 20], [0, 0, 1]]), 1.0, 100.0)
    b.init_from_mpi(sd, tile_layout="lattice")      # (a pitch-1 model's export, read back onto the stack it came from)
    assert b.is_sparse and torch.equal(b.quad_keep, keep) and torch.equal(b.quad_dyn, dyn)
    inside = tiles.quad_to_texel_mask(keep, 41, 61)
    closed = torch.zeros_like(inside)                                                        # closed rectangles of the kept quads
    for d, qy, qx in keep.nonzero().tolist():
        closed[d, qy * 10:qy * 10 + 11, qx * 10:qx * 10 + 11] = True
    sel = closed[:, None].expand(3, 3, 41, 61)
    assert float((b.stack.detach()[sel] - m.stack.detach()[sel]).abs().max()) <= 3e-4      # two fp32 grid_sample passes; a culled neighbour (logit clamped to -30) bleeds ~4e-6 * 30
    assert bool((b.stack.detach()[..., 3][~inside[:, None].expand(3, 3, 41, 61)] == tiles.CULLED_ALPHA).all())


def test_packer_agrees_with_the_restated_reference_packing():
    m, keep, dyn = _sparse_model()
    sd = m.reference_state_dict()
    src = torch.cat([m.stack.detach()[..., :3], m.stack.detach()[..., 3:].clamp_min(-30.0)], -1)     # what the export samples (see export.py)
    ref = CO.pack_reference_state(src, keep, dyn, 5, 7, m.planedepth)
    for k in ("faces", "faces_dyn"):
        assert torch.equal(sd[k], ref[k])
    # tile contents per quad (the two packers may choose different atlas grids): read both back onto a stack
    a = tiles.stack_from_reference_state(sd, 41, 61, 5, 7, 3)
    b = tiles.stack_from_reference_state(ref, 41, 61, 5, 7, 3)
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and float((a[0] - b[0]).abs().max()) <= 1e-5


def test_atlas_grid_follows_get_hw():
    """MPI.py:366-377."""
    for n in (1, 5, 24, 100, 1000, 4513):
        h, w, r = EX.atlas_grid(n)
        assert h * w - n == r and r >= 1 and w / max(h, 1) <= 4 * 1.6 + 2
    assert EX.atlas_grid(0) == (0, 0, 0)
    n_try = np.arange(int(np.sqrt(1000 / 4)), int(np.sqrt(1000)))
    assert EX.atlas_grid(1000)[0] == n_try[np.argmin(n_try - 1000 % n_try)]


def test_dense_model_exports_every_quad_as_dynamic():
    from videoloop3d_amd.MPV import MPMeshVid
    m = MPMeshVid(_args(), 41, 61, np.eye(4), np.array([[50., 0, 30], [0, 50., 20], [0, 0, 1]]), 1.0, 100.0)
    sd = m.reference_state_dict()
    assert sd["faces"].shape[0] == 0 and sd["faces_dyn"].shape[0] == 2 * 3 * 4 * 6 and sd["self.is_sparse"] is False
    st, keep, dyn = tiles.stack_from_reference_state(sd, 41, 61, 5, 7, 3)
    assert bool(keep.all()) and float((st - m.stack.detach()).abs().max()) <= 1e-4


def _read_png(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(raw):
        n, tag = struct.unpack(">I", raw[pos:pos + 4])[0], raw[pos + 4:pos + 8]
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        chunks.setdefault(tag, b"")
        chunks[tag] += body
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    c = 4 if ctype == 6 else 3
    data = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), np.uint8).reshape(h, 1 + w * c)
    assert depth == 8 and (data[:, 0] == 0).all()
    return data[:, 1:].reshape(h, w, c)


def test_save_mesh_and_texture(tmp_path):
    m, keep, dyn = _sparse_model()
    sd = m.reference_state_dict()
    objs = m.save_mesh(str(tmp_path / "mesh"))
    assert [o.split("/")[-1] for o in objs] == ["mesh.obj", "mesh_dyn.obj"]
    for path, n_quads in zip(objs, (int((keep & ~dyn).sum()), int(dyn.sum()))):
        lines = open(path).read().split("\n")
        v = [l for l in lines if l.startswith("v ")]
        vt = [l for l in lines if l.startswith("vt ")]
        f = [l for l in lines if l.startswith("f ")]
        assert len(f) == 2 * n_quads and len(vt) == 4 * n_quads and 0 < len(v) <= 4 * n_quads
        idx = np.array([[int(p.split("/")[0]) for p in l.split()[1:]] for l in f])
        tidx = np.array([[int(p.split("/")[1]) for p in l.split()[1:]] for l in f])
        assert idx.min() == 1 and idx.max() == len(v) and tidx.min() == 1 and tidx.max() == len(vt)        # unused vertices culled
        uv = np.array([[float(t) for t in l.split()[1:]] for l in vt])
        assert uv.min() > 0 and uv.max() < 1                                                                # texel centres (normalize_uv)
    tex = m.save_texture(str(tmp_path / "tex"))
    assert tex[0].endswith("tex_static.png") and len(tex) == 1 + 3
    img = _read_png(tex[0])
    want = torch.sigmoid(sd["atlas"][0].permute(1, 2, 0))
    assert img.shape == tuple(want.shape) and np.abs(img.astype(np.int32) - (want * 255).type(torch.uint8).numpy().astype(np.int32)).max() == 0
    fr = _read_png(tex[2])
    t1 = sd["atlas_dyn"][1].permute(1, 2, 0)
    want1 = (torch.sigmoid(t1[..., :3]) * torch.sigmoid(t1[..., 3:]) * 255).type(torch.uint8).numpy()
    assert fr.shape == want1.shape and (fr == want1).all()


def test_mpmesh_driver_hooks_and_checkpoint_round_trip():
    """MPMesh carries the hooks train_3d.py calls (get_optimizer :159, get_lrate :301, update_step :298, init_from_mpi :185, save_* :324-328)
    and round-trips through the reference's checkpoint layout, dense (one static atlas) and sparsified."""
    from videoloop3d_amd.MPI import MPMesh
    K = np.array([[50., 0, 30], [0, 50., 20], [0, 0, 1]])
    a = _args(optimizer="adam", lrate=0.05, lrate_decay=100)
    m = MPMesh(a, 41, 61, np.eye(4), K, 1.0, 100.0)
    torch.manual_seed(4)
    with torch.no_grad():
        m.stack.uniform_(-2.0, 2.0)
    opt = m.get_optimizer()
    assert isinstance(opt, torch.optim.Adam) and opt.param_groups[0]["lr"] == 0.05
    assert abs(m.get_lrate(50000)[0][1] - 0.05 * 0.1 ** 0.5) < 1e-12
    m.update_step(3)
    sd = m.reference_state_dict()
    assert "faces_dyn" not in sd and sd["atlas"].shape[0] == 1 and not sd["self.is_sparse"]
    m2 = MPMesh(_args(optimizer="adam", lrate=0.05, lrate_decay=100), 41, 61, np.eye(4), K, 1.0, 100.0)
    m2.init_from_mpi(sd)
    # the atlas is resampled at texel centres with fp32 coordinates: |coordinate error| ~1e-5 texel x a neighbour difference of up to 4
    assert float((m2.stack - m.stack).detach().abs().max()) <= 2e-4
    m2.init_from_mpi(m.state_dict())                       # ... and through this package's own state_dict
    assert torch.equal(m2.stack, m.stack) and torch.equal(m2.stack_mask, m.stack_mask)
    # sparsified
    with torch.no_grad():
        m.stack[..., 3] = -8.0
        m.stack[0, 0, 5:30, 8:40, 3] = 2.0
        m.stack[2, 0, 10:35, 20:55, 3] = 1.5
    m.sparsify_faces(erode_num=1)
    sd = m.reference_state_dict()
    m3 = MPMesh(_args(optimizer="adam", lrate=0.05, lrate_decay=100), 41, 61, np.eye(4), K, 1.0, 100.0)
    m3.init_from_mpi(sd, tile_layout="lattice")      # (a pitch-1 model's export, read back onto the stack it came from)
    assert m3.is_sparse and torch.equal(m3.quad_keep, m.quad_keep) and not hasattr(m3, "stack_mask")
    # the default layout keeps every exported tile with its own border texels: the same samples, tile by tile (11 x 11 per quad of 10 plane pixels)
    m4 = MPMesh(_args(optimizer="adam", lrate=0.05, lrate_decay=100), 41, 61, np.eye(4), K, 1.0, 100.0)
    m4.init_from_mpi(sd)
    assert m4.tile_own == (11, 11) and m4.stack.shape[2:4] == (4 * 11, 6 * 11) and torch.equal(m4.quad_keep, m.quad_keep)
    for d, qy, qx in m.quad_keep.nonzero().tolist()[:5]:
        assert float((m4.stack.detach()[d, 0, qy * 11:qy * 11 + 11, qx * 11:qx * 11 + 11] - m.stack.detach()[d, 0, qy * 10:qy * 10 + 11, qx * 10:qx * 10 + 11]).abs().max()) <= 2e-4
    kt = tiles.quad_to_texel_mask(m.quad_keep, 41, 61)[:, None, :, :, None].expand_as(m.stack)
    # texels inside kept quads come back (the export keeps the quad rectangles; one-texel aprons outside them are resampled from borders)
    from videoloop3d_amd.tiles import CULLED_ALPHA
    inner = torch.zeros_like(kt)
    for d, qy, qx in m.quad_keep.nonzero().tolist():
        inner[d, :, qy * 10:qy * 10 + 11, qx * 10:qx * 10 + 11] = True
    assert float((m3.stack - m.stack).detach()[inner].abs().max()) <= 2e-4
