"""Tile culling on the dense plane stack (first stage).

The reference culls the quads (tiles) of its plane meshes whose alpha is negligible and splits the rest into STATIC tiles
(one texture shared by all frames) and DYNAMIC tiles (a texture per frame) -- MPI.py:288-442 `sparsify_faces`, consumed by
MPV.py:235-288 `init_from_mpi` -- and re-packs them into two atlases.  Here the texture stays the dense `(D,T,Hs,Ws,4)`
stack (the layout the MI355X kernels stream), and the same three-way classification is carried as two small boolean quad
maps `(D,QH,QW)`:
  * culled quads: the render kernels treat a sample that falls into one as NOT COVERED by that plane (`quad_keep` of
    `render_planes`, include/vl3d.h "Tile culling") -- the mesh of the reference has no face there -- and skip, per workgroup,
    the planes of which no kept quad is in sight.  Their texels additionally get the alpha logit CULLED_ALPHA, so code that
    renders the stack without the map (evaluation scripts, the stage-1 model itself) sees them as transparent too;
  * static quads: the T copies of their texels are kept identical by summing their gradient over the frames
    (`tie_static_grad`): one shared texture with the summed gradient, as in the reference's static atlas;
  * dynamic quads: free per frame.
The memory saving of the packed atlases is not reproduced (288 GB of HBM hold the dense stack).  Quads are the
(mpi_h_verts-1) x (mpi_w_verts-1) cells of the vertex grid (utils_mpi.py:80-89), each covering [(q)*c, (q+1)*c] plane pixels
with c = (mpi-1)/(verts-1).
"""
import torch
import torch.nn.functional as F

CULLED_ALPHA = -1e4


# The quad maps of a model are constant between two `sparsify_faces` calls, but they live as bool buffers and the C ABI takes bytes: the uint8 form
# is kept per source tensor (storage address, version counter, view geometry) instead of being converted at every render / step / tie (a launch each:
# two to three of the ~33 of a tile-culled stage-2 iteration).  The entry holds a reference to its source, so the address cannot be recycled.
_U8_CACHE = {}


def as_u8(t):
    """bool / uint8 map -> contiguous uint8 tensor on the same device (cached per source tensor and version)."""
    if t is None:
        return None
    if t.dtype == torch.uint8 and t.is_contiguous():
        return t
    key = (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), t.dtype, str(t.device))
    hit = _U8_CACHE.get(key)
    if hit is not None:
        return hit[1]
    if len(_U8_CACHE) >= 16:
        _U8_CACHE.clear()
    u = t.to(torch.uint8).contiguous()
    _U8_CACHE[key] = (t, u)
    return u


def dilate(alpha, kernelsz=3):
    """utils.py:298-306: max filter, zero padding.  alpha [B,L,H,W]."""
    pad = kernelsz // 2
    return F.max_pool2d(F.pad(alpha, (pad, pad, pad, pad), value=0.0), kernelsz, stride=1)


def erode(alpha, kernelsz=3):
    """utils.py:309-317: min filter, zero padding (so the image border erodes)."""
    pad = kernelsz // 2
    return -F.max_pool2d(F.pad(-alpha, (pad, pad, pad, pad), value=0.0), kernelsz, stride=1)


def quad_max(img, QH, QW):
    """max of img [D,H,W] over the closed plane-pixel rectangle of every quad -> [D,QH,QW]."""
    D, H, W = img.shape
    ch, cw = (H - 1) / QH, (W - 1) / QW
    out = img.new_empty((D, QH, QW))
    for qy in range(QH):
        y0, y1 = int(qy * ch), min(int(-(-(qy + 1) * ch // 1)), H - 1)
        rows = img[:, y0:y1 + 1].amax(1)                       # D,W
        for qx in range(QW):
            x0, x1 = int(qx * cw), min(int(-(-(qx + 1) * cw // 1)), W - 1)
            out[:, qy, qx] = rows[:, x0:x1 + 1].amax(1)
    return out


def cell_vertex_uvs(grid, verts):
    """one axis of the reference's vertex UVs (MPI.py:75-81): cell origins k/grid*2-1, vertices at linspace(0,1,verts) of the cell extent
    (1 - the last origin), in float32 like the reference -> [grid, verts]."""
    origin = torch.arange(grid, dtype=torch.float32) / grid * 2 - 1
    return origin[:, None] + torch.linspace(0, 1, verts)[None] * (1 - origin[-1])


def tile_lattice(grid, verts, atlas_size):
    """the per-quad sample lattice `sparsify_faces` cuts out of the atlas along one axis (MPI.py:296-313): n = round(quad UV extent / 2 *
    (atlas_size - 1)) samples from the quad's first vertex over its UV extent -> (n, atlas texel coordinates [grid, verts-1, n] float32)."""
    uv = cell_vertex_uvs(grid, verts)
    ext = float(uv[0, 1] - uv[0, 0])
    n = int(round(ext / 2 * (atlas_size - 1)))
    pos = uv[:, :-1, None] + torch.linspace(0, ext, n)[None, None]
    return n, pos


def classify_quads_atlas(alpha, loopmask, grid_h, hv, wv, erode_num=2, alpha_thresh=0.03, loop_thresh=0.5, rmfirstlayer=0):
    """The reference's tile classification, as it runs it (MPI.py:288-356, pinned by golden G15): the D planes are laid out as the
    grid_h x (D / grid_h) cells of ONE atlas, the erosions / dilations run on that atlas (so neighbouring cells -- other planes -- bleed
    into each other at cell borders and only the atlas border erodes), every quad is then cut out as a tile of bilinear samples
    (`tile_lattice`, grid_sample, align_corners=True) and classified by its largest sample.  `sparsify_rmfirstlayer` zeroes the first
    mpi_h_verts * mpi_w_verts * rm TILES (MPI.py:343-346 as written: vertices per plane, not quads per plane).
    alpha, loopmask: activated [D,H,W] maps (loopmask None: everything kept is dynamic) -> (keep, dyn) bool [D,hv-1,wv-1]."""
    D, H, W = alpha.shape
    if D % grid_h != 0:
        raise RuntimeError("mpi_d and atlas_grid_h should match")                       # MPI.py:46
    gw = D // grid_h
    QH, QW = hv - 1, wv - 1

    def to_atlas(m):
        return m.reshape(grid_h, gw, H, W).permute(0, 2, 1, 3).reshape(1, 1, grid_h * H, gw * W).float()

    nh, ty = tile_lattice(grid_h, hv, grid_h * H)                                      # [gh,QH,nh] normalised v
    nw, tx = tile_lattice(gw, wv, gw * W)                                              # [gw,QW,nw] normalised u
    # one grid_sample over all tiles: rows = (cell row, quad row, sample), columns = (cell column, quad column, sample)
    grid = torch.stack([tx.reshape(1, -1).expand(grid_h * QH * nh, -1), ty.reshape(-1, 1).expand(-1, gw * QW * nw)], -1)[None]

    def tile_max(atlas):
        samp = F.grid_sample(atlas, grid.to(device=atlas.device, dtype=atlas.dtype), mode="bilinear", padding_mode="zeros", align_corners=True)[0, 0]
        samp = samp.reshape(grid_h, QH, nh, gw, QW, nw).amax((2, 5))                    # gh,QH,gw,QW
        return samp.permute(0, 2, 1, 3).reshape(D, QH, QW)

    a = to_atlas(alpha)
    for _ in range(erode_num):
        a = erode(a)
    for _ in range(erode_num + 2):
        a = dilate(a)
    qa = tile_max(a)
    if rmfirstlayer > 0:
        qa.reshape(-1)[:hv * wv * rmfirstlayer] = 0
    keep = qa > alpha_thresh
    if loopmask is None:
        return keep, keep.clone()
    m = to_atlas(loopmask)
    for _ in range(erode_num):
        m = erode(m)
    for _ in range(erode_num):
        m = dilate(m)
    return keep, keep & (tile_max(m) > loop_thresh)


def classify_quads(alpha, loopmask, QH, QW, erode_num=2, alpha_thresh=0.03, loop_thresh=0.5, rmfirstlayer=0):
    """Plane-by-plane variant of `classify_quads_atlas` for textures that have no atlas layout (no bleeding between planes; a quad is
    judged by the texels of its closed rectangle).  alpha, loopmask: activated [D,H,W] maps in [0,1] (loopmask may be None: everything
    kept is dynamic).  -> (keep, dyn) bool [D,QH,QW].  MPMesh.sparsify_faces uses the atlas variant whenever args.atlas_grid_h divides mpi_d."""
    a = alpha[None]
    for _ in range(erode_num):
        a = erode(a)
    for _ in range(erode_num + 2):
        a = dilate(a)
    qa = quad_max(a[0], QH, QW)
    if rmfirstlayer > 0:
        qa[:rmfirstlayer] = 0
    keep = qa > alpha_thresh
    if loopmask is None:
        return keep, keep.clone()
    m = loopmask[None]
    for _ in range(erode_num):
        m = erode(m)
    for _ in range(erode_num):
        m = dilate(m)
    dyn = keep & (quad_max(m[0], QH, QW) > loop_thresh)
    return keep, dyn


def quad_to_texel_mask(qmask, Hs, Ws, tile=None):
    """[D,QH,QW] bool -> [D,Hs,Ws] bool: true for every texel a bilinear tap of a sample inside a true quad can read (the
    quad's closed rectangle grown by one texel).
    tile = (th, tw): the TILE-EXACT layout (every quad owns a th x tw tile with its own border texels, Hs x Ws = QH th x QW tw): a texel
    belongs to exactly one quad."""
    D, QH, QW = qmask.shape
    if tile is not None and tile[0]:
        th, tw = int(tile[0]), int(tile[1])
        if (Hs, Ws) != (QH * th, QW * tw):
            raise RuntimeError(f"tile-exact layout: a plane of {QH} x {QW} tiles of {th} x {tw} texels is {(QH * th, QW * tw)}, got {(Hs, Ws)}")
        return qmask.repeat_interleave(th, 1).repeat_interleave(tw, 2)
    dev = qmask.device
    y = torch.arange(Hs, device=dev, dtype=torch.int64)
    x = torch.arange(Ws, device=dev, dtype=torch.int64)

    def qi(a, S, n):        # floor(a * n / (S - 1)) clamped, in INTEGER arithmetic: the same side of a quad border as every HIP kernel
        return torch.div(a * n, max(S - 1, 1), rounding_mode="floor").clamp(0, n - 1)
    ylo, yhi = qi(y - 1, Hs, QH), qi(y + 1, Hs, QH)
    xlo, xhi = qi(x - 1, Ws, QW), qi(x + 1, Ws, QW)
    rows_lo, rows_hi = qmask[:, ylo], qmask[:, yhi]              # D,Hs,QW
    return rows_lo[:, :, xlo] | rows_lo[:, :, xhi] | rows_hi[:, :, xlo] | rows_hi[:, :, xhi]


def cull_stack_(stack, keep, tile=None):
    """write CULLED_ALPHA into the alpha logit of every texel no kept quad can read.  stack (D,T,Hs,Ws,4), in place."""
    D, T, Hs, Ws, _ = stack.shape
    dead = ~quad_to_texel_mask(keep, Hs, Ws, tile)
    stack[..., 3].masked_fill_(dead[:, None], CULLED_ALPHA)
    return stack


def quad_grid_args(keep, tile=None):
    """(QH, QW) as the C ABI takes them: NEGATIVE for the tile-exact layout (include/vl3d.h "Tile-exact layout")."""
    QH, QW = int(keep.shape[1]), int(keep.shape[2])
    return (-QH, -QW) if (tile is not None and tile[0]) else (QH, QW)


def tie_static_grad_hip(grad, keep, dyn, assume_culled_zero=False, frame0_only=False, tile=None):
    """`tie_static_grad` as ONE in-place HIP kernel on the (fresh) gradient tensor of the stack -- the hook MPMeshVid installs
    (vl3d_tie_static_grad): static texels read T frames and write T frames, dynamic texels are not touched."""
    from . import _lib as L
    L.check_cuda(grad, keep, dyn)
    D, T, Hs, Ws, C4 = grad.shape
    if C4 != 4 or grad.dtype != torch.float32:
        raise RuntimeError("tie_static_grad_hip: gradient must be (D,T,Hs,Ws,4) float32")
    g = grad if grad.is_contiguous() else grad.contiguous()
    k8, d8 = as_u8(keep), as_u8(dyn)
    with torch.cuda.device(g.device):
        L.check(L.lib().vl3d_tie_static_grad(D, T, Hs, Ws, L.ptr(k8), L.ptr(d8), *quad_grid_args(keep, tile), L.ptr(g),
                                             (1 if assume_culled_zero else 0) | (2 if frame0_only else 0), L.stream_ptr(g.device)),
                "vl3d_tie_static_grad")
    return g


def tie_static_grad(grad, keep, dyn, tile=None):
    """DEFINITION (plain torch, any device; used by the tests as the statement of the rule): gradient of the dense stack
    (D,T,Hs,Ws,4) -> the gradient the reference's (static atlas, dynamic atlas) pair would see: texels only static quads
    read get the SUM over frames in every frame's copy; culled texels get 0."""
    D, T, Hs, Ws, _ = grad.shape
    keep_t = quad_to_texel_mask(keep, Hs, Ws, tile)
    dyn_t = quad_to_texel_mask(dyn, Hs, Ws, tile)
    static_t = (keep_t & ~dyn_t)[:, None, :, :, None]
    tied = grad.sum(dim=1, keepdim=True)
    out = torch.where(static_t, tied.expand_as(grad), grad)
    return out * keep_t[:, None, :, :, None].to(grad.dtype)


class TileAdam(torch.optim.Optimizer):
    """torch.optim.Adam(betas, eps; no amsgrad / weight decay -- MPV.py:199-214) for plane-stack parameters (D,T,Hs,Ws,4) of a
    tile-culled model: one HIP kernel per step that walks only the texels a kept quad can read (vl3d_adam_step_tiles).  Culled
    texels never get a gradient, their moments stay zero and Adam would not move them, so the parameters after every step are
    the ones torch.optim.Adam produces (to fp32 rounding of the fused update) while the 7 streams over the culled 80-95 % of the
    stack disappear.  `quad_keep` may be replaced at any time through `.quad_keep` (None = dense: ONE pass over (p, g, m, v) for any
    contiguous float32 parameter, against the 7 chunked multi-tensor passes of torch.optim.Adam's default path -- 49 launches and 2.3 of the
    4.3 ms of GPU time of a 720p stage-1 iteration, profiles/r03_kernel_stats_s1.csv)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, quad_keep=None, quad_dyn=None, tile=None):
        """quad_dyn (with quad_keep): static texels are treated as ONE parameter with T copies -- their gradient is read from
        frame 0 (where tie_static_grad_hip(..., frame0_only=True) leaves the frame sum), their moments live in frame 0, and the
        new value is written to all copies.  tile = (th, tw): the tile-exact layout (quad_to_texel_mask)."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.quad_keep = quad_keep
        self.quad_dyn = quad_dyn
        self.tile = tile

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib as L
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        qk = as_u8(self.quad_keep)
        qd = None if qk is None else as_u8(self.quad_dyn)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                L.check_cuda(p, p.grad)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("TileAdam: parameters must be contiguous float32 tensors")
                if p.dim() != 5 or p.shape[-1] != 4:
                    # any other parameter of a DENSE model (stage 1's loop-mask texture [D,1,Hs,Ws], MPI.py:115-117): Adam is elementwise,
                    # so the tensor is walked as rows of 16-byte groups
                    if qk is not None or p.numel() % 4 != 0:
                        raise RuntimeError("TileAdam: with a quad map the parameters must be plane stacks (D,T,Hs,Ws,4); dense ones need numel % 4 == 0")
                    rows = p.numel() // p.shape[-1]
                    dims = (1, 1, rows, p.shape[-1] // 4) if p.shape[-1] % 4 == 0 and rows <= 4 * 65535 else (1, 1, 1, p.numel() // 4)
                else:
                    dims = tuple(p.shape[:4])
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                D, T, Hs, Ws = dims
                st["step"] += 1
                with torch.cuda.device(p.device):
                    QH_, QW_ = (0, 0) if qk is None else quad_grid_args(qk, self.tile)
                    L.check(L.lib().vl3d_adam_step_tiles(D, T, Hs, Ws, L.ptr(qk), L.ptr(qd), QH_, QW_, L.ptr(p), L.ptr(g), L.ptr(st["exp_avg"]),
                                                         L.ptr(st["exp_avg_sq"]), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                         st["step"], L.stream_ptr(p.device)), "vl3d_adam_step_tiles")
        return loss


# ---------------------------------------------------------------------------------------------------------------------
# Reading the REFERENCE's checkpoints (MPI.py:207-221 / MPV.py:290-304 state_dicts): plane meshes + texture atlases -> dense stack.
#
# Format (MPV.py:56-104, MPI.py:288-442, utils_mpi.py:80-89): `_verts` is a D x hv x wv vertex grid; every quad of the grid that
# exists is two triangles in `faces` / `faces_dyn` ((v0,v1,v3),(v3,v2,v0), v0 = top-left vertex) with the same two triangles over
# atlas corners in `uvfaces*` -> `uvs*` (normalised [-1,1] coordinates into `atlas` [1,4,Ah,Aw] / `atlas_dyn` [1|T,4,Ah',Aw'],
# sampled with grid_sample(align_corners=True), MPV.py:425-427).  A plane pixel inside a quad maps affinely between the quad's
# corner UVs (barycentric interpolation over two coplanar triangles of an axis-aligned rectangle, MPV.py:394-405).
# Unpinned: no checkpoint of the reference is available here; the reader follows the code cited above and is tested against a
# restatement of the reference's packing (oracle/ckpt_oracle.py).

def _decode_quads(faces, uvfaces, uvs, hv, wv):
    """-> (d, vy, vx) of every quad and its top-left / bottom-right atlas UV."""
    if faces.numel() == 0:
        z = torch.zeros(0, dtype=torch.long)
        return z, z, z, torch.zeros(0, 2), torch.zeros(0, 2)
    quads = faces.reshape(-1, 2, 3).long().cpu()
    v0 = quads[:, 0, 0]
    d, rem = v0 // (hv * wv), v0 % (hv * wv)
    uq = uvfaces.reshape(-1, 2, 3).long().cpu()
    uvs = uvs.detach().float().cpu()
    return d, rem // wv, rem % wv, uvs[uq[:, 0, 0]], uvs[uq[:, 0, 2]]


def _aligned_tiles(parts, hv, wv):
    """If every quad of the checkpoint is a texel-aligned tile of ONE common size -- what `sparsify_faces` writes (MPI.py:403-418:
    tile k of an atlas at texels [ky*th, ky*th + th - 1] x [kx*tw, kx*tw + tw - 1], corner UVs on texel centres) -- return
    ((th, tw), [(kind, d, vy, vx, y0, x0, atlas)]) with integer tile origins; else None."""
    size, out = None, []
    for kind, faces, uvfaces, uvs, atlas in parts:
        if faces is None or faces.numel() == 0:
            continue
        d, vy, vx, uv0, uv3 = _decode_quads(faces, uvfaces, uvs, hv, wv)
        Ah, Aw = atlas.shape[-2:]
        x0, x1 = (uv0[:, 0].double() + 1) / 2 * (Aw - 1), (uv3[:, 0].double() + 1) / 2 * (Aw - 1)
        y0, y1 = (uv0[:, 1].double() + 1) / 2 * (Ah - 1), (uv3[:, 1].double() + 1) / 2 * (Ah - 1)
        c = torch.stack([x0, x1, y0, y1])
        if float((c - c.round()).abs().max()) > 1e-3:
            return None
        tw, th = (c[1] - c[0]).round().long() + 1, (c[3] - c[2]).round().long() + 1
        if size is None:
            size = (int(th[0]), int(tw[0]))
        if bool((th != size[0]).any()) or bool((tw != size[1]).any()) or min(size) < 2:
            return None
        out.append((kind, d, vy, vx, c[2].round().long(), c[0].round().long(), atlas.detach().float().cpu()))
    return None if size is None else (size, out)


def _stack_on_tile_lattice(aligned, D, T, QH, QW):
    """tiles of th x tw texels -> the plane lattice of QH*(th-1)+1 x QW*(tw-1)+1 texels on which neighbouring quads share their border
    row / column (the same samples: `sparsify_faces` cuts neighbouring tiles from the same atlas positions).  Where tiles overlap the
    dynamic tile's texels win over a static tile's, equals are averaged (identical in a fresh checkpoint; a trained stage-2 checkpoint's
    duplicated border texels have drifted apart by what the optimiser made of their separate gradients)."""
    (th, tw), lists = aligned
    Hl, Wl = QH * (th - 1) + 1, QW * (tw - 1) + 1
    acc = {k: [torch.zeros((D, Hl, Wl, T, 4)), torch.zeros((D, Hl, Wl, 1, 1))] for k in ("static", "dyn")}
    iy, ix = torch.arange(th), torch.arange(tw)
    for kind, d, vy, vx, y0, x0, atlas in lists:
        A = atlas.shape[0]
        ay = (y0[:, None] + iy[None])[:, :, None].expand(-1, th, tw)                                # n,th,tw atlas rows
        ax = (x0[:, None] + ix[None])[:, None, :].expand(-1, th, tw)
        tl = atlas[:, :, ay, ax].permute(2, 0, 3, 4, 1)                                              # n,A,th,tw,4
        if A == 1:
            tl = tl.expand(-1, T, -1, -1, -1)
        elif A != T:
            raise RuntimeError(f"checkpoint atlas holds {A} frames, expected 1 or {T}")
        ly = (vy[:, None] * (th - 1) + iy[None])[:, :, None].expand(-1, th, tw)
        lx = (vx[:, None] * (tw - 1) + ix[None])[:, None, :].expand(-1, th, tw)
        dd = d[:, None, None].expand(-1, th, tw)
        acc[kind][0].index_put_((dd, ly, lx), tl.permute(0, 2, 3, 1, 4), accumulate=True)
        acc[kind][1].index_put_((dd, ly, lx), torch.ones((len(d), th, tw, 1, 1)), accumulate=True)
    stack = torch.zeros((D, Hl, Wl, T, 4))
    stack[..., 3] = CULLED_ALPHA
    for kind in ("static", "dyn"):
        val, cnt = acc[kind]
        stack = torch.where(cnt > 0, val / cnt.clamp_min(1), stack)
    return stack.permute(0, 3, 1, 2, 4).contiguous()


def reference_tile_size(sd, hv, wv):
    """(th, tw) when the reference state_dict `sd` holds texel-aligned tiles of one size in its static / dynamic atlases (what `sparsify_faces`
    and `MPV.lod` write, MPI.py:403-418 / MPV.py:146-197), else None (dense cell atlases, hand-made UVs)."""
    if "faces_dyn" not in sd:
        return None
    parts = [("static", sd.get("faces"), sd.get("uvfaces"), sd.get("uvs"), sd.get("atlas")),
             ("dyn", sd["faces_dyn"], sd["uvfaces_dyn"], sd["uvs_dyn"], sd["atlas_dyn"])]
    aligned = _aligned_tiles(parts, hv, wv)
    return None if aligned is None else aligned[0]


def _stack_on_own_tiles(aligned, D, T, QH, QW):
    """tiles of th x tw texels -> the TILE-EXACT plane of QH th x QW tw texels: tile (vy, vx) of plane d occupies rows [vy th, vy th + th),
    columns [vx tw, vx tw + tw) -- every texel of the checkpoint exactly once, the two copies of a border sample that neighbouring tiles hold
    (MPI.py:380-400) side by side and INDEPENDENT, as stage 2 trains them (a static tile's copy is one texture, its dynamic neighbour's moves
    per frame).  The render adds the quad index to a sample's lattice coordinate (csrc/vl3d_render_core.h make_taps_i), so a sample's two
    taps per axis stay inside its own tile -- grid_sample on the reference's atlas between the tile's corner texel centres (MPV.py:394-427)."""
    (th, tw), lists = aligned
    stack = torch.zeros((D, T, QH * th, QW * tw, 4))
    stack[..., 3] = CULLED_ALPHA
    iy, ix = torch.arange(th), torch.arange(tw)
    for kind, d, vy, vx, y0, x0, atlas in lists:
        A = atlas.shape[0]
        ay = (y0[:, None] + iy[None])[:, :, None].expand(-1, th, tw)                                # n,th,tw atlas rows
        ax = (x0[:, None] + ix[None])[:, None, :].expand(-1, th, tw)
        tl = atlas[:, :, ay, ax].permute(2, 0, 3, 4, 1)                                              # n,A,th,tw,4
        if A == 1:
            tl = tl.expand(-1, T, -1, -1, -1)
        elif A != T:
            raise RuntimeError(f"checkpoint atlas holds {A} frames, expected 1 or {T}")
        ly = (vy[:, None] * th + iy[None])[:, None, :, None].expand(-1, T, th, tw)
        lx = (vx[:, None] * tw + ix[None])[:, None, None, :].expand(-1, T, th, tw)
        dd = d[:, None, None, None].expand(-1, T, th, tw)
        tt = torch.arange(T)[None, :, None, None].expand(len(d), T, th, tw)
        stack[dd, tt, ly, lx] = tl
    return stack


def stack_from_reference_state(sd, mpi_h, mpi_w, hv, wv, frm_num, lattice=True, own_borders=False):
    """Reference state_dict (stage-1 MPMesh or stage-2 MPMeshVid, sparsified or not) -> (stack (D,T,Hs,Ws,4) float32 on CPU, quad_keep,
    quad_dyn [D,hv-1,wv-1] bool).  Static quads are written into every frame, culled texels get CULLED_ALPHA.
    lattice (default): a SPARSIFIED checkpoint (texel-aligned tiles of one size, MPI.py:364-436) is copied texel for texel onto the tile
    lattice, Hs x Ws = (hv-1)*(th-1)+1 x (wv-1)*(tw-1)+1 -- the reference's own stage-2 resolution (its `lod` resizes tiles, MPV.py:146-163);
    the planes keep their extent, so the caller renders with texel scale (Ws-1)/(mpi_w-1).  Identical weights, identical image
    (golden G17) for a FRESH checkpoint (duplicated border samples equal).
    own_borders (with lattice): the tile-exact plane of (hv-1)*th x (wv-1)*tw texels instead (`_stack_on_own_tiles`): identical weights for
    ANY checkpoint, trained ones included (golden G19); the caller renders with the lattice's scale ((wv-1)(tw-1))/(mpi_w-1) and the tile size
    (RenderSpec.tile).  Otherwise (dense cell atlases, lattice=False): bilinear resampling onto the (mpi_h, mpi_w) grid of pitch 1."""
    D = int(sd["planedepth"].numel())
    QH, QW = hv - 1, wv - 1
    if lattice and "faces_dyn" in sd and bool(sd.get("self.is_sparse", False)):
        parts_ = [("static", sd.get("faces"), sd.get("uvfaces"), sd.get("uvs"), sd.get("atlas")),
                  ("dyn", sd["faces_dyn"], sd["uvfaces_dyn"], sd["uvs_dyn"], sd["atlas_dyn"])]
        aligned = _aligned_tiles(parts_, hv, wv)
        if aligned is not None:
            T_ = int(sd["atlas_dyn"].shape[0]) if sd["atlas_dyn"].shape[0] > 1 else frm_num
            keep = torch.zeros((D, QH, QW), dtype=torch.bool)
            dyn = torch.zeros((D, QH, QW), dtype=torch.bool)
            for kind, d, vy, vx, *_ in aligned[1]:
                keep[d, vy, vx] = True
                if kind == "dyn":
                    dyn[d, vy, vx] = True
            if own_borders:      # every tile with its own border texels: a TRAINED checkpoint's weights, exactly (golden G19)
                return _stack_on_own_tiles(aligned, D, T_, QH, QW), keep, dyn
            return _stack_on_tile_lattice(aligned, D, T_, QH, QW), keep, dyn
    ch, cw = (mpi_h - 1) / QH, (mpi_w - 1) / QW
    has_dyn_lists = "faces_dyn" in sd
    parts = [("static", sd.get("faces"), sd.get("uvfaces"), sd.get("uvs"), sd.get("atlas"))]
    if has_dyn_lists:
        parts.append(("dyn", sd["faces_dyn"], sd["uvfaces_dyn"], sd["uvs_dyn"], sd["atlas_dyn"]))
    T = frm_num
    if has_dyn_lists and sd["atlas_dyn"].shape[0] > 1:
        T = int(sd["atlas_dyn"].shape[0])
    stack = torch.zeros((D, T, mpi_h, mpi_w, 4), dtype=torch.float32)
    stack[..., 3] = CULLED_ALPHA
    keep = torch.zeros((D, QH, QW), dtype=torch.bool)
    dyn = torch.zeros((D, QH, QW), dtype=torch.bool)
    y = torch.arange(mpi_h, dtype=torch.float64)
    x = torch.arange(mpi_w, dtype=torch.float64)
    # a texel on a quad border belongs to both neighbours: candidates floor(t/c) and ceil(t/c)-1 (equal off the borders)
    qy_c = [(y / ch).floor().clamp(0, QH - 1).long(), ((y / ch).ceil() - 1).clamp(0, QH - 1).long()]
    qx_c = [(x / cw).floor().clamp(0, QW - 1).long(), ((x / cw).ceil() - 1).clamp(0, QW - 1).long()]
    for kind, faces, uvfaces, uvs, atlas in parts:
        if faces is None or faces.numel() == 0:
            continue
        d, vy, vx, uv0, uv3 = _decode_quads(faces, uvfaces, uvs, hv, wv)
        qid = torch.full((D, QH, QW), -1, dtype=torch.long)
        qid[d, vy, vx] = torch.arange(len(d))
        keep[d, vy, vx] = True
        if kind == "dyn":
            dyn[d, vy, vx] = True
        atlas = atlas.detach().float().cpu()
        for p in range(D):
            if not (qid[p] >= 0).any():
                continue
            # per texel: the first candidate quad that exists in this list
            q = torch.full((mpi_h, mpi_w), -1, dtype=torch.long)
            fy = torch.zeros((mpi_h, mpi_w), dtype=torch.float64)
            fx = torch.zeros((mpi_h, mpi_w), dtype=torch.float64)
            for qy in qy_c:
                for qx in qx_c:
                    cand = qid[p][qy][:, qx]
                    take = (q < 0) & (cand >= 0)
                    q = torch.where(take, cand, q)
                    fy = torch.where(take, ((y - qy.double() * ch) / ch)[:, None].expand(mpi_h, mpi_w), fy)
                    fx = torch.where(take, ((x - qx.double() * cw) / cw)[None, :].expand(mpi_h, mpi_w), fx)
            valid = q >= 0
            qq = q.clamp_min(0)
            u = uv0[qq, 0].double() + fx * (uv3[qq, 0] - uv0[qq, 0]).double()
            v = uv0[qq, 1].double() + fy * (uv3[qq, 1] - uv0[qq, 1]).double()
            grid = torch.stack([u, v], dim=-1).float()[None]                                  # 1,H,W,2
            samp = F.grid_sample(atlas, grid.expand(atlas.shape[0], -1, -1, -1), mode="bilinear", padding_mode="zeros", align_corners=True)
            samp = samp.permute(0, 2, 3, 1)                                                       # A,H,W,4
            if samp.shape[0] == 1:
                samp = samp.expand(T, -1, -1, -1)
            m = valid[None, :, :, None]
            stack[p] = torch.where(m, samp, stack[p])
    if not has_dyn_lists:            # a stage-1 checkpoint: everything kept is static here; MPV loads it as dynamic (MPV.py:266-288)
        dyn = keep.clone() if not bool(sd.get("self.is_sparse", False)) else dyn
    return stack, keep, dyn
