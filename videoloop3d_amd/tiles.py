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


def classify_quads(alpha, loopmask, QH, QW, erode_num=2, alpha_thresh=0.03, loop_thresh=0.5, rmfirstlayer=0):
    """MPI.py:319-356.  alpha, loopmask: activated [D,H,W] maps in [0,1] (loopmask may be None: everything kept is dynamic).
    -> (keep, dyn) bool [D,QH,QW]."""
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


def quad_to_texel_mask(qmask, Hs, Ws):
    """[D,QH,QW] bool -> [D,Hs,Ws] bool: true for every texel a bilinear tap of a sample inside a true quad can read (the
    quad's closed rectangle grown by one texel)."""
    D, QH, QW = qmask.shape
    dev = qmask.device
    y = torch.arange(Hs, device=dev, dtype=torch.int64)
    x = torch.arange(Ws, device=dev, dtype=torch.int64)

    def qi(a, S, n):        # floor(a * n / (S - 1)) clamped, in INTEGER arithmetic: the same side of a quad border as every HIP kernel
        return torch.div(a * n, max(S - 1, 1), rounding_mode="floor").clamp(0, n - 1)
    ylo, yhi = qi(y - 1, Hs, QH), qi(y + 1, Hs, QH)
    xlo, xhi = qi(x - 1, Ws, QW), qi(x + 1, Ws, QW)
    rows_lo, rows_hi = qmask[:, ylo], qmask[:, yhi]              # D,Hs,QW
    return rows_lo[:, :, xlo] | rows_lo[:, :, xhi] | rows_hi[:, :, xlo] | rows_hi[:, :, xhi]


def cull_stack_(stack, keep):
    """write CULLED_ALPHA into the alpha logit of every texel no kept quad can read.  stack (D,T,Hs,Ws,4), in place."""
    D, T, Hs, Ws, _ = stack.shape
    dead = ~quad_to_texel_mask(keep, Hs, Ws)
    stack[..., 3].masked_fill_(dead[:, None], CULLED_ALPHA)
    return stack


def tie_static_grad_hip(grad, keep, dyn, assume_culled_zero=False, frame0_only=False):
    """`tie_static_grad` as ONE in-place HIP kernel on the (fresh) gradient tensor of the stack -- the hook MPMeshVid installs
    (vl3d_tie_static_grad): static texels read T frames and write T frames, dynamic texels are not touched."""
    from . import _lib as L
    L.check_cuda(grad, keep, dyn)
    D, T, Hs, Ws, C4 = grad.shape
    if C4 != 4 or grad.dtype != torch.float32:
        raise RuntimeError("tie_static_grad_hip: gradient must be (D,T,Hs,Ws,4) float32")
    g = grad if grad.is_contiguous() else grad.contiguous()
    k8, d8 = keep.to(torch.uint8).contiguous(), dyn.to(torch.uint8).contiguous()
    with torch.cuda.device(g.device):
        L.check(L.lib().vl3d_tie_static_grad(D, T, Hs, Ws, L.ptr(k8), L.ptr(d8), keep.shape[1], keep.shape[2], L.ptr(g),
                                             (1 if assume_culled_zero else 0) | (2 if frame0_only else 0), L.stream_ptr(g.device)),
                "vl3d_tie_static_grad")
    return g


def tie_static_grad(grad, keep, dyn):
    """DEFINITION (plain torch, any device; used by the tests as the statement of the rule): gradient of the dense stack
    (D,T,Hs,Ws,4) -> the gradient the reference's (static atlas, dynamic atlas) pair would see: texels only static quads
    read get the SUM over frames in every frame's copy; culled texels get 0."""
    D, T, Hs, Ws, _ = grad.shape
    keep_t = quad_to_texel_mask(keep, Hs, Ws)
    dyn_t = quad_to_texel_mask(dyn, Hs, Ws)
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

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, quad_keep=None, quad_dyn=None):
        """quad_dyn (with quad_keep): static texels are treated as ONE parameter with T copies -- their gradient is read from
        frame 0 (where tie_static_grad_hip(..., frame0_only=True) leaves the frame sum), their moments live in frame 0, and the
        new value is written to all copies."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.quad_keep = quad_keep
        self.quad_dyn = quad_dyn

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib as L
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        qk = None if self.quad_keep is None else self.quad_keep.to(torch.uint8).contiguous()
        qd = None if (qk is None or self.quad_dyn is None) else self.quad_dyn.to(torch.uint8).contiguous()
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
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                D, T, Hs, Ws = dims
                with torch.cuda.device(p.device):
                    L.check(L.lib().vl3d_adam_step_tiles(D, T, Hs, Ws, L.ptr(qk), L.ptr(qd), 0 if qk is None else qk.shape[1],
                                                         0 if qk is None else qk.shape[2], L.ptr(p), L.ptr(g), L.ptr(st["exp_avg"]),
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


def stack_from_reference_state(sd, mpi_h, mpi_w, hv, wv, frm_num):
    """Reference state_dict (stage-1 MPMesh or stage-2 MPMeshVid, sparsified or not) -> (stack (D,T,mpi_h,mpi_w,4) float32 on CPU,
    quad_keep, quad_dyn [D,hv-1,wv-1] bool).  Static quads are written into every frame, culled texels get CULLED_ALPHA."""
    D = int(sd["planedepth"].numel())
    QH, QW = hv - 1, wv - 1
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
