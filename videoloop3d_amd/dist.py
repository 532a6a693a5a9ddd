"""Row-band sharding of the render across the GPUs of one node.

The reference has no distributed code (only independent scenes per GPU, run_all.sh:7-14).  This module adds the
MI355X-native multi-GPU form of the hot path: the output frame is split into horizontal row bands exactly like the
reference's training crops (shifted principal point, train_3dvid.py:60-66 / utils.py:196-200); every rank holds only
the plane-stack rows its band can touch (band +- parallax halo), so stack gradients are owner-computed without any
collective, and ONE all-gather over RCCL/xGMI assembles the composited frame.
"""
from dataclasses import dataclass
from typing import List

import torch

from .render import RenderSpec, render_planes


@dataclass(frozen=True)
class Band:
    rank: int
    row0: int          # first output row of the band
    rows: int          # number of output rows
    src0: int          # first plane-stack row held by this rank
    src1: int          # one past the last plane-stack row held


def split_rows(H: int, world: int) -> List[tuple]:
    """near-equal contiguous row ranges, every rank non-empty when H >= world."""
    base, rem = divmod(H, world)
    out, r = [], 0
    for k in range(world):
        n = base + (1 if k < rem else 0)
        out.append((r, n))
        r += n
    return out


def source_row_range(homos: torch.Tensor, row0: int, rows: int, W: int, Hs: int, spec: RenderSpec, margin: int = 2):
    """Conservative [lo,hi) range of plane-stack rows touched by output rows [row0,row0+rows) over all planes.

    A homography maps the band (a convex quad) to a convex quad on every plane, so the extreme source rows are reached
    at the band's four corners (plus the bilinear +1 tap and a safety margin)."""
    hm = homos.detach().double().cpu()
    c = float(spec.pixel_center)
    xs = torch.tensor([0.0 + c, W - 1.0 + c], dtype=torch.float64)
    ys = torch.tensor([row0 + c, row0 + rows - 1.0 + c], dtype=torch.float64)
    pts = torch.stack([xs[[0, 1, 0, 1]], ys[[0, 0, 1, 1]], torch.ones(4, dtype=torch.float64)], 0)   # 3,4
    p = hm @ pts                                                                                  # D,3,4
    if (p[:, 2] <= 0).any():
        return 0, Hs        # degenerate view: keep everything
    y_src = p[:, 1] / p[:, 2]
    if spec.coord_mode == "utils_mpi":
        ty = y_src * (Hs - 1) / Hs
    else:
        ty = y_src * spec.scale[1] + spec.offset[1]
    lo = int(torch.floor(ty.min()).item()) - margin
    hi = int(torch.ceil(ty.max()).item()) + 1 + margin
    return max(0, lo), min(Hs, hi)


def plan_bands(homos: torch.Tensor, H: int, W: int, Hs: int, world: int, spec: RenderSpec) -> List[Band]:
    bands = []
    for rank, (r0, n) in enumerate(split_rows(H, world)):
        lo, hi = source_row_range(homos, r0, n, W, Hs, spec)
        if hi <= lo:
            lo, hi = 0, 1
        bands.append(Band(rank, r0, n, lo, hi))
    return bands


def band_spec(spec: RenderSpec, band: Band, Hs: int) -> RenderSpec:
    """Spec for rendering from the band-local stack rows [src0,src1): shift the texel row by -src0.
    Needs the affine coordinate mode (the utils_mpi normalisation depends on the full Hs)."""
    if spec.coord_mode == "utils_mpi":
        raise RuntimeError("row-band sharding needs coord_mode='affine' (use RenderSpec.mpv() or an affine spec)")
    return RenderSpec(pixel_center=spec.pixel_center, coord_mode="affine", scale=spec.scale,
                      offset=(spec.offset[0], spec.offset[1] - band.src0), border=spec.border,
                      act_order=spec.act_order, rgb_act=spec.rgb_act, alpha_act=spec.alpha_act, variant=spec.variant)


def render_band(local_stack, homos, band: Band, W: int, Hs: int, spec: RenderSpec):
    """Render this rank's band from its local stack rows.  local_stack: (D,T,src1-src0,Ws,4).

    With border='hardcut' the quad-extent test uses the local row count; the halo guarantees no band pixel reaches a
    local boundary that is not also a true plane boundary."""
    assert local_stack.shape[2] == band.src1 - band.src0
    return render_planes(local_stack, homos, band.rows, W, band_spec(spec, band, Hs), window=(band.row0, 0))


def all_gather_frame(band_rgb: torch.Tensor, bands: List[Band], group=None, algo: str = "auto", layout: str = "dense"):
    """The one collective of the render path: composited bands [T,rows_r,W,C] -> full frame [T,H,W,C] on every rank.
    Uses torch.distributed (backend 'nccl' == RCCL over xGMI on ROCm; 'gloo' in the CPU tests).

    algo:
      "ring"    RCCL's all_gather (all_gather_into_tensor when the bands are equal): ring / tree chosen by RCCL.  On the
                fully connected xGMI mesh of one node (7 links x ~153 GB/s per GPU, point to point) a ring moves (N-1)/N of the
                frame over ONE link per GPU: cfg3 at N = 8, 553 MB frame -> 484 MB per rank at ~153 GB/s = ~3.2 ms (an estimate).
      "direct"  all-peers: every rank sends its band straight to each of the N-1 peers and receives theirs -- ONE send and ONE receive
                per peer (a band [T,rows_k,W,C] is one contiguous message: 14 point-to-point operations at N = 8, where rounds 4-5
                posted one per frame and peer, 700 at T = 50), all in ONE grouped launch (batch_isend_irecv -> ncclGroupStart/End):
                each of the 7 links carries one 69 MB band per direction concurrently -> ~0.45 ms (an estimate: no multi-GPU box here).
      "auto"    "direct" for world > 2 on device tensors, else "ring".
    layout:
      "dense"   the [T,H,W,C] frame (the received bands are copied into place: N - 1 strided copies on the caller's stream, 484 MB at cfg3 /
                N = 8 -- ~0.2 ms beside a backward of ~1.5 ms, on the side stream the caller gathers on);
      "bands"   no copy: the list of the N bands [T,rows_k,W,C] as they arrived (this rank's own band is `band_rgb` itself); `frame_rows`
                assembles a row range from it -- what the band loss needs is its own band and < patch_size rows of its neighbours'.
    Every rank gets bit-identical frames from either algorithm (pure data movement)."""
    import torch.distributed as dist
    T, _, W, C = band_rgb.shape
    rows = [b.rows for b in bands]
    world = len(bands)
    if algo == "auto":
        algo = "direct" if (world > 2 and band_rgb.is_cuda) else "ring"
    if algo not in ("ring", "direct"):
        raise RuntimeError(f"all_gather_frame: unknown algo {algo!r}")
    if layout not in ("dense", "bands"):
        raise RuntimeError(f"all_gather_frame: unknown layout {layout!r}")
    band_rgb = band_rgb.contiguous()
    if algo == "direct":
        rank = dist.get_rank(group)
        parts = [band_rgb if k == rank else torch.empty((T, rows[k], W, C), dtype=band_rgb.dtype, device=band_rgb.device) for k in range(world)]
        ops = []
        for k in range(world):      # every rank lists its peers in rank order: sends and receives of a pair match inside the one group
            if k == rank:
                continue
            peer = k if group is None else dist.get_global_rank(group, k)
            if rows[rank]:
                ops.append(dist.P2POp(dist.isend, band_rgb, peer, group))
            if rows[k]:
                ops.append(dist.P2POp(dist.irecv, parts[k], peer, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return parts if layout == "bands" else _dense_frame(parts)
    # An all-gather of EQUAL pieces: ragged bands (H not a multiple of the world size: 60 rows over 8 ranks are 8,8,8,8,7,7,7,7) are padded to the tallest
    # and trimmed on arrival -- gloo refuses unequal pieces outright, and RCCL's list form falls back to one broadcast per rank.
    rmax = max(rows)
    mine = band_rgb
    if band_rgb.shape[1] != rmax:
        mine = torch.cat([band_rgb, band_rgb.new_zeros((T, rmax - band_rgb.shape[1], W, C))], dim=1)
    if band_rgb.is_cuda:
        out = torch.empty((world, T, rmax, W, C), dtype=band_rgb.dtype, device=band_rgb.device)
        dist.all_gather_into_tensor(out, mine.contiguous(), group=group)
        padded = list(out.unbind(0))
        if rmax == min(rows) and layout == "dense":
            return out.permute(1, 0, 2, 3, 4).reshape(T, sum(rows), W, C)
    else:
        padded = [torch.empty((T, rmax, W, C), dtype=band_rgb.dtype, device=band_rgb.device) for _ in rows]
        dist.all_gather(padded, mine.contiguous(), group=group)
    parts = [q if r == rmax else q[:, :r] for q, r in zip(padded, rows)]
    return parts if layout == "bands" else torch.cat(parts, dim=1)


def _dense_frame(parts):
    """bands [T,rows_k,W,C] -> the frame [T,H,W,C] (one strided copy per band)."""
    T, _, W, C = parts[0].shape
    frame = torch.empty((T, sum(p.shape[1] for p in parts), W, C), dtype=parts[0].dtype, device=parts[0].device)
    r = 0
    for p in parts:
        frame[:, r:r + p.shape[1]].copy_(p)
        r += p.shape[1]
    return frame


def frame_rows(parts, lo: int, hi: int) -> torch.Tensor:
    """rows [lo, hi) of the frame from its bands (all_gather_frame(..., layout="bands")) -> [T,hi-lo,W,C]: a band's own rows plus the few
    neighbouring rows a consumer needs (loss_band_rows), without assembling the whole frame."""
    out, r = [], 0
    for p in parts:
        a, b = max(lo, r), min(hi, r + p.shape[1])
        if b > a:
            out.append(p[:, a - r:b - r])
        r += p.shape[1]
    return out[0] if len(out) == 1 else torch.cat(out, dim=1)


def halo_overlaps(bands: List[Band], rank: int):
    """[(peer, lo, hi)]: the plane-stack row ranges [lo, hi) this rank holds in common with each other rank (the parallax halos:
    rows replicated on both)."""
    me = bands[rank]
    out = []
    for b in bands:
        if b.rank == rank:
            continue
        lo, hi = max(me.src0, b.src0), min(me.src1, b.src1)
        if hi > lo:
            out.append((b.rank, lo, hi))
    return out


def exchange_halo_grads(g_local: torch.Tensor, bands: List[Band], rank: int = None, group=None) -> torch.Tensor:
    """Make the sharded stack gradient training-complete (SURVEY §8e "sum the halo strips with one neighbour exchange").

    Rank r holds the stack rows [src0, src1) its band can touch and its backward produces the gradient of ITS band's pixels
    w.r.t. those rows.  Rows inside a parallax halo are replicated on the neighbouring rank(s), each holding only a partial
    gradient for them; an optimiser step on the partial gradients would let the replicas drift apart.  This adds, IN PLACE, the
    peers' partial gradients of every shared row range: one grouped send/recv of the overlapping strips per step (point to point
    over xGMI; cfg3 at N = 8: a ~70-row strip x D x T = ~0.45 GB per neighbour), no all-reduce of the 23.6 GB gradient.
    Afterwards every replica of a row holds the same complete gradient -- the single-GPU gradient of that row -- bit for bit on
    all its holders (the partial sums are added in rank order on every holder).  g_local: (D,T,src1-src0,Ws,4)."""
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank(group)
    me = bands[rank]
    assert g_local.shape[2] == me.src1 - me.src0
    ov = halo_overlaps(bands, rank)
    if not ov:
        return g_local
    send = {p: g_local[:, :, lo - me.src0:hi - me.src0].contiguous() for p, lo, hi in ov}
    recv = {p: torch.empty_like(send[p]) for p, _, _ in ov}
    ops = []
    for p, _, _ in ov:
        peer = p if group is None else dist.get_global_rank(group, p)
        ops.append(dist.P2POp(dist.isend, send[p], peer, group))
        ops.append(dist.P2POp(dist.irecv, recv[p], peer, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    # rows shared by more than two ranks (halo taller than a band): sum all holders' parts in rank order, identically everywhere
    holders = {}
    for p, lo, hi in ov:
        for r in range(lo, hi):
            holders.setdefault(r, []).append(p)
    # segment the local rows into runs with the same holder set
    r = me.src0
    while r < me.src1:
        hs = tuple(sorted(holders.get(r, [])))
        e = r + 1
        while e < me.src1 and tuple(sorted(holders.get(e, []))) == hs:
            e += 1
        if hs:
            order = sorted(hs + (rank,))
            acc = None
            for q in order:
                if q == rank:
                    part = g_local[:, :, r - me.src0:e - me.src0]
                else:
                    lo_q = max(me.src0, bands[q].src0)
                    part = recv[q][:, :, r - lo_q:e - lo_q]
                acc = part.clone() if acc is None else acc + part
            g_local[:, :, r - me.src0:e - me.src0] = acc
        r = e
    return g_local


# ---------------------------------------------------------------------------------------------------------------------
# Looping loss on row bands: every spatial patch location is independent but needs all frames, so the loss is
# sharded over pixel rows like the render.  After the all-gather every rank holds the full composited frame; rank r owns the pixel
# rows of ITS render band, needs the NN of every patch row that covers one of them (a halo of < ps rows of the neighbours' bands,
# recomputed rather than exchanged) and sums the robust loss over its own rows only.  x enters the loss only through the residual
# at owned rows (y2x is built under no_grad, utils_vid.py:322), so d loss / d frame is non-zero on the rank's own band only: no
# gradient exchange, one scalar all-reduce.

def loss_band_rows(row0: int, rows: int, H: int, ps: int, stride: int):
    """Pixel rows [a, b) of the (already trimmed, (H-ps) % stride == 0) frame a rank needs so that every patch row covering one of
    its owned rows [row0, row0+rows) is complete; the sub-image's own patch grid coincides with the global one."""
    h_o = (H - ps) // stride + 1
    r1 = min(row0 + rows, H)
    by_lo = max(0, -((-(row0 - ps + 1)) // stride))            # ceil((row0 - ps + 1) / stride)
    by_hi = min(h_o - 1, (r1 - 1) // stride)
    return by_lo * stride, by_hi * stride + ps


def looping_loss_band(x, y, row0, rows, patch_size=7, stride=2, patcht_size=7, stridet=2, rou=0, scaling=0.2, alpha=1e10, **_):
    """Rank-local part of Patch3DGPNNLowMemLoss (utils_vid.py:289-349) for the owned pixel rows [row0, row0+rows) of the full frame
    x [1,3,T,H,W] / y [1,3,F,H,W]: returns (sum of robust losses over the owned rows of the trimmed frame, their element count).
    The global loss is all_reduce(sum) / all_reduce(count); the gradient reaches x on the owned rows only."""
    from .utils_vid import _RobustMean, _nn_and_fold, fit_patch
    t, h, w = x.shape[-3:]
    h = fit_patch(h, "patch_height", patch_size, stride)
    w = fit_patch(w, "patch_width", patch_size, stride)
    t = fit_patch(t, "frame_num", patcht_size, stridet)
    r1 = min(row0 + rows, h)                                    # rows trimmed off the frame belong to nobody
    if r1 <= row0:
        return x.new_zeros(()), 0
    a, b = loss_band_rows(row0, r1 - row0, h, patch_size, stride)
    xs, ys = x[..., :t, a:b, :w], y[..., a:b, :w]
    al = None if alpha > 100 else alpha
    with torch.no_grad():
        y2x, _, _ = _nn_and_fold(xs, ys, patch_size, patcht_size, stride, stridet, al, normalize=True)
    x_own = xs[..., row0 - a:r1 - a, :]
    y2x_own = y2x[..., row0 - a:r1 - a, :]
    n = x_own.numel()
    return _RobustMean.apply(x_own, y2x_own, rou, scaling) * n, n
