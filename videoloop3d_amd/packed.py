"""Packed storage of a tile-culled plane stack (reference: MPI.py:364-436 tile packing, MPV.py:235-288, 389-449).

After `sparsify_faces` the reference keeps a STATIC atlas (one frame), a DYNAMIC atlas (T frames) and nothing for culled quads.  The dense
`(D,T,Hs,Ws,4)` stack of this package keeps T copies of static texels and slots for culled ones -- about 6x the reference's texture
memory at the 16 % of kept quads of a typical stage-2 model.  This module stores the same texels, and only them, in a pool of 8 x 8-texel
blocks (the bookkeeping tiles of optim.WindowAdam):

    blocks [D][ceil(Hs/8)][ceil(Ws/8)] int32 = -1                 no kept quad can read a texel of the block: not stored
                                             = slot << 1 | dynamic  slot = index of a 64-texel (1 KiB) unit of the pool;
                                                                    a static block owns ONE slot, a dynamic block T consecutive ones

A texel's class (culled / static / dynamic) is the dense model's (`tiles.quad_to_texel_mask`, `texel_class` in csrc/vl3d_optim.hip); a block
is dynamic when one of its texels is.  A static texel inside a dynamic block keeps its T slots and the dense model's convention (the
parameter lives in frame 0, flush() mirrors it), so every number a packed model computes has the dense model's bits.

The HOT PATH does not change: a training iteration renders from the compact, dense copy of the crop's texel window that
`vl3d_adam_window_catchup` builds (now reading the pool), the backward writes a compact gradient, and `vl3d_adam_window_step` updates the
pool.  Everything else (evaluation renders of chosen frames, lod, export, checkpoints) goes through `PackedLayout.unpack_*`, plane by plane
or frame by frame: the dense stack never exists."""
import torch

from . import tiles

TS = 8      # block side = vl3d_adam_window_tile()


class PackedLayout:
    def __init__(self, quad_keep, quad_dyn, T, Hs, Ws, tile=None):
        """quad_keep / quad_dyn [D,QH,QW] bool (any device; the table is built there).  tile = (th, tw): the tile-exact layout (every quad owns
        its border texels, Hs x Ws = QH th x QW tw): a block is stored when a kept tile overlaps it."""
        self.D, self.T, self.Hs, self.Ws = int(quad_keep.shape[0]), int(T), int(Hs), int(Ws)
        self.tile = (int(tile[0]), int(tile[1])) if (tile is not None and tile[0]) else None
        dev = quad_keep.device
        keep_t = tiles.quad_to_texel_mask(quad_keep.bool(), Hs, Ws, self.tile)                       # D,Hs,Ws: texels a kept quad can read
        dyn_t = tiles.quad_to_texel_mask((quad_keep & quad_dyn).bool(), Hs, Ws, self.tile)           # ... a dynamic quad can read
        th, tw = -(-Hs // TS), -(-Ws // TS)
        pad = (0, tw * TS - Ws, 0, th * TS - Hs)

        def blocks_any(m):
            m = torch.nn.functional.pad(m.float(), pad)
            return m.reshape(self.D, th, TS, tw, TS).amax(dim=(2, 4)) > 0
        present, dynamic = blocks_any(keep_t), blocks_any(dyn_t)
        size = torch.where(dynamic, self.T, 1) * present.long()                            # slots per block
        start = torch.cumsum(size.flatten(), 0) - size.flatten()
        self.n_slots = int(size.sum())
        if self.n_slots >= 2 ** 30:
            raise RuntimeError("packed pool too large for 31-bit slot indices")
        table = torch.where(present.flatten(), start * 2 + dynamic.flatten().long(), torch.full_like(start, -1))
        self.blocks = table.reshape(self.D, th, tw).to(torch.int32).contiguous()
        self.n_static, self.n_dynamic = int((present & ~dynamic).sum()), int(dynamic.sum())
        self._index_cache = {}

    # ---- sizes -----------------------------------------------------------------------------------------------------------
    @property
    def pool_bytes(self):
        return self.n_slots * TS * TS * 16

    @property
    def dense_bytes(self):
        return self.D * self.T * self.Hs * self.Ws * 16

    def to(self, device):
        self.blocks = self.blocks.to(device)
        self._index_cache = {}
        return self

    # ---- addressing (plain torch: conversion, evaluation, export -- not the training loop) ------------------------------------
    def _plane_index(self, d):
        """(texel index of frame 0 [Hs,Ws] int64, frame stride [Hs,Ws] int64, stored [Hs,Ws] bool) of plane d, on the table's device."""
        if d not in self._index_cache:
            if len(self._index_cache) > 4:
                self._index_cache.clear()
            e = self.blocks[d].long()
            dev = e.device
            y = torch.arange(self.Hs, device=dev)
            x = torch.arange(self.Ws, device=dev)
            eb = e[(y // TS)[:, None], (x // TS)[None, :]]
            base = (eb >> 1) * (TS * TS) + ((y % TS) * TS)[:, None] + (x % TS)[None, :]
            self._index_cache[d] = (base.clamp_min(0), (eb & 1) * (TS * TS), eb >= 0)
        return self._index_cache[d]

    def new_pool(self, device, fill=0.0):
        return torch.full((self.n_slots * TS * TS, 4), fill, dtype=torch.float32, device=device)

    @torch.no_grad()
    def pack_plane_(self, pool, d, plane):
        """plane (T,Hs,Ws,4) of the dense model -> the pool (static blocks take frame 0; texels of blocks without storage are dropped)."""
        base, fs, ok = self._plane_index(d)
        plane = plane.to(pool.device, torch.float32)
        for t in range(self.T):
            sel = ok & ((fs > 0) | (t == 0))
            pool[(base + t * fs)[sel]] = plane[t][sel]

    @torch.no_grad()
    def unpack_plane(self, pool, d, frames=None, culled=(0.0, 0.0, 0.0, tiles.CULLED_ALPHA)):
        """-> (len(frames),Hs,Ws,4): plane d of the dense model for the given frames (default all); texels without storage read `culled`."""
        base, fs, ok = self._plane_index(d)
        frames = range(self.T) if frames is None else [int(t) for t in frames]
        out = torch.empty((len(frames), self.Hs, self.Ws, 4), dtype=pool.dtype, device=pool.device)
        fill = torch.tensor(culled, dtype=pool.dtype, device=pool.device)
        for i, t in enumerate(frames):
            v = pool[base + t * fs]
            out[i] = torch.where(ok[..., None], v, fill)
        return out

    @torch.no_grad()
    def unpack_frames(self, pool, frames):
        """-> (D,len(frames),Hs,Ws,4): the frames an evaluation render asks for (MPV.py:439 `atlas_dyn[ts]`)."""
        frames = [int(t) for t in frames]
        if pool.is_cuda and self.blocks.device == pool.device and len(frames) > 0:
            from . import _lib as L
            if min(frames) < 0 or max(frames) >= self.T:
                raise IndexError(f"frame index out of range [0, {self.T})")
            out = torch.empty((self.D, len(frames), self.Hs, self.Ws, 4), dtype=torch.float32, device=pool.device)
            ft = torch.tensor(frames, dtype=torch.int32).to(pool.device, non_blocking=True)
            with torch.cuda.device(pool.device):
                L.check(L.lib().vl3d_packed_unpack_frames(self.D, self.T, self.Hs, self.Ws, L.ptr(self.blocks), L.ptr(pool), len(frames), L.ptr(ft),
                                                          float(tiles.CULLED_ALPHA), L.ptr(out), L.stream_ptr(pool.device)),
                        "vl3d_packed_unpack_frames")
            return out
        return torch.stack([self.unpack_plane(pool, d, frames) for d in range(self.D)], 0)

    @staticmethod
    @torch.no_grad()
    def from_dense(stack, quad_keep, quad_dyn, tile=None):
        """dense (D,T,Hs,Ws,4) stack (any device, e.g. a reference checkpoint resampled on the host) -> (layout, pool on stack.device)."""
        D, T, Hs, Ws, _ = stack.shape
        lay = PackedLayout(quad_keep.to(stack.device), quad_dyn.to(stack.device), T, Hs, Ws, tile)
        pool = lay.new_pool(stack.device)
        for d in range(D):
            lay.pack_plane_(pool, d, stack[d])
        return lay, pool
