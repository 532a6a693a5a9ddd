"""Crop-aware Adam for the dense plane stack (csrc/vl3d_optim.hip; the optimiser of train_3dvid.py:263-290 / MPV.py:199-214).

`WindowAdam` produces the parameters torch.optim.Adam(betas, eps; no amsgrad / weight decay) would produce -- to fp32 rounding of
the one-pass update -- while touching only the texels the current training crop can reach:

  forward   `window_leaf(window)` computes the window's CURRENT parameters (replaying, in registers, the zero-gradient Adam steps its
            bookkeeping tiles (8 x 8 texels) have missed: momentum keeps moving a texel after its gradient is gone) into a compact (D,T,wh,ww,4)
            copy, an autograd LEAF; the render reads it and the backward fills its .grad -- a compact gradient, no zero fill of the
            rest of the stack.  The stack itself is not written.
  step()    replays the same missed steps again and applies the step to the window from that compact gradient: (p, m, v) of the
            window are read once and written once per iteration; every other tile's update stays deferred.
  flush()   replays everything outstanding (before checkpoints, lod(), evaluation renders; MPMeshVid calls it).

Without a pending window (someone filled `p.grad` densely) step() falls back to the dense update: flush + full-window step.

`fused_backward=True`: the step is taken INSIDE the render's backward (vl3d_render_bwd_adam) -- the owner-computes kernel
applies the update where it would have stored a texel's gradient, so the window's gradient is never written or read back (6 streams of
the window instead of 2 + 7).  With quad maps that holds for the DYNAMIC texels; a static texel's gradient is still stored and the step
kernel behind the backward sums it over the frames (static texels only).  `loss.backward()` then leaves the parameters updated and `step()` only does its periodic housekeeping; the
learning rate is the group's at the time of the backward (train_3dvid.py:263-277 sets it before the iteration).  The parameters are bit
for bit those of the two-kernel path.  Contract: one backward per window_leaf(), nothing else reads the leaf's gradient (it stays None).
"""
import ctypes as C

import struct
import warnings

import torch

from . import _lib as L


def tile_side():
    """side of the optimiser's bookkeeping tiles in texels (a constant of the HIP library)."""
    return int(L.lib().vl3d_adam_window_tile())


def align_window(y0, y1, x0, x1, Hs, Ws):
    """[y0,y1) x [x0,x1) clamped to the plane and grown to the bookkeeping tiles -> (y0, x0, wh, ww)."""
    ts = tile_side()
    y0, x0 = max(0, y0) // ts * ts, max(0, x0) // ts * ts
    y1, x1 = min(Hs, -(-min(Hs, y1) // ts) * ts), min(Ws, -(-min(Ws, x1) // ts) * ts)
    return y0, x0, max(y1 - y0, 0), max(x1 - x0, 0)


def crop_window(spec, Hs, Ws, homos, H, W, margin=3, per_plane=False):
    """texel window (y0, x0, wh, ww), aligned to the optimiser's bookkeeping tiles, that contains every tap of every pixel of the
    H x W view on every plane of an Hs x Ws stack rendered under `spec`: the image of the view's corners under the plane homographies
    (convex: extremes at the corners), plus the +1 bilinear tap and a margin.  homos [D,3,3] on the HOST (float64).
    per_plane: also return the planes' own boxes [D,4] = (y0, y1, x0, x1) (same rule per plane; the window is their union), or None
    when a plane has the view behind it."""
    import numpy as np
    c = float(spec.pixel_center)
    pts = torch.tensor([[c, W - 1 + c, c, W - 1 + c], [c, c, H - 1 + c, H - 1 + c], [1.0, 1.0, 1.0, 1.0]], dtype=torch.float64)
    q = homos.double() @ pts                                                                  # D,3,4
    if bool((q[:, 2] <= 1e-9).any()):
        return ((0, 0, Hs, Ws), None) if per_plane else (0, 0, Hs, Ws)
    tx = q[:, 0] / q[:, 2] * spec.scale[0] + spec.offset[0]
    ty = q[:, 1] / q[:, 2] * spec.scale[1] + spec.offset[1]
    # per plane the extremes of the footprint (numpy, not torch: a dozen tiny torch CPU ops cost 2 ms per call on a 256-core host --
    # thread-pool wake-ups -- and starve the launch thread: the iteration fell from 168 to 30 it/s)
    tyn, txn = ty.numpy(), tx.numpy()
    ymin, ymax, xmin, xmax = tyn.min(1), tyn.max(1), txn.min(1), txn.max(1)
    tile = getattr(spec, "tile", (0, 0))
    if tile[0]:
        # tile-exact layout: (tx, ty) are LATTICE coordinates; a texel of quad q lies q columns to the right of its lattice position, and a
        # lattice column on a quad border is two texel columns (the lower end takes the quad strictly below, the upper end the quad at or above)
        th, tw = int(tile[0]), int(tile[1])
        QH, QW = Hs // th, Ws // tw
        ymin = ymin + np.clip(np.ceil(ymin / (th - 1) - 1e-6) - 1, 0, QH - 1)
        ymax = ymax + np.clip(np.floor(ymax / (th - 1) + 1e-6), 0, QH - 1)
        xmin = xmin + np.clip(np.ceil(xmin / (tw - 1) - 1e-6) - 1, 0, QW - 1)
        xmax = xmax + np.clip(np.floor(xmax / (tw - 1) + 1e-6), 0, QW - 1)
    win = align_window(int(np.floor(ymin.min())) - margin, int(np.ceil(ymax.max())) + 2 + margin,
                       int(np.floor(xmin.min())) - margin, int(np.ceil(xmax.max())) + 2 + margin, Hs, Ws)
    if not per_plane:
        return win
    # the planes' boxes by the same rule, vectorised (align_window per plane)
    ts = tile_side()
    ylo = np.maximum(np.floor(ymin).astype(np.int64) - margin, 0) // ts * ts
    xlo = np.maximum(np.floor(xmin).astype(np.int64) - margin, 0) // ts * ts
    yhi = np.minimum(np.ceil(ymax).astype(np.int64) + 2 + margin, Hs)
    xhi = np.minimum(np.ceil(xmax).astype(np.int64) + 2 + margin, Ws)
    yhi = np.minimum(-(-yhi // ts) * ts, Hs)
    xhi = np.minimum(-(-xhi // ts) * ts, Ws)
    boxes = np.stack([ylo, np.maximum(yhi, ylo), xlo, np.maximum(xhi, xlo)], axis=1).astype(np.int32)
    return win, boxes


class WindowAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, quad_keep=None, quad_dyn=None, culled_alpha=-1e4, max_defer=32, layout=None,
                 lean_window=True, fused_backward=False, tile=None):
        """quad_keep / quad_dyn [D,QH,QW] (a tile-culled model, videoloop3d_amd/tiles.py): culled texels are no parameters, a texel only
        static quads can read is ONE parameter stored in frame 0 of the stack (the reference's static atlas, MPV.py:235-288) -- the
        window copy shows it in every frame, the step sums its gradient over the frames and writes that one copy; flush() refreshes
        the other frames' slots so that the dense stack reads consistently."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.quad_keep = None if quad_keep is None else quad_keep.to(torch.uint8).contiguous()
        self.quad_dyn = None if (quad_keep is None or quad_dyn is None) else quad_dyn.to(torch.uint8).contiguous()
        self.culled_alpha = float(culled_alpha)
        # tile = (th, tw): the TILE-EXACT layout (every quad owns its border texels; the planes are QH th x QW tw texels): a texel has the
        # class of the one quad that holds it.  The C ABI takes it as a NEGATIVE quad grid (include/vl3d.h).
        self.tile = (int(tile[0]), int(tile[1])) if (tile is not None and tile[0] and self.quad_keep is not None) else None
        ps = [p for g in self.param_groups for p in g["params"]]
        if len(ps) != 1:
            raise RuntimeError("WindowAdam drives exactly one parameter: the plane stack (D,T,Hs,Ws,4)")
        self.p = ps[0]
        self.pending = None          # (window, compact leaf) of the forward since the last step
        self.t = 0
        # packed.PackedLayout: the parameter is the pool of 8 x 8-texel blocks of a tile-culled model (static blocks once, dynamic blocks
        # per frame, culled blocks not at all) instead of the dense (D,T,Hs,Ws,4) stack; moments live in pools of the same shape
        self.layout = layout
        if layout is not None and self.quad_keep is None:
            raise RuntimeError("WindowAdam: a packed layout belongs to a tile-culled model (quad_keep / quad_dyn)")
        # bound on the deferral: after every step, tiles that have missed max_defer steps are brought up to date (exactly: the same
        # replay), so a crop window that comes back after a whole epoch of other crops replays at most max_defer steps per texel
        # instead of the epoch's length, twice.  0 = unbounded.
        self.max_defer = int(max_defer)
        # lean window copies: the compact leaf is a view of ONE persistent buffer (zero-filled when it is allocated or grown, only ever
        # written by the catch-up: finite everywhere), and the catch-up leaves alone what the render cannot read with a non-zero weight --
        # culled texels (shown as (0, 0, 0, culled_alpha) otherwise) and texels outside their plane's box.  For a tile-culled model those
        # are most of the window: 2.3 GB of stores per iteration at the reference's crop for the 16 % that are parameters.  The leaf of
        # one forward is overwritten by the next window_leaf() call (one windowed forward per step is the contract anyway).
        self.lean_window = bool(lean_window)
        self._compact_buf = None
        self.fused_backward = bool(fused_backward)
        self._fused_ack = self._warned_no_step = False
        self._gfb = self._bwd_scratch = None
        self._boxes_dev = self._class_dev = None
        self.fused_steps = 0             # steps taken inside a backward (diagnostics / tests)

    # ---- state ----------------------------------------------------------------------------------------------------------
    def dims(self):
        if self.layout is not None:
            return self.layout.D, self.layout.T, self.layout.Hs, self.layout.Ws
        return tuple(self.p.shape[:4])

    def _blocks(self):
        return None if self.layout is None else L.ptr(self.layout.blocks)

    def _st(self):
        p = self.p
        L.check_cuda(p)
        if self.layout is not None:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.numel() != self.layout.n_slots * 256 or self.layout.blocks.device != p.device:
                raise RuntimeError("WindowAdam: the parameter must be the contiguous float32 pool of its packed layout, on the layout's device")
        elif p.dim() != 5 or p.shape[-1] != 4 or p.dtype != torch.float32 or not p.is_contiguous():
            raise RuntimeError("WindowAdam: the parameter must be a contiguous float32 plane stack (D,T,Hs,Ws,4)")
        st = self.state[p]
        if not st:
            D, T, Hs, Ws = self.dims()
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
            ts = tile_side()
            st["last_step"] = torch.zeros((D, (Hs + ts - 1) // ts, (Ws + ts - 1) // ts), dtype=torch.int32, device=p.device)
            st["hist"] = torch.zeros((1024, 2), dtype=torch.float32, device=p.device)
        return st

    def _quads(self):
        qk, qd = self.quad_keep, self.quad_dyn
        sgn = -1 if self.tile is not None else 1
        return L.ptr(qk), L.ptr(qd), (0 if qk is None else sgn * qk.shape[1]), (0 if qk is None else sgn * qk.shape[2])

    def _catchup(self, window, upto, compact, mirror=False, boxes=None, lean=False):
        st, p = self._st(), self.p
        D, T, Hs, Ws = self.dims()
        y0, x0, wh, ww = window
        b1, b2 = self.param_groups[0]["betas"]
        qk, qd, QH, QW = self._quads()
        with torch.cuda.device(p.device):
            L.check(L.lib().vl3d_adam_window_catchup_boxes(D, T, Hs, Ws, y0, x0, wh, ww, L.ptr(p), L.ptr(st["exp_avg"]),
                                                           L.ptr(st["exp_avg_sq"]), L.ptr(st["last_step"]), L.ptr(st["hist"]), int(upto),
                                                           float(b1), float(b2), float(self.param_groups[0]["eps"]), L.ptr(compact), qk, qd,
                                                           QH, QW, self.culled_alpha, (1 if mirror else 0) | (2 if lean else 0),
                                                           None if boxes is None else boxes.ctypes.data,
                                                           self._blocks(), L.stream_ptr(p.device)),
                    "vl3d_adam_window_catchup")

    # ---- forward side ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def window_leaf(self, window, plane_boxes=None):
        """window = (y0, x0, wh, ww), tile aligned (align_window) -> compact (D,T,wh,ww,4) leaf (requires_grad) holding the CURRENT
        parameters of the window.  At most one window per step; a second forward before step() takes the dense path.
        plane_boxes (optional): int32 [D,4] = (y0, y1, x0, x1) per plane in plane texels, tile aligned, inside the window -- the part of the
        window this plane's taps can reach.  Outside its box a plane's texels get no gradient in this iteration by construction: their
        slots of the leaf are zeros, their update stays deferred."""
        p = self.p
        D, T, Hs, Ws = self.dims()
        y0, x0, wh, ww = window
        if plane_boxes is not None:                       # stays on the HOST (numpy int32 [D,4]): the table travels in the kernel arguments
            import numpy as np
            plane_boxes = np.ascontiguousarray(np.asarray(plane_boxes, dtype=np.int32).reshape(D, 4))
        if self.lean_window:
            n = D * T * wh * ww * 4
            if self._compact_buf is None or self._compact_buf.numel() < n or self._compact_buf.device != p.device:
                self._compact_buf = None                                       # (release before growing)
                # (an eighth of headroom: the windows of a pyramid level's crops differ by a few bookkeeping tiles, and every growth is a multi-GB
                # hipMalloc in the middle of training -- usually 2 ms, on some boxes 1.2 s: docs/measurement_log.md, round 6)
                self._compact_buf = torch.zeros(n + n // 8, dtype=p.dtype, device=p.device)
            compact = self._compact_buf[:n].view(D, T, wh, ww, 4)
        else:
            compact = torch.empty((D, T, wh, ww, 4), dtype=p.dtype, device=p.device)
        self._catchup(window, self.t, compact, boxes=plane_boxes, lean=self.lean_window)
        compact.requires_grad_(True)
        if self._stepped(self.pending):
            # the previous backward took its step and step() was not called since: legal (step() is housekeeping only then), but the caller
            # should know that nothing accumulates -- every backward of a fused window is one Adam step
            self._check_stepped_leaf(self.pending)
            if not self._warned_no_step:
                self._warned_no_step = True
                warnings.warn("WindowAdam(fused_backward=True): a second windowed forward before optimizer.step() -- each loss.backward() has "
                              "already applied its own Adam step (no gradient accumulation on this path)", RuntimeWarning, stacklevel=3)
            self.pending = None
        if self.pending is not None:
            self.pending = "multiple"
        else:
            self.pending = (window, compact, plane_boxes)
        return compact

    @staticmethod
    def _stepped(pend):
        return isinstance(pend, tuple) and len(pend) == 2 and pend[0] == "stepped"

    @staticmethod
    def _check_stepped_leaf(pend):
        """after a fused backward the window leaf must not hold a gradient: one that arrived through another autograd path would be dropped"""
        if pend[1].grad is not None:
            raise RuntimeError("WindowAdam: the window leaf received a gradient outside the render's fused backward (another autograd path into "
                               "the leaf); that gradient is not part of the step the backward took -- use fused_backward=False for such graphs")

    def acknowledge_fused_backward(self):
        """the training loop knows that loss.backward() applies the update (train_3dvid.run_iter, train_3d.run_iter call this): no warning"""
        self._fused_ack = True

    @torch.no_grad()
    def flush(self):
        """make the whole stack current for the steps taken so far (exact replay of the deferred zero-gradient updates)."""
        if self.t == 0 or not self.state.get(self.p):
            return
        D, T, Hs, Ws = self.dims()
        self._catchup((0, 0, Hs, Ws), self.t, None, mirror=self.quad_keep is not None)

    # ---- checkpointing -------------------------------------------------------------------------------------------------
    def state_dict(self):
        """torch.optim.Adam's layout after a flush(): exp_avg / exp_avg_sq / step per parameter.  The deferral bookkeeping (per-tile
        step table, per-step scalar history) is not state once everything is current: load_state_dict() restarts it at `step`."""
        self.flush()
        sd = super().state_dict()
        # (the packed state shares its inner dicts with the live optimiser: build new ones)
        sd["state"] = {k: {**{kk: vv for kk, vv in st.items() if kk not in ("last_step", "hist")}, "step": torch.tensor(float(self.t))}
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.pending = None
        st = self.state.get(self.p)
        if st:
            self.t = int(st.pop("step", torch.tensor(0.0)).item())
            D, T, Hs, Ws = self.dims()
            ts = tile_side()
            # every tile is current for step t (the saved moments came out of a flush); the scalars of steps <= t are never replayed again
            st["last_step"] = torch.full((D, (Hs + ts - 1) // ts, (Ws + ts - 1) // ts), self.t, dtype=torch.int32, device=self.p.device)
            st["hist"] = torch.zeros((max(1024, 2 * (self.t + 1)), 2), dtype=torch.float32, device=self.p.device)

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none)
        if isinstance(self.pending, tuple) and self.pending[1].grad is not None:
            self.pending[1].grad = None

    # ---- the step inside the render's backward -------------------------------------------------------------------------
    def fuses(self, stack, spec):
        """does the backward of a render of `stack` (the pending window leaf?) under `spec` take this optimiser's step itself?"""
        pend = self.pending
        return (self.fused_backward and isinstance(pend, tuple) and len(pend) == 3 and self._is_leaf(pend, stack) and stack.dtype == torch.float32
                and (stack.shape[1] >= 2 or self.quad_keep is not None) and spec.coord_mode == "affine" and spec.border == "hardcut" and spec.act_order == "post"
                and spec.rgb_act == "sigmoid" and spec.alpha_act == "sigmoid" and (int(spec.variant) & 0xf) in ((0, 3, 5) if self.quad_keep is not None else (0,))
                and not getattr(spec, "uv_noise_seed", 0)       # (add_uv_noise: the atomics backward + the step kernel)
                and (tuple(getattr(spec, "tile", (0, 0))) == (self.tile or (0, 0))))

    @staticmethod
    def _is_leaf(pend, stack):
        return pend[1].data_ptr() == stack.data_ptr() and pend[1].shape == stack.shape

    @torch.no_grad()
    def backward_step(self, desc, stack, homos, rgb, alpha, g_rgb, g_alpha, g_reg, reg_state, g_asum):
        """called by the render's autograd backward (render._RenderPlanes) instead of vl3d_render_bwd: backward + step of the pending window in
        one pass (vl3d_render_bwd_adam).  -> the scratch buffer of the call (its first word: 1 = the owner-computes kernels ran)."""
        pend = self.pending
        if not (isinstance(pend, tuple) and len(pend) == 3 and self._is_leaf(pend, stack)):
            raise RuntimeError("WindowAdam: the fused backward belongs to the pending window leaf (one backward per window_leaf())")
        st, p = self._st(), self.p
        grp = self.param_groups[0]
        b1, b2 = grp["betas"]
        t = self._register_step(st, float(grp["lr"]), b1, b2)
        window, _, boxes = pend
        D, T, Hs, Ws = self.dims()
        dev = p.device
        aw = L.AdamWindow()
        aw.Hs, aw.Ws, aw.y0, aw.x0 = Hs, Ws, window[0], window[1]
        aw.param, aw.exp_avg, aw.exp_avg_sq = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
        aw.last_step, aw.hist = st["last_step"].data_ptr(), st["hist"].data_ptr()
        aw.lr, aw.beta1, aw.beta2, aw.eps, aw.step = float(grp["lr"]), float(b1), float(b2), float(grp["eps"]), t
        if boxes is not None and D <= 128:
            if self._boxes_dev is None or self._boxes_dev.device != dev:
                self._boxes_dev = torch.empty(128 * 4, dtype=torch.int32, device=dev)
            aw.plane_boxes, aw.boxes_scratch = boxes.ctypes.data, self._boxes_dev.data_ptr()
        if self.quad_keep is not None:
            n = int(L.lib().vl3d_render_bwd_adam_class_bytes(desc))
            if self._class_dev is None or self._class_dev.numel() < n or self._class_dev.device != dev:
                self._class_dev = None
                self._class_dev = torch.empty(n, dtype=torch.uint8, device=dev)
            sgn = -1 if self.tile is not None else 1
            aw.quad_keep, aw.QH, aw.QW = self.quad_keep.data_ptr(), sgn * self.quad_keep.shape[1], sgn * self.quad_keep.shape[2]
            aw.quad_dyn = None if self.quad_dyn is None else self.quad_dyn.data_ptr()
            aw.class_scratch = self._class_dev.data_ptr()
            if self.layout is not None:
                aw.blocks = self.layout.blocks.data_ptr()
        # the compact gradient buffer: static texels of a tile-culled model (summed over the frames by the step kernel behind the backward) and
        # everything when the device-side plan finds the view infeasible; untouched otherwise (an allocation, no traffic)
        if not self._fused_ack:
            self._fused_ack = True
            warnings.warn("WindowAdam(fused_backward=True): loss.backward() APPLIES the Adam update of the rendered window (the optimiser step "
                          "runs inside the render backward, vl3d_render_bwd_adam); optimizer.step() afterwards is housekeeping only, and skipping "
                          "it (NaN guards, GradScaler) does NOT skip the update.  Pass fused_adam_backward=False to get_optimizer's args for the "
                          "two-kernel path, or call optimizer.acknowledge_fused_backward() to silence this.", RuntimeWarning, stacklevel=2)
        # (buffers of the call kept on the optimiser, grown on demand like the compact window: no allocator traffic inside autograd)
        if self._gfb is None or self._gfb.numel() < stack.numel() or self._gfb.device != dev:
            self._gfb = None
            self._gfb = torch.empty(stack.numel() + stack.numel() // 8, dtype=torch.float32, device=dev)
        g_fallback = self._gfb[:stack.numel()].view(stack.shape)
        with torch.cuda.device(dev):
            nscratch = max(int(L.lib().vl3d_render_bwd_scratch_bytes(desc)), 64)
            if self._bwd_scratch is None or self._bwd_scratch.numel() * 4 < nscratch or self._bwd_scratch.device != dev:
                self._bwd_scratch = None
                self._bwd_scratch = torch.empty((nscratch + nscratch // 8 + 3) // 4, dtype=torch.float32, device=dev)
            scratch = self._bwd_scratch
            L.check(L.lib().vl3d_render_bwd_adam(desc, L.ptr(stack), L.ptr(homos), L.ptr(rgb), L.ptr(alpha), L.ptr(g_rgb), L.ptr(g_alpha),
                                                 L.ptr(g_reg), L.ptr(reg_state), L.ptr(g_asum), L.ptr(g_fallback), L.ptr(scratch), nscratch,
                                                 C.byref(aw), L.stream_ptr(dev)), "vl3d_render_bwd_adam")
        self.t = t
        self.fused_steps += 1
        self.pending = ("stepped", pend[1])              # the leaf stays referenced: step() / the next window_leaf() check that nothing else reached it
        return scratch

    def reserve(self, window_texels):
        """Size the persistent window buffers (compact copy of a lean model, fallback gradient of the fused backward) for a window of
        `window_texels` = wh x ww texels per plane and frame, once, before the training loop: every later growth is a multi-GB hipMalloc in the middle
        of an epoch (docs/measurement_log.md, round 6).  The drivers call it per pyramid level with the largest window of the level's crops."""
        p = self.p
        D, T = self.dims()[:2]
        n = int(D) * int(T) * int(window_texels) * 4
        n += n // 8
        if self.lean_window and (self._compact_buf is None or self._compact_buf.numel() < n or self._compact_buf.device != p.device):
            self._compact_buf = None
            self._compact_buf = torch.zeros(n, dtype=p.dtype, device=p.device)
        if self.fused_backward and (self._gfb is None or self._gfb.numel() < n or self._gfb.device != p.device):
            self._gfb = None
            self._gfb = torch.empty(n, dtype=torch.float32, device=p.device)

    def _register_step(self, st, lr, b1, b2):
        """the scalars of the step about to be taken into the history table -> its number."""
        t = self.t + 1
        if t >= st["hist"].shape[0]:
            st["hist"] = torch.cat([st["hist"], torch.zeros_like(st["hist"])])
        # (row t itself is written by the step's own launch -- vl3d_adam_window_step* / vl3d_render_bwd_adam put (lr / bc1, sqrt(bc2)) of the step they
        # take into hist[step], include/vl3d.h: the 8-byte fill the host issued here was one launch per iteration)
        return t

    def _bound_deferral(self, st, t):
        # every `every` steps, tiles that have missed max_defer - every steps or more are brought up to date: nothing is ever older than
        # max_defer when its window comes back, and the sweep over the step table (a wave per tile) is paid on one step in `every`
        p = self.p
        D, T, Hs, Ws = self.dims()
        grp = self.param_groups[0]
        b1, b2 = grp["betas"]
        qk, qd, QH, QW = self._quads()
        every = max(1, min(8, self.max_defer // 4))
        if self.max_defer > 0 and t >= self.max_defer - every and t % every == 0:
            with torch.cuda.device(p.device):
                L.check(L.lib().vl3d_adam_flush_older(D, T, Hs, Ws, L.ptr(p), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), L.ptr(st["last_step"]),
                                                      L.ptr(st["hist"]), t, max(1, self.max_defer - every), float(b1), float(b2), float(grp["eps"]),
                                                      qk, qd, QH, QW, self._blocks(), L.stream_ptr(p.device)), "vl3d_adam_flush_older")

    # ---- the step -------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        st, p = self._st(), self.p
        grp = self.param_groups[0]
        b1, b2 = grp["betas"]
        lr, eps = float(grp["lr"]), float(grp["eps"])
        pending, self.pending = self.pending, None
        if self._stepped(pending):                        # the backward took the step (fused_backward): housekeeping only
            self._check_stepped_leaf(pending)
            self._bound_deferral(st, self.t)
            return loss
        if pending == "multiple":
            raise RuntimeError("WindowAdam: two windowed forwards before one step(); accumulate through the dense path (model.stack.grad) instead")
        dense = pending is None
        if dense and p.grad is None:
            return loss                                   # nothing flowed: the step does not count (like torch.optim.Adam)
        if not dense and pending[1].grad is None:
            if p.grad is None:
                return loss
            dense = True                                   # the graph went around the window leaf
        t = self._register_step(st, lr, b1, b2)
        D, T, Hs, Ws = self.dims()
        if dense and self.layout is not None:
            raise RuntimeError("WindowAdam: a packed model trains through its window leaf only (no dense gradient of the pool exists)")
        if dense:        # (the step kernel replays what is outstanding itself: no flush needed first)
            window, g, boxes = (0, 0, Hs, Ws), (p.grad if p.grad.is_contiguous() else p.grad.contiguous()), None
        else:
            window, g, boxes = pending[0], pending[1].grad, pending[2]
            if g.dtype != torch.float32 or tuple(g.shape) != (D, T, window[2], window[3], 4) or not g.is_contiguous():
                raise RuntimeError(f"WindowAdam: the window leaf's gradient must be contiguous float32 {(D, T, window[2], window[3], 4)}, "
                                   f"got {g.dtype} {tuple(g.shape)}")
            if p.grad is not None:
                raise RuntimeError("WindowAdam: both the window leaf and the dense parameter received a gradient in one step")
        y0, x0, wh, ww = window
        qk, qd, QH, QW = self._quads()
        with torch.cuda.device(p.device):
            L.check(L.lib().vl3d_adam_window_step_boxes(D, T, Hs, Ws, y0, x0, wh, ww, L.ptr(p), L.ptr(g), L.ptr(st["exp_avg"]),
                                                        L.ptr(st["exp_avg_sq"]), L.ptr(st["last_step"]), L.ptr(st["hist"]), lr, float(b1),
                                                        float(b2), eps, t, qk, qd, QH, QW,
                                                        1 if (dense and qk is not None) else 0,  # a dense p.grad of a sparsified model went through the tie hook
                                                        None if boxes is None else boxes.ctypes.data, self._blocks(), L.stream_ptr(p.device)),
                    "vl3d_adam_window_step")
        self.t = t
        self._bound_deferral(st, t)
        return loss


class Stage1Adam:
    """The optimiser `MPMesh.get_optimizer()` hands to train_3d.py (:159, 284-301): torch.optim.Adam over all parameters in ONE group
    (MPI.py:122-141: the planar path has no vertex group), as two engines -- the crop-aware `WindowAdam` for the plane stack (a stage-1
    iteration renders one 180 x 320 crop of one view, train_3d.py:20-95: about a third of every plane), `tiles.TileAdam` (one pass per
    parameter) for whatever else the model trains (the loop-mask texture).  The driver sees `param_groups[0]` (learning rate set per
    iteration, train_3d.py:303-310), zero_grad(), step(), state_dict()."""

    def __init__(self, stack_param, other_params, lr, betas=(0.9, 0.999), eps=1e-8, quad_keep=None, culled_alpha=-1e4, fused_backward=True, tile=None):
        from .tiles import TileAdam
        self.window = WindowAdam([stack_param], lr=lr, betas=betas, eps=eps, quad_keep=quad_keep, culled_alpha=culled_alpha,
                                 fused_backward=fused_backward, tile=tile)
        other_params = list(other_params)
        self.other = TileAdam([{'params': other_params}], lr=lr, betas=betas, eps=eps) if other_params else None
        self.param_groups = [dict(params=[stack_param] + other_params, lr=lr, betas=betas, eps=eps)]

    def _engines(self):
        return [self.window] + ([self.other] if self.other is not None else [])

    def sync(self):
        """the driver's learning rate into both engines (called before the forward: a fused backward reads it there)."""
        for o in self._engines():
            for g in o.param_groups:
                g["lr"] = self.param_groups[0]["lr"]

    def window_leaf(self, window, plane_boxes=None):
        self.sync()
        return self.window.window_leaf(window, plane_boxes)

    def flush(self):
        self.window.flush()

    def acknowledge_fused_backward(self):
        self.window.acknowledge_fused_backward()

    def zero_grad(self, set_to_none=True):
        for o in self._engines():
            o.zero_grad(set_to_none)

    def step(self, closure=None):
        self.sync()
        loss = self.window.step(closure)
        if self.other is not None:
            self.other.step()
        return loss

    def state_dict(self):
        return {"window": self.window.state_dict(), "other": None if self.other is None else self.other.state_dict(),
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.window.load_state_dict(sd["window"])
        if self.other is not None and sd.get("other") is not None:
            self.other.load_state_dict(sd["other"])
        self.param_groups[0].update(sd["param_groups"][0])
