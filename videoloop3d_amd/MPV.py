"""MPMeshVid on the MI355X-native render + looping-loss kernels: drop-in for the hot path of the reference's MPV.py.

Mirrors /root/reference/MPV.py:26-138 (constructor), :351-475 (render) and :477-556 (forward) for PLANAR, un-deformed
geometry -- which is all the shipped configs ever produce (vertices get no gradient: MPV.py:354; optimize_geo_start is
"currently not used": config_parser.py:153-154).  Differences by design:
  * the learnable texture is the dense plane stack `stack` (D,T,mpi_h,mpi_w,4) (one 16-byte rgba texel, coalesced HBM
    reads) instead of the (T,4,Ah,Aw) atlas grid of cells of MPV.py:75-104; `atlas_to_stack`/`stack_to_atlas` convert.
    Texel pitch is exactly 1 plane pixel (the atlas pitch of (Aw-1)/(gw*(mpi_w-1)) is available through `texel_scale`).
  * no rasteriser: coverage and UVs of fronto-parallel quads are the analytic per-plane homography, with
    the two pytorch3d-side constants exposed as `pixel_center` (0.5) and hard-cut borders.
  * lod / get_optimizer / get_lrate / update_step (the stage-2 driver's hooks, train_3dvid.py:264-281) act on the stack;
  * init_from_mpi takes the state_dict of videoloop3d_amd.MPI.MPMesh / MPMeshVid (dense stack + culled/static/dynamic quad
    maps, videoloop3d_amd/tiles.py) AND the reference's own checkpoints (plane meshes + packed atlases: their tiles are
    resampled onto the dense stack, tiles.stack_from_reference_state); static quads stay one shared texture because their
    gradient is summed over the frames.  reference_state_dict / save_mesh / save_texture export back to the reference's layout
    (videoloop3d_amd/export.py).
"""
import dataclasses

import numpy as np
import torch
import torch.nn as nn

from .render import RenderSpec, render_planes, render_planes_with_regularisers
from .utils_mpi import compute_homography, make_depths, overcompose
from .utils_vid import Patch3DAvg, Patch3DGPNNDirectLoss, Patch3DGPNNLowMemLoss, Patch3DMSE

# activations the HIP kernels implement (subset of MPI.py:21-31; shipped configs use sigmoid/sigmoid)
ACTIVATES = {'relu': torch.relu, 'sigmoid': torch.sigmoid, 'none': lambda x: x,
             'clamp': lambda x: torch.clamp(x, 0, 1), 'abs': torch.abs}


def get_new_intrin(old_intrin, new_h_start, new_w_start):
    """utils.py:196-200."""
    new_intrin = old_intrin.clone() if isinstance(old_intrin, torch.Tensor) else old_intrin.copy()
    new_intrin[..., 0, 2] -= new_w_start
    new_intrin[..., 1, 2] -= new_h_start
    return new_intrin


def sparsity_ratio(alpha_sums, eps):
    """per-pixel |a|_1 / max(|a|_2, eps) over the planes from the fused sums (sum a, sum a^2) [..., 2]
    (alphas are >= 0 for the supported activations, so |a|_1 = sum a)."""
    n1, n2s = alpha_sums[..., 0], alpha_sums[..., 1]
    n2 = n2s.clamp_min(1e-30).sqrt()          # keeps sqrt's gradient finite where no plane covers the pixel
    return n1 / n2.clamp_min(eps)


def _resize_tiles(tl, h, w, antialias):
    """[N,C,th,tw] -> [N,C,h,w], every tile on its own, bilinear with align_corners=False (torchvision's Resize of a tensor, MPV.py:157-162).
    Without antialiasing this is F.interpolate's arithmetic spelt with gathers: torch's plain bilinear kernel parallelises over the OUTPUT pixels
    only and loops over N x C inside a thread -- 0.2 s per plane for the 110 000 twelve-texel tiles of a stage-2 plane, 6 s per `lod` call -- where
    the antialiased kernel (0.2 ms) does not.  Same source index, same two taps per axis, same weights (upsample_bilinear2d's
    `area_pixel_compute_source_index`); the sum is formed in its order, so the values agree to the rounding of a fused multiply-add."""
    if antialias:
        return torch.nn.functional.interpolate(tl, size=(h, w), mode="bilinear", align_corners=False, antialias=True)

    def axis(n_in, n_out):
        src = ((torch.arange(n_out, dtype=torch.float32, device=tl.device) + 0.5) * (float(n_in) / float(n_out)) - 0.5).clamp_min(0.0)
        i0 = src.floor().clamp_max(n_in - 1)
        l1 = src - i0
        i0 = i0.long()
        return i0, (i0 + 1).clamp_max(n_in - 1), 1.0 - l1, l1
    y0, y1, ly0, ly1 = axis(tl.shape[2], h)
    x0, x1, lx0, lx1 = axis(tl.shape[3], w)
    r0, r1 = tl.index_select(2, y0), tl.index_select(2, y1)
    top = lx0 * r0.index_select(3, x0) + lx1 * r0.index_select(3, x1)
    bot = lx0 * r1.index_select(3, x0) + lx1 * r1.index_select(3, x1)
    return ly0[:, None] * top + ly1[:, None] * bot


class _LoopPrologue(torch.autograd.Function):
    """MPV.py:484-507 between the render and the loss in three launches (csrc/vl3d_loss.hip `loop_*_k`): loop padding
    `cat(rgb, rgb[:pad])`, the scale-invariant gain `(exp(mean(log((mean_f res + .01) / (mean_t rgb.detach() + .01)))) + 3) / 4` and the
    layout the loss takes, [1,3,T+pad,h,w]; the backward folds the pad frames' gradient back and writes the NHWC gradient the render's
    backward reads.  rgb [T,h,w,3] contiguous float32 CUDA; res [F,3,h,w] or None (no gain)."""

    @staticmethod
    def forward(ctx, rgb, res, pad, want_gram=False):
        ctx.set_materialize_grads(False)      # outputs the loss does not use come back as None, not as zero-filled tensors (a fill each, and reads in the backward kernels)
        from . import _lib as L
        L.check_cuda(rgb)
        T, h, w, _ = rgb.shape
        rgb = rgb.contiguous()
        dev = rgb.device
        log_sum = None
        with torch.cuda.device(dev):
            if res is not None:
                L.check_cuda(res)
                if res.dim() != 4 or res.shape[1:] != (3, h, w):
                    raise RuntimeError(f"scale-invariant gain: res must be [F,3,{h},{w}], got {tuple(res.shape)}")
                # (a crop of the captured clip goes in through its strides: the copy was 23 us of a 1.7 ms tile-culled iteration)
                res = res.detach() if (res.dtype == torch.float32 and res.stride(3) == 1) else res.detach().to(torch.float32).contiguous()
                log_sum = torch.empty(1, dtype=torch.float64, device=dev)
                L.check(L.lib().vl3d_loop_gain_strided(T, res.shape[0], h, w, L.ptr(rgb), L.ptr(res), res.stride(0), res.stride(1), res.stride(2),
                                                       L.ptr(log_sum), L.stream_ptr(dev)), "vl3d_loop_gain")
            x = torch.empty((1, 3, T + pad, h, w), dtype=torch.float32, device=dev)
            if want_gram:      # x's form for the NN search in the same pass over the render's output (utils_vid.PreparedX)
                xg = torch.empty(int(L.lib().vl3d_gram_major_bytes(T + pad, h, w)) // 4, dtype=torch.float32, device=dev)
                L.check(L.lib().vl3d_loop_pad_fwd_gram(T, pad, h, w, L.ptr(rgb), L.ptr(log_sum), L.ptr(x), L.ptr(xg), L.stream_ptr(dev)), "vl3d_loop_pad_fwd_gram")
            else:
                xg = torch.empty(0, dtype=torch.float32, device=dev)
                L.check(L.lib().vl3d_loop_pad_fwd(T, pad, h, w, L.ptr(rgb), L.ptr(log_sum), L.ptr(x), L.stream_ptr(dev)), "vl3d_loop_pad_fwd")
        ctx.log_sum, ctx.dims = log_sum, (T, pad, h, w)
        ctx.mark_non_differentiable(xg)
        return x, xg

    @staticmethod
    def backward(ctx, gx, _gxg=None):
        from . import _lib as L
        T, pad, h, w = ctx.dims
        if gx is None:
            return None, None, None, None
        gx = gx[0]
        if gx.stride(3) != 1 or gx.stride(2) != w:
            gx = gx.contiguous()
        dev = gx.device
        g_rgb = torch.empty((T, h, w, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().vl3d_loop_pad_bwd(T, pad, h, w, L.ptr(gx), gx.stride(0), gx.stride(1), L.ptr(ctx.log_sum), L.ptr(g_rgb),
                                              L.stream_ptr(dev)), "vl3d_loop_pad_bwd")
        return g_rgb, None, None, None


class _PixelTerms(torch.autograd.Function):
    """(mean_p |a|_1 / max(|a|_2, eps), mean_p |alpha - 1|) from the render's per-pixel outputs in one launch each way (vl3d_pixel_terms):
    the sparsity and density regularisers of MPI.py:599-603, 647-650 / MPV.py:511-515, 533-536 were ~30 scalar torch launches per
    iteration around a 0.2 ms render.  alpha [..] or None, alpha_sums [..,2] or None -> float32 [2] (0 for an absent input)."""

    @staticmethod
    def forward(ctx, alpha, alpha_sums, eps):
        ctx.set_materialize_grads(False)      # outputs the loss does not use come back as None, not as zero-filled tensors (a fill each, and reads in the backward kernels)
        from . import _lib as L
        ref = alpha if alpha is not None else alpha_sums
        L.check_cuda(ref)
        dev = ref.device
        a = None if alpha is None else alpha.detach().to(torch.float32).contiguous()
        s = None if alpha_sums is None else alpha_sums.detach().to(torch.float32).contiguous()
        n = a.numel() if a is not None else s.numel() // 2
        if a is not None and s is not None and s.numel() != 2 * n:
            raise RuntimeError(f"alpha {tuple(alpha.shape)} and alpha_sums {tuple(alpha_sums.shape)} do not describe the same pixels")
        sums = torch.empty(2, dtype=torch.float64, device=dev)
        need_a = a is not None and alpha.requires_grad
        need_s = s is not None and alpha_sums.requires_grad
        ga = torch.empty_like(a) if need_a else None
        gs = torch.empty_like(s) if need_s else None
        with torch.cuda.device(dev):
            L.check(L.lib().vl3d_pixel_terms(n, L.ptr(a), L.ptr(s), float(eps), L.ptr(sums), L.ptr(gs), L.ptr(ga), L.stream_ptr(dev)), "vl3d_pixel_terms")
        ctx.ga, ctx.gs, ctx.n = ga, gs, n
        # TWO outputs (0-d views of one tensor), not one tensor the caller slices: a slice's backward is a zero fill, a copy and an add per term
        return tuple((sums / n).to(torch.float32).unbind(0))

    @staticmethod
    def backward(ctx, g_sparsity, g_density):
        ga = None if (ctx.ga is None or g_density is None) else ctx.ga * (g_density / ctx.n)
        gs = None if (ctx.gs is None or g_sparsity is None) else ctx.gs * (g_sparsity / ctx.n)
        return ga, gs, None


class _SmoothTerms(torch.autograd.Function):
    """rgb_smooth / a_smooth from the four fused sums (MPV.py:517-531): (c0 s0 + c1 s1, c2 s2 + c3 s3) in two launches each way instead of
    ~16 scalar kernels.  coef: device float32 [4]."""

    @staticmethod
    def forward(ctx, sums, coef):
        ctx.set_materialize_grads(False)      # outputs the loss does not use come back as None, not as zero-filled tensors (a fill each, and reads in the backward kernels)
        ctx.save_for_backward(coef)
        return tuple((sums * coef).view(2, 2).sum(1).unbind(0))      # (rgb_smooth, a_smooth): two outputs, see _PixelTerms

    @staticmethod
    def backward(ctx, g_rgb, g_a):
        (coef,) = ctx.saved_tensors
        z = coef.new_zeros(())
        g = torch.stack([z if g_rgb is None else g_rgb.reshape(()), z if g_a is None else g_a.reshape(())])
        return (coef.view(2, 2) * g[:, None]).reshape(4), None


class _LinearHead(torch.autograd.Function):
    """total = sum_i coef_i v_i over the device scalars v = (main, rest[0..n-2]) -> (total, groups [ngroups]) in one launch (vl3d_linear_head_fwd),
    coef * g in one more on the way back: the weighted total of train_3dvid.py:230-240 without a stack, a multiply and a sum on one-element
    tensors and their autograd mirrors (~14 launches of a stage-2 iteration that is bound by the GPU's timeline)."""

    @staticmethod
    def forward(ctx, main, rest, coef, groups, ngroups):
        ctx.set_materialize_grads(False)      # outputs the loss does not use come back as None, not as zero-filled tensors (a fill each, and reads in the backward kernels)
        from . import _lib as L
        L.check_cuda(main, coef)
        n = 1 + (0 if rest is None else rest.numel())
        m = main.detach().reshape(1)
        m = m if m.dtype == torch.float32 else m.to(torch.float32)
        r = None if rest is None else rest.detach().to(torch.float32).contiguous()
        out = torch.empty(1 + ngroups, dtype=torch.float32, device=main.device)
        with torch.cuda.device(main.device):
            L.check(L.lib().vl3d_linear_head_fwd(n, int(groups), int(ngroups), L.ptr(m), L.ptr(r), L.ptr(coef), L.ptr(out), L.stream_ptr(main.device)),
                    "vl3d_linear_head_fwd")
        ctx.coef, ctx.n, ctx.main_shape = coef, n, tuple(main.shape)
        parts = out[1:]
        ctx.mark_non_differentiable(parts)
        return out[0], parts

    @staticmethod
    def backward(ctx, g, _gp):
        from . import _lib as L
        if g is None:
            return None, None, None, None, None
        gs = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.to(torch.float32).contiguous()
        gt = torch.empty(ctx.n, dtype=torch.float32, device=gs.device)
        with torch.cuda.device(gs.device):
            L.check(L.lib().vl3d_linear_head_bwd(ctx.n, L.ptr(ctx.coef), L.ptr(gs), L.ptr(gt), L.stream_ptr(gs.device)), "vl3d_linear_head_bwd")
        return gt[0].view(ctx.main_shape), (gt[1:] if ctx.n > 1 else None), None, None, None


def atlas_to_stack(atlas_dyn, mpi_d, grid_h):
    """(T,4,Ah,Aw) atlas of grid_h x grid_w plane cells (MPV.py:37-44,75-81: plane p <-> cell (p // grid_w, p % grid_w))
    -> (D,T,mpi_h,mpi_w,4) stack."""
    T, C, Ah, Aw = atlas_dyn.shape
    grid_w = mpi_d // grid_h
    mh, mw = Ah // grid_h, Aw // grid_w
    cells = atlas_dyn.reshape(T, C, grid_h, mh, grid_w, mw).permute(2, 4, 0, 3, 5, 1)     # gh,gw,T,mh,mw,C
    return cells.reshape(mpi_d, T, mh, mw, C).contiguous()


def stack_to_atlas(stack, grid_h):
    D, T, mh, mw, C = stack.shape
    grid_w = D // grid_h
    cells = stack.reshape(grid_h, grid_w, T, mh, mw, C).permute(2, 5, 0, 3, 1, 4)          # T,C,gh,mh,gw,mw
    return cells.reshape(T, C, grid_h * mh, grid_w * mw).contiguous()


class MPMeshVid(nn.Module):
    def __init__(self, args, H, W, ref_extrin, ref_intrin, near, far, pixel_center=0.5, texel_scale=(1.0, 1.0), atlas_exact=False, device=None):
        """device: where the plane stack is CREATED (an addition to the reference's signature: its `torch.randn` on the host followed by `.to(device)`
        is 7 GB and ~7 s for the shipped stage-2 shape -- D = 32, T = 50, 396 x 704 planes -- before init_from_mpi overwrites it; the small camera
        buffers still follow `.to()`).
        atlas_exact=True: sample the stack exactly like the reference samples its atlas of plane cells (MPV.py:75-81, 394-439:
        pitch (Aw-1)/(gw*(mpi_w-1)), per-cell sub-texel origin, neighbour-cell bleed at cell edges) -- videoloop3d_amd/atlas.py.
        A parity mode for weights that come from / go to the reference (atlas_to_stack / stack_to_atlas); needs args.atlas_grid_h."""
        super().__init__()
        self.atlas_exact = bool(atlas_exact)
        self.per_plane_boxes = True      # crop-aware optimiser: each plane's own texel box inside the crop's window (optim.WindowAdam); False = the union window for every plane
        self.atlas_grid_h = int(getattr(args, "atlas_grid_h", 1))
        if self.atlas_exact and args.mpi_d % self.atlas_grid_h != 0:
            raise RuntimeError("mpi_d and atlas_grid_h should match")                                    # MPV.py:38
        self.args = args
        self.frm_num = args.mpv_frm_num
        self.isloop = args.mpv_isloop
        mpi_h, mpi_w = int(args.mpi_h_scale * H), int(args.mpi_w_scale * W)
        self.mpi_h, self.mpi_w = mpi_h, mpi_w
        self.mpi_d, self.near, self.far = args.mpi_d, near, far
        self.H, self.W = H, W
        if getattr(args, "fp16", False):
            raise RuntimeError("fp16 is marked 'do NOT use' in the reference (config_parser.py:32-33); fp32 only")
        if getattr(args, "rgb_mlp_type", "direct") != "direct":
            raise RuntimeError(f"rgbmlp_type = {args.rgb_mlp_type} not supported (shipped configs use 'direct', mpv_base.txt:28)")
        ref_extrin, ref_intrin = np.asarray(ref_extrin), np.asarray(ref_intrin)
        assert ref_extrin.shape == (4, 4) and ref_intrin.shape == (3, 3)
        self.register_buffer("ref_extrin", torch.tensor(ref_extrin))
        self.register_buffer("ref_intrin", torch.tensor(ref_intrin).float())
        self.register_buffer("planedepth", make_depths(self.mpi_d, near, far).float().flip(0))   # plane 0 = nearest (MPV.py:51)
        # intrinsics that map the whole (larger) MPI plane to plane pixels (MPV.py:55-56)
        self.H_start, self.W_start = (mpi_h - H) // 2, (mpi_w - W) // 2
        self.register_buffer("ref_intrin_mpi", get_new_intrin(self.ref_intrin, -self.H_start, -self.W_start))

        stack = torch.randn((self.mpi_d, self.frm_num, mpi_h, mpi_w, 4), device=device) * args.init_std          # MPV.py:84-85
        stack[..., -1] = -2                                                                       # MPV.py:109-110
        self.stack = nn.Parameter(stack, requires_grad=True)

        if args.rgb_activate not in ACTIVATES or args.alpha_activate not in ACTIVATES:
            raise RuntimeError(f"activation ({args.rgb_activate}, {args.alpha_activate}) not implemented by the HIP kernels")
        self.rgb_activate, self.alpha_activate = ACTIVATES[args.rgb_activate], ACTIVATES[args.alpha_activate]
        self.texel_scale = tuple(float(v) for v in texel_scale)
        self.spec = dataclasses.replace(RenderSpec.mpv(rgb_act=args.rgb_activate, alpha_act=args.alpha_activate,
                                                       scale=self.texel_scale), pixel_center=float(pixel_center))
        self.optimize_geometry = False
        self.is_sparse, self.has_dyn = False, False
        self.register_buffer("quad_keep", None)      # [D,QH,QW] bool maps of a sparsified stage-1 MPI (init_from_mpi)
        self.register_buffer("quad_dyn", None)
        self._tie_hook = None
        self._static_compact = False
        self._window_opt = None          # the crop-aware Adam handed out by get_optimizer (dense CUDA models)
        self.packed = None               # packed.PackedLayout once pack_() has replaced the dense stack by the pool `stack_pool`
        self.tile_full = None            # (th, tw): texels per quad of a model loaded from a sparsified REFERENCE checkpoint (tile lattice)
        self.tile_own = None             # (th, tw) at the current pyramid level when every quad owns its border texels (TILE-EXACT layout, the
                                         # default for sparsified reference checkpoints: init_from_mpi); None: neighbouring quads share them

        self.swd_patch_size, self.swd_patcht_size = args.swd_patch_size, args.swd_patcht_size
        self.swd_stride, self.swd_stridet = args.swd_stride, args.swd_stridet
        self.losses = {                                   # MPV.py:131-138 ('swd' is None there too; 'gpnn_down' is off-path)
            'swd': None,
            'gpnn': Patch3DGPNNDirectLoss(),
            'gpnn_lm': Patch3DGPNNLowMemLoss(),
            'mse': Patch3DMSE,
            'avg': Patch3DAvg,
        }

    # ---- packed storage of a tile-culled model (videoloop3d_amd/packed.py) -------------------------------------------------
    def _apply(self, fn, *a, **k):
        """module.to() / .cuda() / .cpu(): the block table of a packed model is plain state (not a buffer: its entries are addresses into
        the pool, meaningless to load_state_dict) and moves with the pool."""
        out = super()._apply(fn, *a, **k)
        if self.packed is not None and self.packed.blocks.device != self.stack_pool.device:
            self.packed.to(self.stack_pool.device)
            self._window_opt = None          # optimiser state lives on the old device: the driver asks for a new one
        return out

    def _param(self):
        """the texture parameter: the dense stack, or the pool of a packed model."""
        return self.stack_pool if self.packed is not None else self.stack

    def stack_dims(self):
        """(D, T, Hs, Ws) of the plane stack, whichever way it is stored."""
        if self.packed is not None:
            return self.packed.D, self.packed.T, self.packed.Hs, self.packed.Ws
        return tuple(self.stack.shape[:4])

    def stack_plane(self, d, frames=None):
        """(T' ,Hs,Ws,4): plane d of the dense stack (all frames or `frames`), current (deferred updates flushed by the caller)."""
        if self.packed is not None:
            return self.packed.unpack_plane(self.stack_pool.data, d, frames)
        return self.stack.data[d] if frames is None else self.stack.data[d, torch.as_tensor(frames, device=self.stack.device).long()]

    def _set_texture_geometry(self, hs, ws):
        """the render spec of a texture of hs x ws texels per plane: the planes keep their extent (MPV.py:75-81: normalised UVs), so the
        plane-pixel -> texel scale follows the texture size; in the tile-exact layout the scale gives the LATTICE coordinate (a quad spans
        tile - 1 of them) and the spec carries the tile size (render.RenderSpec.tile)."""
        if self.tile_own is not None:
            qh, qw = int(self.args.mpi_h_verts) - 1, int(self.args.mpi_w_verts) - 1
            th, tw = hs // qh, ws // qw
            if (qh * th, qw * tw) != (hs, ws) or min(th, tw) < 2:
                raise RuntimeError(f"tile-exact layout: planes of {(hs, ws)} texels are not {qh} x {qw} whole tiles")
            self.tile_own = (th, tw)
            self.spec = dataclasses.replace(self.spec, tile=(th, tw),
                                            scale=(self.texel_scale[0] * (qw * (tw - 1)) / max(self.mpi_w - 1, 1),
                                                   self.texel_scale[1] * (qh * (th - 1)) / max(self.mpi_h - 1, 1)))
            return
        self.spec = dataclasses.replace(self.spec, tile=(0, 0), scale=(self.texel_scale[0] * (ws - 1) / max(self.mpi_w - 1, 1),
                                                                       self.texel_scale[1] * (hs - 1) / max(self.mpi_h - 1, 1)))

    @torch.no_grad()
    def pack_(self, stack=None):
        """Replace the dense stack of a SPARSIFIED model by the packed pool: static blocks stored once, dynamic blocks per frame, culled
        blocks not at all (the reference's static / dynamic atlases, MPI.py:364-436, MPV.py:235-288) -- about a seventh of the dense
        bytes at 16 % kept quads.  Training (window path of the crop-aware Adam), evaluation renders of chosen frames, lod(), state_dict()
        and the export to the reference's layout work on the pool; results have the dense model's bits.
        stack (optional): a dense (D,T,Hs,Ws,4) tensor on ANY device (e.g. a reference checkpoint resampled on the host) to pack instead
        of self.stack -- it never has to exist on the GPU."""
        from .packed import PackedLayout
        if not (self.is_sparse and self.quad_keep is not None):
            raise RuntimeError("pack_() needs the quad maps of a sparsified model (init_from_mpi of a sparsified MPI)")
        if self.atlas_exact:
            raise RuntimeError("atlas_exact renders the dense stack")
        self._flush_deferred_updates()
        self._window_opt = None
        dev = self._param().device
        src = self.stack.data if stack is None else stack
        lay = PackedLayout(self.quad_keep.to(dev), self.quad_dyn.to(dev), src.shape[1], src.shape[2], src.shape[3], self.tile_own)
        pool = lay.new_pool(dev)
        for d in range(lay.D):
            lay.pack_plane_(pool, d, src[d])
        if self._tie_hook is not None:
            self._tie_hook.remove()
            self._tie_hook = None
        if "stack" in self._parameters:
            del self._parameters["stack"]
        if "stack_pool" in self._parameters:
            del self._parameters["stack_pool"]
        self.register_parameter("stack_pool", nn.Parameter(pool, requires_grad=True))
        self.packed = lay
        self.frm_num = lay.T
        return self

    # ---- stage-1 -> stage-2 hand-over (MPV.py:235-304) ------------------------------------------------------------------
    def init_from_mpi(self, state_dict, packed=False, tile_layout=None):
        """tile_layout (sparsified REFERENCE checkpoints; default args.tile_layout or "exact"): "exact" = every quad keeps its tile with its OWN
        border row / column, the reference's representation (MPI.py:380-418) -- identical weights for ANY checkpoint, trained ones included
        (golden G19), and the reference's training trajectory (the two copies of a border sample are separate parameters); "lattice" = the
        shared-border stack of rounds 4-5 (exact for FRESH checkpoints only; 6-12 % fewer texels).
        packed=True (sparsified checkpoints): go straight to the packed pool (pack_()); the checkpoint's dense form stays where the
        caller put it (the host) and never exists on the GPU.  Checkpoints of a packed model ('stack_pool') are loaded as such.
        MPV.py:235-288 for the dense representation: take the stage-1 MPI (`MPMesh.state_dict()`) as the initial value of
        every frame.  With a sparsified MPI the quad maps come along: culled quads stay invisible, static quads stay ONE
        texture shared by all frames (their gradient is summed over the frames, as the reference's static atlas sees it),
        dynamic quads are free per frame; without them everything is dynamic ("load static as dynamic", MPV.py:266-288)."""
        if "stack" not in state_dict and ("atlas" in state_dict or "atlas_dyn" in state_dict):
            # a checkpoint of the REFERENCE (plane meshes + packed texture atlases, MPI.py:207-221 / MPV.py:290-304): resample its
            # tiles onto the dense stack and recover the culled / static / dynamic quad maps from its face lists
            from . import tiles
            hv, wv = int(self.args.mpi_h_verts), int(self.args.mpi_w_verts)
            layout = tile_layout if tile_layout is not None else getattr(self.args, "tile_layout", "exact")
            if layout not in ("exact", "lattice"):
                raise RuntimeError(f"tile_layout must be 'exact' or 'lattice', got {layout!r}")
            sparse = bool(state_dict.get("self.is_sparse", False))
            tile_ref = tiles.reference_tile_size(state_dict, hv, wv) if sparse else None      # (th, tw) of a checkpoint `sparsify_faces` wrote
            own = layout == "exact" and tile_ref is not None and not self.atlas_exact
            st, keep, dyn = tiles.stack_from_reference_state(state_dict, self.mpi_h, self.mpi_w, hv, wv, self.frm_num, own_borders=own)
            # a sparsified checkpoint arrives tile for tile (identical weights); the tile size drives lod() like the reference's (MPV.py:146-151)
            tile_own = tile_ref if own else None
            if own and "self.atlas_full_h" in state_dict and int(state_dict.get("self.atlas_grid_h", 0)) > 0 and int(state_dict.get("self.atlas_grid_w", 0)) > 0:
                # (a checkpoint saved at a pyramid level: the FULL tile size is what lod() scales, MPV.py:149-151)
                tile_ref_full = (int(state_dict["self.atlas_full_h"]) // int(state_dict["self.atlas_grid_h"]),
                                 int(state_dict["self.atlas_full_w"]) // int(state_dict["self.atlas_grid_w"]))
            else:
                tile_ref_full = tile_ref
            tile = tile_ref_full if own else (((st.shape[2] - 1) // (hv - 1) + 1, (st.shape[3] - 1) // (wv - 1) + 1)
                                         if tuple(st.shape[2:4]) != (self.mpi_h, self.mpi_w) else None)
            state_dict = {"ref_extrin": state_dict["ref_extrin"], "ref_intrin": state_dict["ref_intrin"],
                          "planedepth": state_dict["planedepth"], "stack": st, "quad_keep": keep, "quad_dyn": dyn,
                          "self.is_sparse": sparse, "self.has_dyn": sparse, "self.tile_full": tile, "self.tile_own": tile_own}
        self.ref_extrin.data = state_dict['ref_extrin'].type_as(self.ref_extrin)
        self.ref_intrin.data = state_dict['ref_intrin'].type_as(self.ref_intrin)
        self.planedepth.data = state_dict['planedepth'].type_as(self.planedepth)
        self.ref_intrin_mpi.data = get_new_intrin(self.ref_intrin, -self.H_start, -self.W_start)
        dev = self._param().device
        tf = state_dict.get("self.tile_full", None)
        self.tile_full = None if tf is None else (int(tf[0]), int(tf[1]))
        to = state_dict.get("self.tile_own", None)
        self.tile_own = None if to is None else (int(to[0]), int(to[1]))
        if "stack_pool" in state_dict:       # a checkpoint of a packed model of this package: quad maps + dims rebuild the block table
            from .packed import PackedLayout
            D, T, hs, ws = (int(v) for v in state_dict["self.packed_dims"])
            self.register_buffer("quad_keep", state_dict["quad_keep"].to(dev).bool())
            self.register_buffer("quad_dyn", state_dict["quad_dyn"].to(dev).bool())
            lay = PackedLayout(self.quad_keep, self.quad_dyn, T, hs, ws, self.tile_own)
            pool = state_dict["stack_pool"].to(dev, torch.float32).reshape(-1, 4).contiguous()
            if pool.shape[0] != lay.n_slots * 64:
                raise RuntimeError("packed checkpoint: the pool does not match the block table of its quad maps")
            for name in ("stack", "stack_pool"):
                self._parameters.pop(name, None)
            self.register_parameter("stack_pool", nn.Parameter(pool, requires_grad=True))
            self.packed, self.frm_num, self.is_sparse, self.has_dyn = lay, T, True, True
            self._window_opt = None
            self._set_texture_geometry(hs, ws)
            return
        mpi = state_dict['stack']
        if mpi.dim() != 5 or mpi.shape[0] != self.mpi_d or mpi.shape[-1] != 4:
            raise RuntimeError(f"checkpoint stack {tuple(mpi.shape)} does not match mpi_d={self.mpi_d}")
        if mpi.shape[1] not in (1, self.frm_num):        # a stage-2 checkpoint with another frame count (MPV.py:262-265)
            print(f"Warnining, inconsistent frame number detected, change from {self.frm_num} to {mpi.shape[1]}")
            self.frm_num = int(mpi.shape[1])
        if packed:
            if not bool(state_dict.get("self.is_sparse", False)):
                raise RuntimeError("init_from_mpi(packed=True) needs a sparsified checkpoint (quad maps)")
            self.is_sparse, self.has_dyn = True, bool(state_dict.get("self.has_dyn", False))
            self.register_buffer("quad_keep", state_dict["quad_keep"].to(dev).bool())
            self.register_buffer("quad_dyn", state_dict["quad_dyn"].to(dev).bool())
            hs, ws = mpi.shape[2:4]
            self._set_texture_geometry(hs, ws)
            self.pack_(stack=mpi.float().expand(-1, self.frm_num, -1, -1, -1))      # (a view: a static MPI is not copied T times)
            return
        if self.packed is not None:
            self._parameters.pop("stack_pool", None)
            self.packed = None
        with torch.no_grad():
            new = mpi.to(dev, torch.float32).expand(-1, self.frm_num, -1, -1, -1).contiguous()
        self.register_parameter("stack", nn.Parameter(new, requires_grad=True))
        # planes saved at a pyramid level (lod) keep their extent: the plane-pixel -> texel scale follows the texture size
        hs, ws = new.shape[2:4]
        self._set_texture_geometry(hs, ws)
        self.is_sparse = bool(state_dict.get("self.is_sparse", False))
        self.has_dyn = bool(state_dict.get("self.has_dyn", False))
        if self.is_sparse:
            self.register_buffer("quad_keep", state_dict["quad_keep"].to(self.stack.device).bool())
            self.register_buffer("quad_dyn", state_dict["quad_dyn"].to(self.stack.device).bool())
        else:
            self.register_buffer("quad_keep", None)
            self.register_buffer("quad_dyn", None)
        self._install_tie_hook()

    def _install_tie_hook(self):
        """(re-)attach the gradient hook that keeps static quads one shared texture; the parameter object changes in lod()."""
        if self._tie_hook is not None:
            self._tie_hook.remove()
            self._tie_hook = None
        if self.packed is not None:      # a packed model trains through the window leaf: static gradients are summed inside the step
            return
        if self.is_sparse and self.quad_keep is not None:
            from . import tiles
            if self.stack.is_cuda:    # the gradient comes from the culled render: culled texels already hold exact zeros
                # frame0_only once get_optimizer has handed out the tile-aware Adam, which reads static gradients from frame 0
                self._tie_hook = self.stack.register_hook(
                    lambda g: tiles.tie_static_grad_hip(g, self.quad_keep, self.quad_dyn, assume_culled_zero=True,
                                                        frame0_only=self._static_compact, tile=self.tile_own))
            else:                     # (CPU: host-logic tests only)
                self._tie_hook = self.stack.register_hook(lambda g: tiles.tie_static_grad(g, self.quad_keep, self.quad_dyn, self.tile_own))

    def state_dict(self, *args, **kwargs):
        """MPV.py:290-304: tensors + python scalars under "self.*" keys.  Loaded back with init_from_mpi(), like every driver of the reference
        does (train_3dvid.py:210, scripts/script_render_video.py:119): nn.Module.load_state_dict knows nothing of the "self.*" scalars, of a
        texture whose resolution changed in lod(), or of a packed model's block table."""
        self._flush_deferred_updates()
        sd = super().state_dict(*args, **kwargs)
        sd["self.is_sparse"] = self.is_sparse
        sd["self.has_dyn"] = self.has_dyn
        if self.tile_full is not None:
            sd["self.tile_full"] = self.tile_full
        if self.tile_own is not None:
            sd["self.tile_own"] = self.tile_own
        if self.packed is not None:
            sd["self.packed_dims"] = self.stack_dims()      # with quad_keep / quad_dyn this rebuilds the block table (init_from_mpi)
        return sd

    # ---- driver hooks (train_3dvid.py:264-281) ------------------------------------------------------------------------
    def lod(self, factor):
        """MPV.py:140-197 (dense branch): resample the learnable texture to `factor` of its full resolution for one level of
        the training pyramid.  Every plane of the stack is resized to (int(mpi_h*factor), int(mpi_w*factor)) with the
        bilinear filter torchvision's Resize applies to the atlas (`args.lod_antialias`, below); the plane quads keep their extent, so the
        plane-pixel -> texel scale of the render spec becomes (w'-1)/(mpi_w-1) (the reference gets this from its
        normalised UVs, MPV.py:75-81)."""
        self._flush_deferred_updates()
        self._window_opt = None          # the parameter object changes: the driver asks for a new optimiser (train_3dvid.py:264-265)
        # torchvision's Resize on a TENSOR: no antialiasing in the release the reference pins (requirements.txt: torch==1.10 -> torchvision 0.11:
        # `antialias=None` means False for tensors, transforms/functional_tensor.py), antialiased from torchvision 0.17 on (the default became
        # True).  Only a DOWN-sampling call differs -- the first lod() of a run, from the full-resolution initialisation to the coarsest level;
        # the later ones up-sample, where the two filters coincide.  Default: the pinned release's; `args.lod_antialias = True` for the newer one.
        aa = bool(getattr(self.args, "lod_antialias", False))
        h, w = max(int(self.mpi_h * factor), 2), max(int(self.mpi_w * factor), 2)
        if self.tile_full is not None:
            # a model on the reference's tile lattice: every quad holds max(int(tile * factor), 2) texels per axis at this level, as the
            # reference resizes its tiles (MPV.py:146-151); neighbouring quads share their border texels
            qh, qw = int(self.args.mpi_h_verts) - 1, int(self.args.mpi_w_verts) - 1
            h = qh * (max(int(self.tile_full[0] * factor), 2) - 1) + 1
            w = qw * (max(int(self.tile_full[1] * factor), 2) - 1) + 1
        D, T, hs, ws = self.stack_dims()
        if self.tile_own is not None:
            # tile-exact layout: the reference's own operation (MPV.py:146-163) -- every tile resized ON ITS OWN to max(int(tile_full * factor), 2)
            # texels per axis (nothing bleeds between tiles, static and dynamic tiles alike; culled tiles hold nothing that is ever read)
            qh, qw = int(self.args.mpi_h_verts) - 1, int(self.args.mpi_w_verts) - 1
            nth, ntw = max(int(self.tile_full[0] * factor), 2), max(int(self.tile_full[1] * factor), 2)
            h, w = qh * nth, qw * ntw
            print(f"MPV.lod:: Sparse! Resizing the tiles from {self.tile_own} to {(nth, ntw)}")
            if (hs, ws) != (h, w):
                oth, otw = self.tile_own

                def resize_plane(planes):          # (T,hs,ws,4) -> (T,h,w,4), tile by tile
                    t_ = planes.shape[0]
                    tl = planes.reshape(t_, qh, oth, qw, otw, 4).permute(0, 1, 3, 5, 2, 4).reshape(t_ * qh * qw, 4, oth, otw)
                    tl = _resize_tiles(tl, nth, ntw, aa)
                    return tl.reshape(t_, qh, qw, 4, nth, ntw).permute(0, 1, 4, 2, 5, 3).reshape(t_, h, w, 4)
                with torch.no_grad():
                    if self.packed is not None:
                        from .packed import PackedLayout
                        dev = self.stack_pool.device
                        lay = PackedLayout(self.quad_keep.to(dev), self.quad_dyn.to(dev), T, h, w, (nth, ntw))
                        pool = lay.new_pool(dev)
                        for d in range(D):
                            lay.pack_plane_(pool, d, resize_plane(self.packed.unpack_plane(self.stack_pool.data, d)))
                        del self._parameters["stack_pool"]
                        self.register_parameter("stack_pool", nn.Parameter(pool, requires_grad=True))
                        self.packed = lay
                    else:
                        from . import tiles
                        new = torch.empty((D, T, h, w, 4), dtype=self.stack.dtype, device=self.stack.device)
                        for d in range(D):
                            new[d] = resize_plane(self.stack.data[d])
                        tiles.cull_stack_(new, self.quad_keep, (nth, ntw))
                        self.register_parameter("stack", nn.Parameter(new, requires_grad=True))
            self._set_texture_geometry(h, w)
            self._install_tie_hook()
            print("MPV.los:: Resizing successful !")
            return
        print(f"MPV.lod:: Resizing the planes from {(hs, ws)} to {(h, w)}")
        if (hs, ws) != (h, w) and self.packed is not None:
            # packed model: plane by plane through the dense form of ONE plane (1/D of the dense stack), the same mask-weighted filter
            from . import tiles
            from .packed import PackedLayout
            with torch.no_grad():
                dev = self.stack_pool.device
                lay = PackedLayout(self.quad_keep.to(dev), self.quad_dyn.to(dev), T, h, w)
                pool = lay.new_pool(dev)
                for d in range(D):
                    planes = self.packed.unpack_plane(self.stack_pool.data, d).permute(0, 3, 1, 2)               # T,4,hs,ws
                    m = tiles.quad_to_texel_mask(self.quad_keep[d:d + 1], hs, ws).to(planes.dtype)[None]         # 1,1,hs,ws
                    num = torch.nn.functional.interpolate(planes * m, size=(h, w), mode="bilinear", align_corners=False, antialias=aa)
                    den = torch.nn.functional.interpolate(m, size=(h, w), mode="bilinear", align_corners=False, antialias=aa)
                    new = (num / den.clamp_min(1e-6)).permute(0, 2, 3, 1).contiguous()
                    tiles.cull_stack_(new[None], self.quad_keep[d:d + 1])
                    lay.pack_plane_(pool, d, new)
            del self._parameters["stack_pool"]
            self.register_parameter("stack_pool", nn.Parameter(pool, requires_grad=True))
            self.packed = lay
        elif (hs, ws) != (h, w):
            sparse = self.is_sparse and self.quad_keep is not None
            with torch.no_grad():
                new = torch.empty((D, T, h, w, 4), dtype=self.stack.dtype, device=self.stack.device)
                if sparse:
                    # The reference resizes every tile on its own (MPV.py:157-164), so nothing bleeds between tiles.  On the dense
                    # stack the culled texels hold the alpha logit CULLED_ALPHA (-1e4): a plain filter would pull the logits of
                    # kept texels next to a culled region to -1e3..-1e4 (sigmoid = 0, zero gradient: dead for good).  Resample
                    # with the kept-texel mask as the weight -- interpolate(v * m) / interpolate(m), a convex combination of
                    # KEPT values only -- and re-apply the culling at the new resolution.
                    from . import tiles
                    kept = tiles.quad_to_texel_mask(self.quad_keep, hs, ws).to(self.stack.dtype)            # D,hs,ws
                for d in range(D):      # plane by plane: bounds the temporaries at stage-2 sizes
                    planes = self.stack.data[d].permute(0, 3, 1, 2)                                          # T,4,hs,ws
                    if sparse:
                        m = kept[d][None, None]                                                              # 1,1,hs,ws
                        num = torch.nn.functional.interpolate(planes * m, size=(h, w), mode="bilinear", align_corners=False, antialias=aa)
                        den = torch.nn.functional.interpolate(m, size=(h, w), mode="bilinear", align_corners=False, antialias=aa)
                        planes = num / den.clamp_min(1e-6)
                    else:
                        planes = torch.nn.functional.interpolate(planes, size=(h, w), mode="bilinear", align_corners=False, antialias=aa)
                    new[d] = planes.permute(0, 2, 3, 1)
                if sparse:
                    tiles.cull_stack_(new, self.quad_keep)
            self.register_parameter("stack", nn.Parameter(new, requires_grad=True))
        sx = self.texel_scale[0] * (w - 1) / max(self.mpi_w - 1, 1)
        sy = self.texel_scale[1] * (h - 1) / max(self.mpi_h - 1, 1)
        self.spec = dataclasses.replace(self.spec, scale=(sx, sy))
        self._install_tie_hook()
        print("MPV.los:: Resizing successful !")

    def _flush_deferred_updates(self):
        """the crop-aware Adam defers the zero-gradient updates of texels outside the current crop's window: replay them before
        anything reads the whole stack (checkpoints, lod, evaluation renders)."""
        if self._window_opt is not None:
            self._window_opt.flush()

    def crop_window(self, homos, H, W, margin=3, per_plane=False):
        """texel window (y0, x0, wh, ww), aligned to the optimiser's bookkeeping tiles, that contains every tap of every pixel of the
        H x W view on every plane: the image of the view's corners under the plane homographies (convex: extremes at the corners),
        plus the +1 bilinear tap and a margin.  homos [D,3,3] on the HOST (float64).
        per_plane: also return the planes' own boxes [D,4] = (y0, y1, x0, x1) (same rule per plane; the window is their union), or None
        when a plane has the view behind it."""
        from .optim import crop_window
        Hs, Ws = self.stack_dims()[2:4]
        return crop_window(self.spec, Hs, Ws, homos, H, W, margin=margin, per_plane=per_plane)

    # ---- export to the reference's layout (MPV.py:290-341) ------------------------------------------------------------------
    def reference_state_dict(self):
        """the state_dict of the REFERENCE's MPMeshVid for these weights: plane meshes + packed static / dynamic atlases."""
        self._flush_deferred_updates()
        from .export import reference_state_dict
        return reference_state_dict(self)

    def save_mesh(self, prefix):
        """MPV.py:306-323."""
        from .export import save_mesh
        return save_mesh(self, prefix, self.reference_state_dict())

    def save_texture(self, prefix):
        """MPV.py:325-341 (dynamic frames as PNG files: no imageio / ffmpeg here)."""
        from .export import save_texture
        return save_texture(self, prefix, self.reference_state_dict())

    def get_lrate(self, step):
        """MPV.py:216-225."""
        args = self.args
        scaling = 0.1 ** (step / (args.lrate_decay * 1000))
        return [("lr", args.lrate * scaling), ("vertlr", args.lrate * getattr(args, "optimize_verts_gain", 1) * scaling)]

    def get_optimizer(self, step):
        """MPV.py:199-214.  The planar path has no vertex parameters, so there is one parameter group (the reference's
        second group holds `_verts`, which never receive a gradient in the shipped configs)."""
        (_, base_lr), _ = self.get_lrate(step)
        params = [{'params': [p for _, p in self.named_parameters()]}]
        self._flush_deferred_updates()      # the optimiser handed out before may still hold deferred zero-gradient updates: they belong to the stack
        self._static_compact = False
        self._window_opt = None
        if self.packed is not None and self.args.optimizer != 'adam':
            raise RuntimeError(f"a packed model trains through the crop-aware Adam only (optimizer = {self.args.optimizer})")
        if self.args.optimizer == 'adam':
            if self.packed is not None:
                from .optim import WindowAdam
                from .tiles import CULLED_ALPHA
                self._window_opt = WindowAdam([{'params': [self.stack_pool]}], lr=base_lr, betas=(0.9, 0.999), eps=6e-8, quad_keep=self.quad_keep,
                                              quad_dyn=self.quad_dyn, culled_alpha=CULLED_ALPHA, layout=self.packed, tile=self.tile_own,
                                              fused_backward=bool(getattr(self.args, "fused_adam_backward", True))
                                              and not getattr(self.args, "finite_window_grad", False))
                return self._window_opt
            if self.stack.is_cuda and self.is_sparse and getattr(self.args, "tile_adam", False):
                # (the round-1 optimiser of sparsified models, kept selectable: one pass over the kept texels of the WHOLE stack, static
                # gradients summed into frame 0 by the tie hook while it is the optimiser handed out last)
                from .tiles import TileAdam
                self._static_compact = True
                return TileAdam(params, lr=base_lr, betas=(0.9, 0.999), eps=6e-8, quad_keep=self.quad_keep, quad_dyn=self.quad_dyn, tile=self.tile_own)
            if self.stack.is_cuda and self.is_sparse and not self.atlas_exact:
                # sparsified model: the crop-aware Adam with the quad maps -- culled texels are no parameters, a static texel is ONE
                # parameter stored once (frame 0; the window copy shows it in every frame, its gradient is summed over the frames inside
                # the step), dynamic texels one per frame; only the crop's window is touched per step
                from .optim import WindowAdam
                from .tiles import CULLED_ALPHA
                # (fused: dynamic texels are stepped inside the render's backward, static ones by the step kernel behind it -- see the dense branch)
                fused = bool(getattr(self.args, "fused_adam_backward", True)) and not getattr(self.args, "finite_window_grad", False)
                self._window_opt = WindowAdam(params, lr=base_lr, betas=(0.9, 0.999), eps=6e-8, quad_keep=self.quad_keep,
                                              quad_dyn=self.quad_dyn, culled_alpha=CULLED_ALPHA, fused_backward=fused, tile=self.tile_own)
                return self._window_opt
            if self.stack.is_cuda and not self.atlas_exact:
                # dense model: crop-aware Adam -- the render reads a compact copy of the crop's texel window, the backward writes a
                # compact gradient, the step touches the window only; the zero-gradient updates of everything else are deferred and
                # replayed exactly (videoloop3d_amd/optim.py).  While it is attached, training renders go through its window.
                from .optim import WindowAdam
                # The step is taken INSIDE the render's backward (vl3d_render_bwd_adam: the owner's store applies it; same bits as the two
                # kernels, 6 instead of 9 streams of the window) -- `loss.backward(); optimizer.step()` of train_3dvid.py:242-244 then update
                # the parameters at the backward.  args.fused_adam_backward = False or args.finite_window_grad (someone reads the window
                # leaf's gradient: clipping, norm logging) keep the gradient tensor and the separate step kernel.
                fused = bool(getattr(self.args, "fused_adam_backward", True)) and not getattr(self.args, "finite_window_grad", False)
                self._window_opt = WindowAdam(params, lr=base_lr, betas=(0.9, 0.999), eps=6e-8, fused_backward=fused)
                return self._window_opt
            return torch.optim.Adam(params=params, lr=base_lr, betas=(0.9, 0.999), eps=6e-8)
        if self.args.optimizer == 'sgd':
            return torch.optim.SGD(params=params, lr=base_lr, momentum=0.9)
        raise RuntimeError(f"Unrecongnized optimizer type {self.args.optimizer}")

    def update_step(self, step):
        """MPV.py:227-229; geometry optimisation itself is not on the planar path."""
        if step >= getattr(self.args, "optimize_geo_start", 10000000):
            self.optimize_geometry = True

    # ---- geometry ----------------------------------------------------------------------------------------------------
    def reserve_windows(self, views):
        """Pre-size the crop-aware optimiser's window buffers for a set of training views -- (h, w, tar_extrin [1,4,4], tar_intrin [1,3,3]) on the host,
        what `forward` will be called with -- so that no iteration of the coming epochs has to grow them (a multi-GB hipMalloc each time: 2 ms on most
        boxes, a second on some).  One bookkeeping tile of slack per axis covers `add_intrin_noise`.  -> the largest window in texels (0: nothing done)."""
        opt = self._window_opt
        opt = getattr(opt, "window", opt)
        if opt is None or not hasattr(opt, "reserve"):
            return 0
        from .optim import tile_side
        ts, best = tile_side(), 0
        for h, w, e, k in views:
            e, k = torch.as_tensor(e).detach().cpu(), torch.as_tensor(k).detach().cpu()
            extrin = e @ self._on(e.device, "ref_extrin")[None, ...].inverse().to(e.dtype)
            homos = self.plane_homographies(extrin, k)
            (_, _, wh, ww), _ = self.crop_window(torch.as_tensor(homos).detach().cpu(), int(h), int(w), per_plane=True)
            if wh > 0 and ww > 0:
                Hs, Ws = self.stack_dims()[2:4]
                best = max(best, min(wh + ts, Hs) * min(ww + ts, Ws))
        if best:
            opt.reserve(best)
        return best

    def plane_homographies(self, extrin, intrin):
        """[D,3,3] target pixel -> plane pixel for the view `extrin` (ref -> target, [1,4,4]) / `intrin` [1,3,3]
        (utils_mpi.py:240-273 with src = the reference camera, plane normal (0,0,1), distance = planedepth)."""
        dev = extrin.device
        if dev.type == "cpu" and extrin.dtype == torch.float64 and not extrin.requires_grad and not getattr(self.args, "torch_homographies", False):
            # float64 host poses: the closed form in numpy (utils_mpi.plane_homographies_host) -- the same bits as the torch spelling below at a
            # third of the host time.  (float32 poses, as the reference's drivers hold them, keep the torch operators: their rounding is the
            # reference's own, which the goldens pin.)
            from .utils_mpi import plane_homographies_host
            return plane_homographies_host(self._host_np("ref_intrin_mpi"), self._host_np("planedepth"), extrin[0].numpy(),
                                           torch.as_tensor(intrin)[0].detach().cpu().numpy())
        eye = torch.eye(4, dtype=extrin.dtype, device=dev)[None]
        normal = torch.tensor([0., 0., 1.], dtype=extrin.dtype, device=dev).expand(1, self.mpi_d, 3)
        return compute_homography(eye, self._on(dev, "ref_intrin_mpi")[None].to(extrin.dtype), extrin, intrin.to(dev), normal,
                                  self._on(dev, "planedepth")[None].to(extrin.dtype))[0].float()

    def _host_np(self, name):
        """numpy mirror of a (small, constant) camera buffer, refreshed when the buffer changes."""
        buf = getattr(self, name)
        cache = self.__dict__.setdefault("_host_np_mirrors", {})
        key = (buf.data_ptr(), buf._version, str(buf.device))
        if cache.get(name, (None,))[0] != key:
            cache[name] = (key, buf.detach().cpu().numpy().copy())
        return cache[name][1]

    def _on(self, dev, name):
        """the (small, constant) camera buffers on the device of the pose tensors: poses that arrive on the HOST (as the DataLoader
        produces them, train_3dvid.py:214-216) are turned into homographies there, with no device round trip."""
        buf = getattr(self, name)
        if buf.device == dev:
            return buf
        cache = self.__dict__.setdefault("_host_mirrors", {})
        key = (name, str(dev), buf.data_ptr(), buf._version)
        if cache.get(name, (None,))[0] != key:
            cache[name] = (key, buf.detach().to(dev))
        return cache[name][1]

    # ---- render ------------------------------------------------------------------------------------------------------
    def _all_frames(self, ts):
        return len(ts) == self.frm_num and bool((torch.as_tensor(ts).cpu() == torch.arange(self.frm_num)).all())

    def _frames(self, ts):
        if self.packed is not None:
            return None if self._all_frames(ts) else self.packed.unpack_frames(self.stack_pool.data, torch.as_tensor(ts).tolist())
        if self._all_frames(ts):
            return self.stack
        return self.stack[:, torch.as_tensor(ts, device=self.stack.device).long()]

    def render(self, H, W, extrin, intrin, ts, need_layers=False, need_smooth=False):
        """MPV.py:351-475 -> (rgb [T',H,W,3], variables).  `variables['mpi']`/`['blend_weight']` (the warped per-layer
        rgba, only consumed by the smoothness/sparsity regularisers) are materialised on demand with the unfused operators."""
        if self.packed is not None and not (self.training and torch.is_grad_enabled()) and not need_layers and not need_smooth:
            # an evaluation render of a packed model reads the pool itself (vl3d_render_fwd_packed): static blocks once, dynamic blocks
            # per frame, culled blocks nowhere -- like the reference's render from its tile lists (MPV.py:389-449); no dense frames are built
            from .render import render_planes_packed
            from .tiles import CULLED_ALPHA
            self._flush_deferred_updates()
            dev = self.stack_pool.device
            homos = self.plane_homographies(extrin, intrin).to(dev)
            rgb, alpha = render_planes_packed(self.packed, self.stack_pool.data, torch.as_tensor(ts).tolist(), homos, H, W, self.spec,
                                              self.quad_keep, CULLED_ALPHA)
            variables = {"pix_to_face": None, "blend_weight": None, "mpi": None, "disp_norm": None, "alpha": alpha, "smooth_sums": None,
                         "alpha_sums": None}
            if len(self.args.bg_color) > 0:                                                     # MPV.py:455-461 (as written)
                if self.args.bg_color == "random":
                    bg_color = torch.rand(3).type_as(rgb)
                else:
                    r, g, b = map(float, self.args.bg_color.split('#'))
                    bg_color = torch.tensor([r, g, b]).type_as(rgb)
                rgb = rgb * alpha[..., None] + bg_color[None, None, None] * (- alpha[..., None] + 1)
            return rgb[..., :3], variables
        if self.packed is not None and not self._all_frames(ts):
            self._flush_deferred_updates()
        # an evaluation render of a run of consecutive frames of a dense model reads them where they lie (vl3d_render_fwd_frames): gathering
        # stack[:, ts] first moved 571 MB per 720p frame, more than the render itself reads (scripts/script_render_video.py renders one frame per
        # camera of its path)
        frame_run = None
        # (no autograd on that path: taken only where none is asked for -- an eval-mode render with grad enabled on a trainable stack, e.g. a
        # gradient check or test-time optimisation, goes through render_planes below and stays differentiable)
        if (self.packed is None and (not torch.is_grad_enabled() or not self.stack.requires_grad) and not need_layers and not need_smooth and not self.atlas_exact
                and self.stack.is_cuda and self.stack.is_contiguous() and not self._all_frames(ts)):
            tl = torch.as_tensor(ts).tolist()
            if len(tl) >= 1 and tl == list(range(tl[0], tl[0] + len(tl))) and 0 <= tl[0] and tl[-1] < self.stack.shape[1]:
                frame_run = (tl[0], len(tl))
        stack = self.stack if frame_run is not None else self._frames(ts)          # (a packed model: None for the full clip -- the window path below, or unpacked on demand)
        all_frames = stack is None or stack is getattr(self, "stack", None)
        homos = self.plane_homographies(extrin, intrin)
        smooth_sums = alpha_sums = None
        spec = self.spec
        if self.training and getattr(self.args, "add_uv_noise", False):
            # MPV.py:420-423: every sample's UV jittered by half a texel while training, one draw per (pixel, layer), shared by the frames.  The
            # kernels draw the field from a counter hash of (seed, plane, frame pixel); the seed comes from torch's host generator (no device
            # round trip; reproducible under torch.manual_seed).  Forward: the one-frame kernels; backward: the atomics kernel.
            if need_layers or self.atlas_exact:
                raise RuntimeError("add_uv_noise: not available with the materialised-layer path / atlas_exact")
            spec = dataclasses.replace(spec, uv_noise_seed=int(torch.randint(1, 2 ** 31 - 1, (1,))))
        cull_window = None
        if self._window_opt is not None:
            if self.training and torch.is_grad_enabled() and all_frames:
                # crop-aware training step: render from a compact, up-to-date copy of the texel window this view can reach
                # (homographies on the host: a few hundred bytes; CPU inputs cost nothing, device inputs one small sync)
                (y0, x0, wh, ww), boxes = self.crop_window(homos.detach().cpu(), H, W, per_plane=True)
                if (wh <= 0 or ww <= 0) and self.packed is not None:
                    # the view sees no texel of any plane (a pose far off the planes): a packed model has no dense stack to fall back to, so
                    # the step runs over one bookkeeping tile the view cannot reach -- zero gradient, Adam's zero-gradient update, as dense
                    from .optim import tile_side
                    Hs_, Ws_ = self.stack_dims()[2:4]
                    y0, x0, wh, ww, boxes = 0, 0, min(tile_side(), Hs_), min(tile_side(), Ws_), None
                if wh > 0 and ww > 0:
                    cull_window = (y0, x0) + tuple(self.stack_dims()[2:4])
                    stack = self._window_opt.window_leaf((y0, x0, wh, ww), boxes if self.per_plane_boxes else None)
                    spec = dataclasses.replace(spec, offset=(spec.offset[0] - x0, spec.offset[1] - y0))
            else:
                self._flush_deferred_updates()
        if stack is None:
            if self.training and torch.is_grad_enabled():
                raise RuntimeError("a packed model trains through the crop-aware optimiser: call get_optimizer() first (train_3dvid.py:264-265)")
            # an evaluation render of the whole clip from the pool: the frames are unpacked (the dense stack exists for this call only)
            self._flush_deferred_updates()
            stack = self.packed.unpack_frames(self.stack_pool.data, range(self.frm_num))
        if homos.device.type == "cpu" and self._param().is_cuda:
            # host homographies (a few hundred bytes): through a pinned staging buffer and an asynchronous copy -- a pageable upload is a
            # full host-device synchronisation in the middle of the forward, after which the GPU idles while the launches catch up
            homos = homos.pin_memory().to(self._param().device, non_blocking=True)
        else:
            homos = homos.to(self._param().device)
        # The window leaf's gradient goes to WindowAdam, which never reads culled texels: the backward leaves those slots UNWRITTEN
        # (uninitialised memory) instead of zero-filling them.  Off whenever something else adds to the same leaf (the materialised-layer
        # path: garbage + 0 is garbage) or the caller asked for a finite gradient everywhere (args.finite_window_grad: gradient clipping,
        # norm logging, isfinite checks on leaf.grad).
        lean_grad = cull_window is not None and not need_layers and not getattr(self.args, "finite_window_grad", False)
        # WindowAdam(fused_backward=True): the render's backward takes the optimiser's step for the window leaf (nothing else may add to it)
        fused_adam = self._window_opt if (cull_window is not None and not need_layers and getattr(self._window_opt, "fused_backward", False)) else None
        if self.atlas_exact:
            if need_smooth or self.is_sparse or tuple(stack.shape[2:4]) != (self.mpi_h, self.mpi_w):
                raise RuntimeError("atlas_exact renders the dense full-resolution stack without the fused regularisers / tile culling / lod")
            from .atlas import render_atlas_exact
            rgb, alpha = render_atlas_exact(stack, homos, H, W, self.atlas_grid_h, pixel_center=self.spec.pixel_center,
                                            rgb_act=self.spec.rgb_act, alpha_act=self.spec.alpha_act)
        elif need_smooth:
            # (the window leaf's gradient goes to WindowAdam, which never reads culled texels: the backward need not zero-fill them)
            rgb, alpha, smooth_sums, alpha_sums = render_planes_with_regularisers(stack, homos, H, W, spec,
                                                                                  quad_keep=self.quad_keep if self.is_sparse else None,
                                                                                  cull_window=cull_window,
                                                                                  grad_culled_unwritten=lean_grad, fused_adam=fused_adam)
        else:
            # a sparsified model renders with tile culling: samples in culled quads are uncovered, workgroups skip planes without kept quads
            if frame_run is not None:
                from .render import render_frame_run
                rgb, alpha = render_frame_run(stack.detach(), frame_run[0], frame_run[1], homos, H, W, spec, quad_keep=self.quad_keep if self.is_sparse else None)
            else:
                rgb, alpha = render_planes(stack, homos, H, W, spec, quad_keep=self.quad_keep if self.is_sparse else None, cull_window=cull_window,
                                           grad_culled_unwritten=lean_grad, fused_adam=fused_adam)
        variables = {"pix_to_face": None, "blend_weight": None, "mpi": None, "disp_norm": None, "alpha": alpha,
                     "smooth_sums": smooth_sums, "alpha_sums": alpha_sums}
        if need_layers and self.tile_own is not None:
            raise RuntimeError("the materialised-layer path (d_smooth_loss_weight > 0: off in every shipped configuration) is built for shared-border "
                               "stacks: load the checkpoint with init_from_mpi(..., tile_layout='lattice')")
        if need_layers:
            # the reference's `mpi` [T',H,W,K,4] (hit-slot order, MPV.py:441-449) and `blend_weight` [T',H,W,K] (MPV.py:451-453), on request
            # only: materialised with the unfused operators (no shipped configuration reads them; the fused kernels never build them)
            # `disp_norm` = sum_k blend_weight_k / z_k with z the view-space depth of the hit (MPV.py:385, 463-466: 1 / zbuf): for planar
            # geometry the hit point is the plane point under the pixel, moved into the target camera.  `pix_to_face` has no planar analogue.
            mpi, planes, inv_z = self._layers(stack, homos, H, W, spec, cull_window, extrin)
            variables["mpi"] = mpi
            variables["blend_weight"] = overcompose(mpi[..., -1], mpi[..., :-1])[1]
            variables["disp_norm"] = (overcompose(planes[..., -1], planes[..., :-1])[1] * inv_z[None]).sum(-1)
        if len(self.args.bg_color) > 0:                                                     # MPV.py:455-461 (as written)
            if self.args.bg_color == "random":
                bg_color = torch.rand(3).type_as(rgb)
            else:
                r, g, b = map(float, self.args.bg_color.split('#'))
                bg_color = torch.tensor([r, g, b]).type_as(rgb)
            rgb = rgb * alpha[..., None] + bg_color[None, None, None] * (- alpha[..., None] + 1)
        return rgb[..., :3], variables

    def _layers(self, stack, homos, H, W, spec=None, cull_window=None, extrin=None):
        """warped + activated per-layer rgba via the unfused warp kernel (differentiable, videoloop3d_amd/layers.py): (slot-ordered
        [T',H,W,K,4] = the reference's `mpi`, MPV.py:441-449; plane-indexed [T',H,W,D,4]; 1 / view-space depth of every plane under every
        pixel [H,W,D]).  spec / cull_window: the render spec and (y0, x0, Hs_plane, Ws_plane) when `stack` is the compact window copy of a
        training step."""
        from . import layers as LY
        spec = self.spec if spec is None else spec
        slots, planes, _, (_, _, xm, ym) = LY.materialise(stack, homos, H, W, spec, self.rgb_activate, self.alpha_activate,
                                                          self.quad_keep if self.is_sparse else None, cull_window)
        inv_z = None
        if extrin is not None:
            inv_z = LY.inverse_depth(xm, ym, self._on(stack.device, "ref_intrin_mpi"), self._on(stack.device, "planedepth"), extrin[0])
        return slots, planes, inv_z

    # ---- forward -----------------------------------------------------------------------------------------------------
    def objective(self, h, w, tar_extrins, tar_intrins, res, losscfg):
        """One stage-2 training objective, train_3dvid.py:228-240 on MPV.py:477-556: render the crop, the looping loss against `res`, the
        regularisers with a positive `args.<name>_loss_weight` and their weighted total
            -> (loss, swd_loss, {name: weighted term})        loss differentiable, the parts detached
        -- what `forward` + `train_3dvid.weighted_total` give, with the total formed in one launch each way (vl3d_linear_head_*).  Falls back to
        that spelling where a per-pixel term (sparsity, density, d_smooth: off in configs/mpv_base.txt) is on."""
        a = self.args
        wts = {k: float(getattr(a, f"{k}_loss_weight", 0) or 0) for k in ("sparsity", "rgb_smooth", "a_smooth", "density", "d_smooth")}
        if (not self.training or not self.stack_device_is_cuda() or wts["sparsity"] > 0 or wts["density"] > 0 or wts["d_smooth"] > 0
                or getattr(a, "unfused_terms", False)):
            from .train_3dvid import weighted_total
            _, extra = self(h, w, tar_extrins, tar_intrins, res=res, losscfg=losscfg)
            swd = extra.pop("swd")
            loss, (swd_loss,), extra_losses = weighted_total([swd], extra, lambda k: wts.get(k, 0))
            return loss, swd_loss.detach(), {k: v.detach() for k, v in extra_losses.items()}
        main_loss, gain, variables, T_ = self(h, w, tar_extrins, tar_intrins, res=res, losscfg=losscfg, _head=True)
        K_ = self.mpi_d
        nx, ny = T_ * h * (w - 1) * K_, T_ * (h - 1) * w * K_
        sums = variables["smooth_sums"] if (wts["rgb_smooth"] > 0 or wts["a_smooth"] > 0) and min(nx, ny) > 0 else None
        names = [k for k in ("rgb_smooth", "a_smooth") if wts[k] > 0] if sums is not None else []
        # coefficients: the view's loss gain on every term (MPV.py:507-531), the means' counts and the weights folded in
        if sums is not None:
            gi = {k: 1 + i for i, k in enumerate(names)}      # group 0 = swd; a term with weight 0 joins it with coefficient 0
            coef = (gain, wts["rgb_smooth"] * gain / (3 * nx), wts["rgb_smooth"] * gain / (3 * ny), wts["a_smooth"] * gain / nx, wts["a_smooth"] * gain / ny)
            grp = (0, gi.get("rgb_smooth", 0), gi.get("rgb_smooth", 0), gi.get("a_smooth", 0), gi.get("a_smooth", 0))
        else:
            coef, grp = (gain,), (0,)
        groups = sum(g << (4 * i) for i, g in enumerate(grp))
        cache = self.__dict__.setdefault("_head_coef", {})
        key = (coef, str(main_loss.device))
        cd = cache.get(key)
        if cd is None:
            if len(cache) > 64:
                cache.clear()
            cd = cache[key] = torch.tensor(coef, dtype=torch.float32, device=main_loss.device)
        total, parts = _LinearHead.apply(main_loss, sums, cd, groups, 1 + len(names))
        return total, parts[0], {k: parts[1 + i] for i, k in enumerate(names)}

    def stack_device_is_cuda(self):
        p = self.stack_pool if self.packed is not None else self.stack
        return p.is_cuda

    def forward(self, h, w, tar_extrins, tar_intrins, ts=None, res=None, losscfg=None, _head=False):
        """MPV.py:477-556.  train -> (None, {'swd': [1,1], ...}); eval -> (rgb [T',3,h,w], {})."""
        tar_extrins, tar_intrins = torch.as_tensor(tar_extrins), torch.as_tensor(tar_intrins)      # (numpy arrays pass nn.DataParallel's scatter untouched: host poses)
        if tar_extrins.is_cuda and self.training and self._window_opt is not None:
            # the crop-aware step sizes its texel window on the host: device poses (nn.DataParallel's scatter) come back once, here, instead
            # of as homographies after ~40 small device kernels
            tar_extrins, tar_intrins = tar_extrins.cpu(), tar_intrins.cpu()
        extrins = tar_extrins @ self._on(tar_extrins.device, "ref_extrin")[None, ...].inverse().to(tar_extrins.dtype)
        if ts is None:
            ts = torch.arange(self.frm_num).long()
        a = self.args
        # rgb_smooth / a_smooth / sparsity are fused into the render kernels: the [T,h,w,K,4] layer tensor is never built
        need_layers = self.training and getattr(a, "d_smooth_loss_weight", 0) > 0      # the one term that reads materialised layers (slow path)
        need_smooth = self.training and (a.rgb_smooth_loss_weight > 0 or a.a_smooth_loss_weight > 0 or a.sparsity_loss_weight > 0)
        rgb, variables = self.render(h, w, extrins, tar_intrins, ts, need_layers=need_layers, need_smooth=need_smooth)
        rgb_nhwc = rgb
        rgb = rgb.permute(0, 3, 1, 2)
        extra = {}
        if not self.training:
            return rgb, {}
        assert res is not None
        losscfg = {k: v[0].item() if torch.is_tensor(v) else v[0] for k, v in losscfg.items()}   # un-collate (MPV.py:494)
        loss_name = losscfg.pop('loss_name')
        loss_gain = losscfg.pop('loss_gain', 1.)
        loss = self.losses[loss_name]
        pad_frame = self.swd_patcht_size - 1 if self.isloop else 0
        if rgb_nhwc.is_cuda and rgb_nhwc.dtype == torch.float32 and pad_frame <= rgb_nhwc.shape[0] and not getattr(a, "unfused_prologue", False):
            # loop padding + scale-invariant gain + the loss's layout (and their backward) in three launches
            # (a patch-NN loss with a prepared captured clip takes x in the search's own form too, written in the same pass)
            want_gram = losscfg.get("y_prepared") is not None and str(loss_name).startswith("gpnn")
            x, xg = _LoopPrologue.apply(rgb_nhwc, res[0] if a.scale_invariant else None, pad_frame, want_gram)
            if want_gram:
                from .utils_vid import PreparedX
                losscfg = dict(losscfg, x_prepared=PreparedX(xg, rgb_nhwc.shape[0] + pad_frame, rgb_nhwc.shape[1], rgb_nhwc.shape[2]))
        else:
            rgb_pad = rgb
            if self.isloop:
                rgb_pad = torch.cat([rgb, rgb[:pad_frame]], 0)
            if a.scale_invariant and self.training:
                res_avg = res[0].mean(dim=0)
                rgb_avg = rgb.detach().mean(dim=0)
                scale = torch.exp(torch.log((res_avg + 0.01) / (rgb_avg + 0.01)).mean())
                scale = (scale + 3) / 4
                rgb_pad = rgb_pad * scale
            x = rgb_pad.permute(1, 0, 2, 3)[None]
        main_loss = loss(x, res.permute(0, 2, 1, 3, 4), **losscfg)
        if _head:      # (MPMeshVid.objective: the caller forms the weighted total itself, in one launch)
            return main_loss, float(loss_gain), variables, rgb.shape[0]
        extra['swd'] = main_loss.reshape(1, -1) * loss_gain

        fused_terms = None
        if variables["alpha"].is_cuda and (a.sparsity_loss_weight > 0 or a.density_loss_weight > 0) and not getattr(a, "unfused_terms", False):
            fused_terms = _PixelTerms.apply(variables["alpha"] if a.density_loss_weight > 0 else None,
                                            variables["alpha_sums"] if a.sparsity_loss_weight > 0 else None, 1e-4)
        if a.sparsity_loss_weight > 0 and fused_terms is not None:
            extra["sparsity"] = (fused_terms[0] * (float(loss_gain) / np.sqrt(self.mpi_d))).reshape(1, -1)
        elif a.sparsity_loss_weight > 0:
            sparsity = sparsity_ratio(variables["alpha_sums"], 1e-4)                       # MPV.py:511-515
            extra["sparsity"] = (sparsity.mean() / np.sqrt(self.mpi_d) * loss_gain).reshape(1, -1)
        if need_smooth:
            # means over [T,h,w-1,K,(3)] / [T,h-1,w,K,(3)] from the fused sums (MPV.py:517-531; K = mpi_d layers here)
            T_, K_ = rgb.shape[0], self.mpi_d
            nx, ny = T_ * h * (w - 1) * K_, T_ * (h - 1) * w * K_
            sums = variables["smooth_sums"]
            denorm = K_ / self.mpi_d
            if sums.is_cuda and min(nx, ny) > 0:
                gd = float(loss_gain) * denorm
                # (one small upload per distinct (crop size, view gain): the ref view and the other views alternate within an epoch)
                cache = self.__dict__.setdefault("_smooth_coef", {})
                key = (nx, ny, gd, str(sums.device))
                if key not in cache:
                    if len(cache) > 64:
                        cache.clear()
                    cache[key] = torch.tensor([gd / (3 * nx), gd / (3 * ny), gd / nx, gd / ny], dtype=torch.float32, device=sums.device)
                terms = _SmoothTerms.apply(sums, cache[key])
                if a.rgb_smooth_loss_weight > 0:
                    extra["rgb_smooth"] = terms[0].view(1, 1)
                if a.a_smooth_loss_weight > 0:
                    extra["a_smooth"] = terms[1].view(1, 1)
            else:
                if a.rgb_smooth_loss_weight > 0:
                    extra["rgb_smooth"] = ((sums[0] / (3 * nx) + sums[1] / (3 * ny)) * (loss_gain * denorm)).reshape(1, -1)
                if a.a_smooth_loss_weight > 0:
                    extra["a_smooth"] = ((sums[2] / nx + sums[3] / ny) * (loss_gain * denorm)).reshape(1, -1)
        if a.density_loss_weight > 0 and fused_terms is not None:
            extra["density"] = fused_terms[1].reshape(1, -1)
        elif a.density_loss_weight > 0:
            extra["density"] = (variables["alpha"] - 1).abs().mean().reshape(1, -1)
        if getattr(a, "d_smooth_loss_weight", 0) > 0:                                             # MPV.py:539-551 (off in every shipped configuration)
            disp = variables["disp_norm"]
            depth_grad = (disp[:, 1:, :-1] - disp[:, 1:, 1:]).abs() + (disp[:, :-1, 1:] - disp[:, 1:, 1:]).abs()
            extra["d_smooth"] = depth_grad.mean().reshape(1, -1)
        return None, extra
