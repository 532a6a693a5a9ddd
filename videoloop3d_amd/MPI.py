"""MPMesh (stage 1, static MPI) on the MI355X-native render kernels: drop-in for the hot path of the reference's MPI.py.

Mirrors /root/reference/MPI.py:36-131 (constructor), :452-594 (render) and :596-652 (forward) for PLANAR geometry with
rgb_mlp_type = 'direct' (configs/mpi_base.txt), dense and after sparsify_faces.  As in videoloop3d_amd/MPV.py the texture is the dense plane
stack `stack` (D,1,Hs,Ws,4) (+ `stack_mask` (D,1,Hs,Ws) for the learned loop mask, MPI.py:115-117) instead of the atlas grid, and coverage / UVs
are the analytic per-plane homography instead of pytorch3d's rasteriser.
  * The loop-mask channel (MPI.py:568-583: sigmoid(mask texture) composited with the DETACHED layer alphas) rides the colour pass as a fifth
    channel (vl3d_render_fwd_mask / _bwd_mask), or is a second pass of the fused renderer on (mask, -, -, alpha.detach()).
  * sparsify_faces (MPI.py:288-442, the paper's tile culling) classifies the quads exactly like the reference on identical weights
    (tiles.classify_quads_atlas, golden G15) into culled / static / dynamic maps instead of re-packing atlases; afterwards render() passes the
    quad map to the culled kernels (a sample in a culled quad is not covered: MPI.py:483-487, 544-548; train_3d.py:282-285 keeps training it).
  * init_from_mpi reads this package's checkpoints and the reference's (a sparsified one texel for texel onto its tile lattice);
    reference_state_dict / save_* write the reference's layout (videoloop3d_amd/export.py).
  * l_smooth / d_smooth / normalize_blendweight_fordepth / variables['mpi' | 'blend_weight' | 'disp_norm' | 'loopmask3d'] (off in every shipped
    configuration) come from the materialised-layer slow path (videoloop3d_amd/layers.py); direct2sh is out of scope.
Module-level parity: tests/test_gpu_reference_modules.py (the reference's own forward, golden G17), tests/test_gpu_mpv.py (oracle).
"""
import dataclasses

import numpy as np
import torch
import torch.nn as nn

from . import tiles
from .MPV import ACTIVATES, _PixelTerms, get_new_intrin, sparsity_ratio
from .render import RenderSpec, mask_channel_supported, render_planes, render_planes_with_mask, render_planes_with_regularisers
from .utils_mpi import compute_homography, make_depths

ALPHA_INIT_VAL = -3.     # MPI.py:33


class _LoopMaskLabel(torch.autograd.Function):
    """label = composite of sigmoid(mask texture) with the DETACHED layer alphas (MPI.py:568-583): a pass of the fused renderer over a
    PERSISTENT (D,1,Hs,Ws,4) buffer whose channel 0 is the mask logit and channel 3 the alpha logit -- written in place each call
    instead of torch.cat([m, m, m, alpha.detach()]) (a new 16-byte-per-texel tensor per iteration, its cat backward and three channel
    gradients to sum: 2.7 of the 8.5 ms of a 720p stage-1 iteration).  Gradient to the mask only."""

    @staticmethod
    def forward(ctx, mask, stack, buf, homos, H, W, spec):
        # (buf was filled by the caller, once per render() call: the views of a batch share it and their graphs hold it)
        with torch.enable_grad():
            leaf = buf.detach().requires_grad_(True)
            lab, _ = render_planes(leaf, homos, H, W, spec)
        ctx.leaf, ctx.lab = leaf, lab
        return lab.detach()[..., :1].contiguous()

    @staticmethod
    def backward(ctx, g):
        g3 = torch.zeros(ctx.lab.shape, dtype=g.dtype, device=g.device)
        g3[..., :1] = g
        (gb,) = torch.autograd.grad(ctx.lab, ctx.leaf, g3)
        ctx.leaf = ctx.lab = None
        return gb[..., 0].contiguous(), None, None, None, None, None, None


def crop_aware_pays(stack_shape, view_h, view_w):
    """The rule behind args.crop_aware_adam = "auto" (MPMesh.get_optimizer): a stack [D,1,Hs,Ws,4] fp32 of >= 768 MB of which a view_h x
    view_w view reaches at most a THIRD of a plane (half, until the one-pass Adam learnt to skip the texels no view has reached: the whole-
    stack step of a 720p frame on 1.6x planes went from 314 to 376 it/s, past the crop-aware 366)."""
    D_, _, Hs_, Ws_, _ = stack_shape
    return D_ * Hs_ * Ws_ * 16 >= 768 * 2 ** 20 and view_h * view_w * 3 <= Hs_ * Ws_


class MPMesh(nn.Module):
    def __init__(self, args, H, W, ref_extrin, ref_intrin, near, far, pixel_center=0.5, texel_scale=(1.0, 1.0), atlas_exact=False):
        """atlas_exact=True: sample the stack exactly like the reference samples its atlas of plane cells (MPI.py:75-81, 490-520: cell pitch
        (Aw-1)/(gw*(mpi_w-1)), per-cell sub-texel origin, neighbour-cell bleed) -- videoloop3d_amd/atlas.py, as MPMeshVid(atlas_exact=True).  A
        parity mode for dense weights that come from / go to the reference (MPV.atlas_to_stack / stack_to_atlas): the image and the loop-mask label
        (a second pass), no fused regularisers, no tile culling; needs args.atlas_grid_h."""
        super().__init__()
        self.atlas_exact = bool(atlas_exact)
        self.args = args
        mpi_h, mpi_w = int(args.mpi_h_scale * H), int(args.mpi_w_scale * W)
        self.mpi_h, self.mpi_w = mpi_h, mpi_w
        self.mpi_d, self.near, self.far = args.mpi_d, near, far
        self.H, self.W = H, W
        if getattr(args, "rgb_mlp_type", "direct") != "direct":
            raise RuntimeError(f"rgbmlp_type = {args.rgb_mlp_type} not supported (shipped configs use 'direct', mpi_base.txt:28)")
        ref_extrin, ref_intrin = np.asarray(ref_extrin), np.asarray(ref_intrin)
        assert ref_extrin.shape == (4, 4) and ref_intrin.shape == (3, 3)
        self.register_buffer("ref_extrin", torch.tensor(ref_extrin))
        self.register_buffer("ref_intrin", torch.tensor(ref_intrin).float())
        self.register_buffer("planedepth", make_depths(self.mpi_d, near, far).float().flip(0))     # MPI.py:57
        self.H_start, self.W_start = (mpi_h - H) // 2, (mpi_w - W) // 2
        self.register_buffer("ref_intrin_mpi", get_new_intrin(self.ref_intrin, -self.H_start, -self.W_start))
        stack = torch.rand((self.mpi_d, 1, mpi_h, mpi_w, 4))                                       # MPI.py:102-103
        stack[..., -1] = ALPHA_INIT_VAL
        self.stack = nn.Parameter(stack, requires_grad=True)
        self.learn_loop_mask = bool(getattr(args, "learn_loop_mask", False))
        if self.learn_loop_mask:
            self.stack_mask = nn.Parameter(torch.ones((self.mpi_d, 1, mpi_h, mpi_w)) * ALPHA_INIT_VAL, requires_grad=True)
        if args.rgb_activate not in ACTIVATES or args.alpha_activate not in ACTIVATES:
            raise RuntimeError(f"activation ({args.rgb_activate}, {args.alpha_activate}) not implemented by the HIP kernels")
        self.texel_scale = tuple(float(v) for v in texel_scale)
        self.spec = dataclasses.replace(RenderSpec.mpv(rgb_act=args.rgb_activate, alpha_act=args.alpha_activate,
                                                       scale=self.texel_scale), pixel_center=float(pixel_center))
        # the loop-mask pass: label = sigmoid(mask), alpha = the (detached) layer alpha with the model's activation
        self.spec_mask = dataclasses.replace(self.spec, rgb_act="sigmoid")
        self.alpha_activate = ACTIVATES[args.alpha_activate]
        # quads of the vertex grid (utils_mpi.py:80-89); classified by sparsify_faces
        self.quad_h, self.quad_w = max(int(getattr(args, "mpi_h_verts", 12)) - 1, 1), max(int(getattr(args, "mpi_w_verts", 15)) - 1, 1)
        self.is_sparse = False
        self.tile_own = None             # (th, tw): tile-exact layout of a sparsified REFERENCE checkpoint (init_from_mpi)
        self.has_dyn = False
        self._window_opt = None          # the crop-aware optimiser handed out by get_optimizer (optim.Stage1Adam): training renders go through its window

    def _flush_deferred_updates(self):
        """bring the whole stack up to date with the optimiser handed out last (optim.WindowAdam defers the zero-gradient updates of texels
        outside the training crops' windows): before anything reads the stack as a whole."""
        if self._window_opt is not None:
            self._window_opt.flush()

    def _apply(self, fn, *a, **k):
        self._flush_deferred_updates()
        dev = self.stack.device
        out = super()._apply(fn, *a, **k)
        if self.stack.device != dev:
            self._window_opt = None      # optimiser state lives on the old device: the driver asks for a new one
        return out

    @torch.no_grad()
    def sparsify_faces(self, erode_num=2, alpha_thresh=0.03, loop_thresh=0.5):
        """Tile Culling Algorithm of the paper (MPI.py:288-442) on the dense stack: quads whose (eroded, dilated) alpha never
        exceeds `alpha_thresh` are culled, kept quads whose (eroded, dilated) loop mask exceeds `loop_thresh` are dynamic, the
        rest static.  Registers `quad_keep`, `quad_dyn` [D,QH,QW] and writes the culling into the alpha logits."""
        print("Sparsifying the faces")
        self._flush_deferred_updates()
        self._window_opt = None          # (train_3d.py:284-286 asks for a new optimiser after this call)
        a_logit = self.stack[:, 0, :, :, 3].detach().clone()
        a_logit[a_logit == ALPHA_INIT_VAL] = -10                                                  # MPI.py:318
        alpha = self.alpha_activate(a_logit)
        loop = None
        if self.learn_loop_mask:
            m = self.stack_mask[:, 0].detach().clone()
            m[m == ALPHA_INIT_VAL] = -10                                                          # MPI.py:321
            loop = torch.sigmoid(m)
        grid_h = int(getattr(self.args, "atlas_grid_h", 0))
        rm = int(getattr(self.args, "sparsify_rmfirstlayer", 0))
        if grid_h > 0 and self.mpi_d % grid_h == 0 and tuple(self.stack.shape[2:4]) == (self.mpi_h, self.mpi_w):
            # the reference's classification as it runs it: morphology on the ATLAS of plane cells, quads judged by their tile samples
            # (same kept / dynamic quads as MPI.py:288-356 on identical weights: golden G15)
            # (on the model's device: a one-shot of max-pool / grid_sample passes over the 19-Mpixel atlas of the shipped size -- 3.8 s on the
            # host, which was a fifth of an end-to-end stage-1 run here, examples/stage1_train.py)
            keep, dyn = tiles.classify_quads_atlas(alpha, loop, grid_h, self.quad_h + 1, self.quad_w + 1, erode_num, alpha_thresh, loop_thresh, rm)
        else:
            keep, dyn = tiles.classify_quads(alpha, loop, self.quad_h, self.quad_w, erode_num, alpha_thresh, loop_thresh, rm)
        n_quad, n_mask, n_dyn = keep.numel(), int(keep.sum()), int(dyn.sum())
        print(f"mask {n_mask} / {n_quad} ({100 * n_mask / n_quad:.2f}%) quads")
        print(f"   of {n_mask}, {n_dyn} ({100 * n_dyn / max(n_mask, 1):.2f}%) is dynamic quads")
        self.register_buffer("quad_keep", keep)
        self.register_buffer("quad_dyn", dyn)
        tiles.cull_stack_(self.stack.data, keep)
        self.is_sparse = True
        self.has_dyn = True
        self.args.learn_loop_mask = False                                                         # MPI.py:440-441
        self.learn_loop_mask = False
        if hasattr(self, "stack_mask"):
            del self.stack_mask

    def state_dict(self, *args, **kwargs):
        """MPI.py-style: the tensors plus python scalars under "self.*" keys (consumed by MPMeshVid.init_from_mpi)."""
        self._flush_deferred_updates()
        sd = super().state_dict(*args, **kwargs)
        sd["self.is_sparse"] = self.is_sparse
        sd["self.quad_h"], sd["self.quad_w"] = self.quad_h, self.quad_w
        if getattr(self, "tile_own", None) is not None:
            sd["self.tile_own"] = self.tile_own
        if self.has_dyn:
            sd["self.has_dyn"] = self.has_dyn
        return sd

    # ---- driver hooks of train_3d.py (:159, 185, 284-301, 316-332) ------------------------------------------------------------
    def get_optimizer(self):
        """MPI.py:122-141 (the planar path has no vertex parameters: one parameter group)."""
        a = self.args
        params = [{'params': [p for _, p in self.named_parameters()]}]
        self._flush_deferred_updates()      # the optimiser handed out before may still hold deferred updates: they belong to the stack
        self._window_opt = None
        if a.optimizer == 'adam':
            crop_aware = getattr(a, "crop_aware_adam", "auto")
            if crop_aware == "auto":
                # the crop-aware engine pays where the optimiser's streams over the WHOLE stack dominate the iteration and a view reaches a minor
                # part of it: a large stack (>= 768 MB: seven streams of it are >= 1 ms) of which a view's footprint -- the training crop, or the
                # frame, on planes mpi_h_scale x mpi_w_scale larger -- is at most a third.  Measured (profiles/s1_crop_aware.py, D = 32, it/s whole
                # stack | crop-aware; round 5 session 4, with MPMesh.objective and the zero-skip Adam): a 360 x 640 crop of 720p planes at 1.6x
                # (1.2 GB) 608 | 872; the 720p frame on them 376 | 366; the reference's native shape (302 MB) 1498 | 1049 and a 720p frame on 1.1x
                # planes (571 MB) 470 | 389: there the window's launches and host work cost more than the streams.
                ph, pw = min(int(getattr(a, "patch_h_size", self.H)), self.H), min(int(getattr(a, "patch_w_size", self.W)), self.W)
                crop_aware = crop_aware_pays(tuple(self.stack.shape), ph, pw)
            if self.stack.is_cuda and not getattr(a, "torch_adam", False) and not self.atlas_exact and crop_aware:
                # (args.crop_aware_adam = True / False / "auto"): torch.optim.Adam's parameters through the crop-aware engine for the plane stack (a stage-1
                # iteration renders ONE crop of one view, train_3d.py:20-95: the render reads a compact copy of the crop's texel window, the step
                # touches the window only, the zero-gradient updates of the rest are deferred and replayed exactly -- optim.WindowAdam, as in
                # stage 2; a sparsified model, train_3d.py:282-286, takes its step inside the render's backward) and the one-pass Adam for the
                # loop-mask texture.  Measured in round 4 (examples/stage1_step.py, D = 32, 576 x 1024 planes, 180 x 320 crops): 635 it/s against
                # 766 with the one pass over the whole stack below, 710-860 against 1015-1100 after the module's host path was trimmed -- the
                # optimiser's GPU time halves (0.38 -> 0.2 ms) but the window adds ~15 launches and host work to an iteration of ~1 ms.  Hence
                # not the default.
                from .optim import Stage1Adam
                from .tiles import CULLED_ALPHA
                others = [p for _, p in self.named_parameters() if p is not self.stack]
                qk = self.quad_keep if (self.is_sparse and getattr(self, "quad_keep", None) is not None) else None
                fused = bool(getattr(a, "fused_adam_backward", True)) and not getattr(a, "finite_window_grad", False)
                self._window_opt = Stage1Adam(self.stack, others, lr=a.lrate, betas=(0.9, 0.999), eps=1e-8, quad_keep=qk,
                                              culled_alpha=CULLED_ALPHA, fused_backward=fused, tile=getattr(self, "tile_own", None) if qk is not None else None)
                return self._window_opt
            # torch.optim.Adam's update in one pass per parameter (tiles.TileAdam without a quad map; getattr(args, 'torch_adam') keeps torch's)
            if self.stack.is_cuda and not getattr(a, "torch_adam", False):
                return tiles.TileAdam(params, lr=a.lrate, betas=(0.9, 0.999), eps=1e-8)
            return torch.optim.Adam(params=params, lr=a.lrate, betas=(0.9, 0.999))
        if a.optimizer == 'sgd':
            return torch.optim.SGD(params=params, lr=a.lrate, momentum=0.9)
        raise RuntimeError(f"Unrecongnized optimizer type {a.optimizer}")

    def get_lrate(self, step):
        """MPI.py:143-152."""
        a = self.args
        scaling = 0.1 ** (step / (a.lrate_decay * 1000))
        return [("lr", a.lrate * scaling), ("vertlr", a.lrate * getattr(a, "optimize_verts_gain", 1) * scaling)]

    def update_step(self, step):
        """MPI.py:154-156; geometry optimisation itself is not on the planar path."""
        if step >= getattr(self.args, "optimize_geo_start", 10000000):
            self.optimize_geometry = True

    def init_from_mpi(self, state_dict, tile_layout=None):
        """MPI.py:174-205 (resume / warm start, train_3d.py:176-186): a state_dict of this class, or of the REFERENCE's MPMesh (plane
        meshes + packed atlas: resampled onto the dense stack, quad maps recovered from its face lists; the loop-mask texture of a
        reference checkpoint is not carried -- the reference drops it at sparsify time, MPI.py:440-441).
        tile_layout (sparsified reference checkpoints; default args.tile_layout or "exact", as MPMeshVid.init_from_mpi): "exact" = every quad
        keeps its tile with its own border texels -- the epochs stage 1 trains AFTER the switch-over (train_3d.py:282-285) move the two
        copies of a border sample apart (golden G19 j); "lattice" = neighbouring quads share them."""
        self._window_opt = None          # (the parameters are replaced: the driver asks for a new optimiser)
        if "stack" not in state_dict and "atlas" in state_dict:
            hv, wv = int(self.args.mpi_h_verts), int(self.args.mpi_w_verts)
            layout = tile_layout if tile_layout is not None else getattr(self.args, "tile_layout", "exact")
            if layout not in ("exact", "lattice"):
                raise RuntimeError(f"tile_layout must be 'exact' or 'lattice', got {layout!r}")
            sparse = bool(state_dict.get("self.is_sparse", False))
            tile_ref = tiles.reference_tile_size(state_dict, hv, wv) if sparse else None
            own = layout == "exact" and tile_ref is not None and not self.atlas_exact
            st, keep, dyn = tiles.stack_from_reference_state(state_dict, self.mpi_h, self.mpi_w, hv, wv, 1, own_borders=own)
            state_dict = {"ref_extrin": state_dict["ref_extrin"], "ref_intrin": state_dict["ref_intrin"], "planedepth": state_dict["planedepth"],
                          "stack": st[:, :1], "quad_keep": keep, "quad_dyn": dyn, "self.is_sparse": sparse, "self.has_dyn": sparse,
                          "self.tile_own": tile_ref if own else None}
        to = state_dict.get("self.tile_own", None)
        self.tile_own = None if to is None else (int(to[0]), int(to[1]))
        dev = self.stack.device
        self.ref_extrin.data = state_dict['ref_extrin'].type_as(self.ref_extrin)
        self.ref_intrin.data = state_dict['ref_intrin'].type_as(self.ref_intrin)
        self.planedepth.data = state_dict['planedepth'].type_as(self.planedepth)
        self.ref_intrin_mpi.data = get_new_intrin(self.ref_intrin, -self.H_start, -self.W_start)
        st = state_dict["stack"]
        if st.dim() != 5 or st.shape[0] != self.mpi_d or st.shape[1] != 1 or st.shape[-1] != 4:
            raise RuntimeError(f"checkpoint stack {tuple(st.shape)} does not match this model's {tuple(self.stack.shape)}")
        with torch.no_grad():
            if tuple(st.shape) != tuple(self.stack.shape):
                # another texture resolution over the same plane extent (a sparsified reference checkpoint on its tile lattice): the
                # plane-pixel -> texel scale follows the texture size, as in MPMeshVid.init_from_mpi / lod
                if "stack_mask" in state_dict and hasattr(self, "stack_mask"):
                    raise RuntimeError("a loop-mask texture needs the stack's resolution")
                self.stack = nn.Parameter(st.to(dev, torch.float32).contiguous(), requires_grad=True)
            else:
                self.stack.copy_(st.to(dev))
            if "stack_mask" in state_dict and hasattr(self, "stack_mask"):
                self.stack_mask.copy_(state_dict["stack_mask"].to(dev))
        hs, ws = self.stack.shape[2:4]
        if self.tile_own is not None:      # tile-exact layout: the scale gives the LATTICE coordinate, the spec carries the tile (render.RenderSpec.tile)
            th, tw = self.tile_own
            if (self.quad_h * th, self.quad_w * tw) != (hs, ws):
                raise RuntimeError(f"tile-exact checkpoint: planes of {(hs, ws)} texels are not {self.quad_h} x {self.quad_w} tiles of {self.tile_own}")
            self.spec = dataclasses.replace(self.spec, tile=(th, tw), scale=(self.texel_scale[0] * self.quad_w * (tw - 1) / max(self.mpi_w - 1, 1),
                                                                             self.texel_scale[1] * self.quad_h * (th - 1) / max(self.mpi_h - 1, 1)))
        else:
            self.spec = dataclasses.replace(self.spec, tile=(0, 0), scale=(self.texel_scale[0] * (ws - 1) / max(self.mpi_w - 1, 1),
                                                                           self.texel_scale[1] * (hs - 1) / max(self.mpi_h - 1, 1)))
        self.spec_mask = dataclasses.replace(self.spec, rgb_act="sigmoid")
        self.is_sparse = bool(state_dict.get("self.is_sparse", False))
        self.has_dyn = bool(state_dict.get("self.has_dyn", False))
        if self.is_sparse:
            self.register_buffer("quad_keep", state_dict["quad_keep"].to(dev).bool())
            self.register_buffer("quad_dyn", state_dict["quad_dyn"].to(dev).bool())
            self.args.learn_loop_mask = self.learn_loop_mask = False
            if hasattr(self, "stack_mask"):
                del self.stack_mask

    def reference_state_dict(self):
        """the state_dict of the REFERENCE's MPMesh for these weights (MPI.py:207-221): plane mesh + packed atlas tiles."""
        from .export import reference_state_dict
        self._flush_deferred_updates()
        return reference_state_dict(self)

    def save_mesh(self, prefix):
        """MPI.py:223-240."""
        from .export import save_mesh
        return save_mesh(self, prefix, self.reference_state_dict())

    def save_texture(self, prefix):
        """MPI.py:242-261."""
        from .export import save_texture
        return save_texture(self, prefix, self.reference_state_dict())

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
        produces them, train_3d.py:190-191) are turned into homographies there -- the 4 x 4 inverse and the chain of small matrix
        products were ~40 kernel launches per view on the device."""
        buf = getattr(self, name)
        if buf.device == dev:
            return buf
        cache = self.__dict__.setdefault("_host_mirrors", {})
        key = (str(dev), buf.data_ptr(), buf._version)
        if cache.get(name, (None,))[0] != key:
            cache[name] = (key, buf.detach().to(dev))
        return cache[name][1]

    def plane_homographies(self, extrin, intrin):
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

    def _layer_variables(self, homos, H, W, extrin, qk):
        """the materialised tensors of one view (slow path, videoloop3d_amd/layers.py): `mpi` [1,H,W,K,4] and `loopmask3d` [1,H,W,K,1] in hit-slot
        order (MPI.py:538-548, 575-577), `blend_weight` [1,H,W,K], `disp_norm` [1,H,W] (MPI.py:483-485, 558-561: the disparity of every
        hit normalised between far and near, blended; args.normalize_blendweight_fordepth divides the weights by alpha first)."""
        if getattr(self, "tile_own", None) is not None:
            raise RuntimeError("the materialised-layer path (d_smooth / l_smooth: off in every shipped configuration) is built for shared-border "
                               "stacks: load the checkpoint with init_from_mpi(..., tile_layout='lattice')")
        from . import layers as LY
        from .MPV import ACTIVATES
        from .utils_mpi import overcompose
        a = self.args
        mpi, planes, cov, (_, _, xm, ym) = LY.materialise(self.stack, homos, H, W, self.spec, ACTIVATES[a.rgb_activate], ACTIVATES[a.alpha_activate], qk)
        bw = overcompose(mpi[..., -1], mpi[..., :-1])[1]                                            # slot order, like the reference's
        bw_planes = overcompose(planes[..., -1], planes[..., :-1])[1]
        if getattr(a, "normalize_blendweight_fordepth", False):
            alpha = bw.sum(-1)
            bw = bw / alpha.clamp_min(1e-10)[..., None]
            bw_planes = bw_planes / alpha.clamp_min(1e-10)[..., None]
        inv_z = LY.inverse_depth(xm, ym, self._on(self.stack.device, "ref_intrin_mpi"), self._on(self.stack.device, "planedepth"), extrin)
        disp = (bw_planes * ((inv_z - 1 / self.far) / (1 / self.near - 1 / self.far))[None]).sum(-1)
        mask3d = None
        if self.learn_loop_mask:
            lab = torch.sigmoid(LY.sample_planes(self.stack_mask[..., None], homos, H, W, self.spec))   # 1,D,1,H,W
            mask3d = LY.to_slots((lab * cov[None, :, None]).permute(0, 3, 4, 1, 2), cov)
        return mpi, bw, disp, mask3d

    def render(self, H, W, extrin, intrin, need_reg=False, need_layers=False, cat_label=True):
        """MPI.py:452-594 -> (rgbl [B,H,W,3|4], variables).  One fused render per view (the kernels share one camera per call).
        need_layers: also materialise `mpi` / `blend_weight` / `disp_norm` / `loopmask3d` (slow path; no shipped configuration reads them)."""
        B = len(extrin)
        rgbs, alphas, labels, ssums, asums = [], [], [], [], []
        lay = []
        # a sparsified model (train_3d.py:282-285: the last epochs of stage 1 train it) renders with its quad map: a sample inside a culled
        # quad is NOT covered by that plane -- no face there in the reference (MPI.py:483-487, 544-548) -- so the slot-ordered smoothness
        # terms skip it and workgroups skip the planes of which they see no kept quad
        qk = self.quad_keep if (self.is_sparse and getattr(self, "quad_keep", None) is not None) else None
        # the loop mask rides the colour pass as a fifth channel where the kernels are built for it (the shipped planar convention, a CUDA
        # stack); args.loop_mask_two_pass keeps the separate label pass (A/B, cross-checks)
        if self.atlas_exact and (need_reg or need_layers or self.is_sparse or tuple(self.stack.shape[2:4]) != (self.mpi_h, self.mpi_w)):
            raise RuntimeError("atlas_exact renders the dense full-resolution stack without regularisers / materialised layers / tile culling")
        # add_uv_noise with the loop mask (MPI.py:519-522 with :568-572): the colour samples are jittered, the mask is sampled at the PLAIN positions and
        # composited with the jittered samples' alphas -- two positions per layer: the label comes from its own kernel pair (render.loop_mask_label_with_uv_noise)
        noisy_label = self.learn_loop_mask and self.training and bool(getattr(self.args, "add_uv_noise", False))
        fused_mask = (self.learn_loop_mask and self.stack.is_cuda and mask_channel_supported(self.stack, self.spec) and not noisy_label
                      and not self.is_sparse and not self.atlas_exact and not getattr(self.args, "loop_mask_two_pass", False))
        if self.learn_loop_mask and not fused_mask and not self.atlas_exact and not noisy_label:
            if getattr(self, "_mask_buf", None) is None or self._mask_buf.shape != self.stack.shape or self._mask_buf.device != self.stack.device:
                self._mask_buf = torch.zeros_like(self.stack)
            with torch.no_grad():          # channel 0: mask logit, channel 3: the layer alpha logit (detached, MPI.py:572)
                self._mask_buf[..., 0].copy_(self.stack_mask)
                self._mask_buf[..., 3].copy_(self.stack[..., 3])
        # crop-aware training (optim.Stage1Adam handed out by get_optimizer): ONE view per iteration (the reference's DataLoader(dataset, 1)),
        # rendered from a compact, up-to-date copy of the texel window the crop can reach; every other case reads the whole (flushed) stack
        windowed = (self._window_opt is not None and self.training and torch.is_grad_enabled() and B == 1 and not need_layers and not self.atlas_exact
                    and (fused_mask or not self.learn_loop_mask))
        if self._window_opt is not None and not windowed:
            self._flush_deferred_updates()
        for b in range(B):
            homos = self.plane_homographies(extrin[b:b + 1], intrin[b:b + 1])
            stack, mask, spec, cull_window, fused_adam, lean = self.stack, (self.stack_mask if self.learn_loop_mask else None), self.spec, None, None, False
            if self.training and getattr(self.args, "add_uv_noise", False):      # MPI.py:519-522 (see MPMeshVid.render); one field per view
                if need_layers or self.atlas_exact:
                    raise RuntimeError("add_uv_noise: not available with the materialised-layer path / atlas_exact")
                spec = dataclasses.replace(spec, uv_noise_seed=int(torch.randint(1, 2 ** 31 - 1, (1,))))
            if windowed:
                from .optim import crop_window
                Hs_, Ws_ = self.stack.shape[2:4]
                (y0, x0, wh, ww), boxes = crop_window(self.spec, Hs_, Ws_, homos.detach().cpu(), H, W, per_plane=True)
                if wh > 0 and ww > 0:
                    stack = self._window_opt.window_leaf((y0, x0, wh, ww), boxes)
                    spec = dataclasses.replace(spec, offset=(spec.offset[0] - x0, spec.offset[1] - y0))
                    cull_window = (y0, x0, Hs_, Ws_)
                    if mask is not None:
                        mask = mask[:, :, y0:y0 + wh, x0:x0 + ww].contiguous()
                    fused_adam = self._window_opt.window if self._window_opt.window.fused_backward else None
                    lean = not getattr(self.args, "finite_window_grad", False)
                else:
                    self._flush_deferred_updates()
            if homos.device.type == "cpu" and self.stack.is_cuda:
                # host homographies (a few hundred bytes): through a pinned staging buffer and an asynchronous copy -- a pageable upload blocks
                # the host until everything queued before it has run (1.7 of the 2.9 ms of a 720p stage-1 iteration, profiles/host_profile_stage1.py)
                homos = homos.pin_memory().to(self.stack.device, non_blocking=True)
            else:
                homos = homos.to(self.stack.device)
            if self.atlas_exact:
                from .atlas import render_atlas_exact
                gh = int(self.args.atlas_grid_h)
                rgb, alpha = render_atlas_exact(self.stack, homos, H, W, gh, pixel_center=self.spec.pixel_center, rgb_act=self.spec.rgb_act,
                                                alpha_act=self.spec.alpha_act)
                if self.learn_loop_mask:      # MPI.py:568-583: sigmoid(sample(mask texture)) composited with the DETACHED layer alphas
                    z = torch.zeros_like(self.stack_mask)
                    lab_stack = torch.stack([self.stack_mask, z, z, self.stack[..., 3].detach()], -1)
                    labels.append(render_atlas_exact(lab_stack, homos, H, W, gh, pixel_center=self.spec.pixel_center)[0][..., :1])
            elif fused_mask:
                rgb, alpha, label, ss, asum = render_planes_with_mask(stack, mask, homos, H, W, spec, with_regularisers=need_reg)
                labels.append(label[..., None])
                if need_reg:
                    ssums.append(ss)
                    asums.append(asum)
            elif need_reg:
                rgb, alpha, ss, asum = render_planes_with_regularisers(stack, homos, H, W, spec, quad_keep=qk, cull_window=cull_window if qk is not None else None,
                                                                       grad_culled_unwritten=lean and qk is not None, fused_adam=fused_adam)
                ssums.append(ss)
                asums.append(asum)
            else:
                rgb, alpha = render_planes(stack, homos, H, W, spec, quad_keep=qk, cull_window=cull_window if qk is not None else None,
                                           grad_culled_unwritten=lean and qk is not None, fused_adam=fused_adam)
            if len(self.args.bg_color) > 0:                                                       # MPI.py:550-556
                if self.args.bg_color == "random":
                    bg = torch.rand(3).type_as(rgb)
                else:
                    r, g, b_ = map(float, self.args.bg_color.split('#'))
                    bg = torch.tensor([r, g, b_]).type_as(rgb)
                rgb = rgb * alpha[..., None] + bg[None, None, None] * (- alpha[..., None] + 1)
            rgbs.append(rgb)
            alphas.append(alpha)
            if need_layers:
                lay.append(self._layer_variables(homos, H, W, extrin[b], qk))
            if noisy_label and not self.atlas_exact:
                from .render import loop_mask_label_with_uv_noise
                labels.append(loop_mask_label_with_uv_noise(self.stack_mask, self.stack, homos, H, W, spec)[..., None])      # (spec: this view's jitter field)
            elif self.learn_loop_mask and not fused_mask and not self.atlas_exact:                  # MPI.py:568-583
                labels.append(_LoopMaskLabel.apply(self.stack_mask, self.stack, self._mask_buf, homos, H, W, self.spec_mask))
        cat0 = lambda ts: ts[0] if len(ts) == 1 else torch.cat(ts, 0)      # noqa: E731  (B = 1, the reference's DataLoader(dataset, 1): no copy, no launch)
        rgb = cat0(rgbs)
        # (cat_label = False, MPMesh.objective: the fused head reads colour and label where the render wrote them -- no concatenation, no split
        # of its gradient on the way back)
        rgbl = (torch.cat([rgb, cat0(labels)], dim=-1) if self.learn_loop_mask else rgb) if cat_label else None
        variables = {"pix_to_face": None, "blend_weight": None, "mpi": None, "loopmask3d": None, "disp_norm": None,
                     "rgb": rgb, "label": cat0(labels) if self.learn_loop_mask else None,
                     "alpha": cat0(alphas),
                     "smooth_sums": (ssums[0] if len(ssums) == 1 else torch.stack(ssums).sum(0)) if ssums else None,
                     "alpha_sums": cat0(asums) if asums else None}
        if lay:
            # the B views are rasterised in one call by the reference: K = the deepest pixel of the batch
            kmax = max(m[0].shape[3] for m in lay)
            padk = lambda t, ax: torch.nn.functional.pad(t, (0, 0) * (t.dim() - 1 - ax) + (0, kmax - t.shape[ax]))   # noqa: E731
            variables["mpi"] = torch.cat([padk(m[0], 3) for m in lay], 0)
            variables["blend_weight"] = torch.cat([padk(m[1], 3) for m in lay], 0)
            variables["disp_norm"] = torch.cat([m[2] for m in lay], 0)
            if self.learn_loop_mask:
                variables["loopmask3d"] = torch.cat([padk(m[3], 3) for m in lay], 0)
        return rgbl, variables

    def forward(self, h, w, tar_extrins, tar_intrins):
        """MPI.py:596-652 -> (rgbl [B,3|4,h,w], extra)."""
        a = self.args
        tar_extrins, tar_intrins = torch.as_tensor(tar_extrins), torch.as_tensor(tar_intrins)      # (numpy arrays pass nn.DataParallel's scatter untouched: host poses)
        extrins = tar_extrins @ self._on(tar_extrins.device, "ref_extrin")[None, ...].inverse().to(tar_extrins.dtype)
        need_reg = self.training and (a.sparsity_loss_weight > 0 or a.rgb_smooth_loss_weight > 0 or a.a_smooth_loss_weight > 0)
        # d_smooth / l_smooth read materialised layers (off in every shipped configuration: slow path)
        need_layers = self.training and (getattr(a, "d_smooth_loss_weight", 0) > 0 or (getattr(a, "l_smooth_loss_weight", 0) > 0 and self.learn_loop_mask))
        rgbl, variables = self.render(h, w, extrins, tar_intrins, need_reg=need_reg, need_layers=need_layers)
        B = rgbl.shape[0]
        rgbl = rgbl.permute(0, 3, 1, 2)
        extra = {}
        if self.training:
            K_ = self.mpi_d
            denorm = K_ / self.mpi_d
            fused_terms = None
            if variables["alpha"].is_cuda and (a.sparsity_loss_weight > 0 or a.density_loss_weight > 0) and not getattr(a, "unfused_terms", False):
                fused_terms = _PixelTerms.apply(variables["alpha"] if a.density_loss_weight > 0 else None,
                                                variables["alpha_sums"] if a.sparsity_loss_weight > 0 else None, 1e-6)
            if a.sparsity_loss_weight > 0:
                if a.alpha_activate == "none":
                    raise RuntimeError("the fused sparsity term needs a non-negative alpha activation")
                if fused_terms is not None:
                    extra["sparsity"] = (fused_terms[0] * (1.0 / np.sqrt(self.mpi_d))).reshape(1, -1)
                else:
                    sp = sparsity_ratio(variables["alpha_sums"], 1e-6)                            # MPI.py:599-603
                    extra["sparsity"] = (sp.mean() / np.sqrt(self.mpi_d)).reshape(1, -1)
            if a.rgb_smooth_loss_weight > 0 or a.a_smooth_loss_weight > 0:
                nx, ny = B * h * (w - 1) * K_, B * (h - 1) * w * K_
                sums = variables["smooth_sums"]
                if sums.is_cuda and min(nx, ny) > 0 and not getattr(a, "unfused_terms", False):
                    # both means from the four fused sums in two launches each way (MPV._SmoothTerms) instead of ~14 scalar kernels each way:
                    # a stage-1 iteration is bound by its launches (one small upload per distinct crop size)
                    from .MPV import _SmoothTerms
                    cache = self.__dict__.setdefault("_smooth_coef", {})
                    key = (nx, ny, denorm, str(sums.device))
                    if key not in cache:
                        if len(cache) > 64:
                            cache.clear()
                        cache[key] = torch.tensor([denorm / (3 * nx), denorm / (3 * ny), denorm / nx, denorm / ny], dtype=torch.float32, device=sums.device)
                    terms = _SmoothTerms.apply(sums, cache[key])
                    if a.rgb_smooth_loss_weight > 0:                                             # MPI.py:605-611
                        extra["rgb_smooth"] = terms[0].view(1, 1)
                    if a.a_smooth_loss_weight > 0:                                               # MPI.py:613-619
                        extra["a_smooth"] = terms[1].view(1, 1)
                else:
                    if a.rgb_smooth_loss_weight > 0:                                             # MPI.py:605-611
                        extra["rgb_smooth"] = ((sums[0] / (3 * nx) + sums[1] / (3 * ny)) * denorm).reshape(1, -1)
                    if a.a_smooth_loss_weight > 0:                                               # MPI.py:613-619
                        extra["a_smooth"] = ((sums[2] / nx + sums[3] / ny) * denorm).reshape(1, -1)
            if getattr(a, "d_smooth_loss_weight", 0) > 0:                                        # MPI.py:622-637
                disp = variables["disp_norm"]
                depth_grad = (disp[:, 1:, :-1] - disp[:, 1:, 1:]).abs() + (disp[:, :-1, 1:] - disp[:, 1:, 1:]).abs()
                c = rgbl[:, :3]
                edge = (c[..., 1:, :-1] - c[..., 1:, 1:]).abs().sum(dim=1) + (c[..., :-1, 1:] - c[..., 1:, 1:]).abs().sum(dim=1)
                extra["d_smooth"] = (depth_grad * (- edge * a.edge_scale + 1).clamp_min(0)).mean().reshape(1, -1)
            if getattr(a, "l_smooth_loss_weight", 0) > 0 and variables["loopmask3d"] is not None:   # MPI.py:639-645
                lm = variables["loopmask3d"][..., 0]
                sm = (lm[:, :, :-1] - lm[:, :, 1:]).abs().mean() + (lm[:, :-1] - lm[:, 1:]).abs().mean()
                extra["l_smooth"] = (sm * (lm.shape[-1] / self.mpi_d)).reshape(1, -1)
            if a.density_loss_weight > 0:                                                        # MPI.py:647-650
                extra["density"] = (fused_terms[1] if fused_terms is not None else (variables["alpha"] - 1).abs().mean()).reshape(1, -1)
        return rgbl, extra


    def objective(self, h, w, tar_extrins, tar_intrins, target, target_mask=None, scale_invariant=True):
        """One stage-1 training objective, train_3d.py:196-232 on MPI.py:596-652: render the view, form img_loss (+ loop_loss) against `target`
        [B,3,h,w] (`target_mask` [B,h,w]), the regularisers with a positive `args.<name>_loss_weight`, and their weighted total
            -> (loss, img_loss, loop_loss, {name: weighted term})        loss differentiable, the parts detached
        -- what `forward` + `image_and_loop_loss` + `train_3dvid.weighted_total` give (tests/test_gpu_stage1_driver.py compares the two),
        but with the whole scalar head in three launches and one on the way back (vl3d_stage1_objective): a stage-1 iteration at the
        reference's crop is ~0.45 ms of GPU work that the ~45 one-element torch launches of the generic spelling kept host bound at 0.74 ms.
        Falls back to the generic spelling where the head is not built for the call (CPU, several views, materialised-layer terms)."""
        a = self.args
        wts = {k: float(getattr(a, f"{k}_loss_weight", 0) or 0) for k in ("sparsity", "rgb_smooth", "a_smooth", "density", "d_smooth", "l_smooth")}
        slow = (not self.training or not self.stack.is_cuda or len(tar_extrins) != 1 or getattr(a, "unfused_terms", False)
                or wts["d_smooth"] > 0 or (wts["l_smooth"] > 0 and self.learn_loop_mask) or self.atlas_exact
                or (self.learn_loop_mask and target_mask is None))
        if slow:
            from .train_3dvid import weighted_total
            rgbl, extra = self(h, w, tar_extrins, tar_intrins)
            img_loss, loop_loss = image_and_loop_loss(rgbl, target, target_mask if self.learn_loop_mask else None, scale_invariant=scale_invariant)
            mains = [img_loss] + ([loop_loss] if torch.is_tensor(loop_loss) else [])
            loss, _, extra_losses = weighted_total(mains, extra, lambda k: wts.get(k, 0))
            return loss, img_loss.detach(), (loop_loss.detach() if torch.is_tensor(loop_loss) else loop_loss), {k: v.detach() for k, v in extra_losses.items()}
        tar_extrins, tar_intrins = torch.as_tensor(tar_extrins), torch.as_tensor(tar_intrins)
        extrins = tar_extrins @ self._on(tar_extrins.device, "ref_extrin")[None, ...].inverse().to(tar_extrins.dtype)
        if wts["sparsity"] > 0 and a.alpha_activate == "none":
            raise RuntimeError("the fused sparsity term needs a non-negative alpha activation")
        need_reg = wts["sparsity"] > 0 or wts["rgb_smooth"] > 0 or wts["a_smooth"] > 0
        _, var = self.render(h, w, extrins, tar_intrins, need_reg=need_reg, cat_label=False)
        B, K_ = 1, self.mpi_d
        nx, ny = B * h * (w - 1) * K_, B * (h - 1) * w * K_
        smooth_on = (wts["rgb_smooth"] > 0 or wts["a_smooth"] > 0) and min(nx, ny) > 0
        coef = (1.0 / (3 * nx), 1.0 / (3 * ny), 1.0 / nx, 1.0 / ny) if smooth_on else (0.0,) * 4      # (denorm = K / mpi_d = 1, MPI.py:605-619)
        cfg = (int(B), int(h), int(w), bool(scale_invariant), 1.0, 1.0, wts["sparsity"], wts["density"],
               wts["rgb_smooth"] if smooth_on else 0.0, wts["a_smooth"] if smooth_on else 0.0, 1.0 / float(np.sqrt(self.mpi_d)), 1e-6) + coef
        total, parts = _Stage1Objective.apply(var["rgb"], var["label"], var["alpha"] if wts["density"] > 0 else None,
                                              var["alpha_sums"] if wts["sparsity"] > 0 else None, var["smooth_sums"] if smooth_on else None,
                                              target, target_mask if self.learn_loop_mask else None, cfg)
        extra = {}
        if wts["sparsity"] > 0:
            extra["sparsity"] = parts[3]
        if wts["rgb_smooth"] > 0 and smooth_on:
            extra["rgb_smooth"] = parts[5]
        if wts["a_smooth"] > 0 and smooth_on:
            extra["a_smooth"] = parts[6]
        if wts["density"] > 0:
            extra["density"] = parts[4]
        return total, parts[1], (parts[2] if self.learn_loop_mask else 0), extra


class _Stage1Objective(torch.autograd.Function):
    """(rgb [B,h,w,3], label [B,h,w] | None, alpha [B,h,w] | None, alpha_sums [B,h,w,2] | None, smooth_sums [4] | None) -- the render's outputs
    as it wrote them -- and the targets -> (total, parts [8]) through vl3d_stage1_objective; the gradients of the total w.r.t. every input
    exist after the forward, the backward scales them by the upstream gradient in one launch (none when it is 1)."""

    @staticmethod
    def forward(ctx, rgb, label, alpha, asum, ssums, target, tmask, cfg):
        ctx.set_materialize_grads(False)      # outputs the loss does not use come back as None, not as zero-filled tensors (a fill each, and reads in the backward kernels)
        from . import _lib as L
        L.check_cuda(rgb, target)
        B, h, w = cfg[0], cfg[1], cfg[2]
        n = B * h * w
        f32c = lambda t: None if t is None else (t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().to(torch.float32).contiguous())  # noqa: E731
        rgb_, label_, alpha_, asum_, ss_ = (f32c(t) for t in (rgb, label, alpha, asum, ssums))
        # the targets go in through their strides (a crop of the view's image: no copy) as long as their columns are contiguous
        strided = lambda t: None if t is None else (t.detach() if (t.dtype == torch.float32 and t.stride(-1) == 1) else t.detach().to(torch.float32).contiguous())  # noqa: E731
        tgt_, tm_ = strided(target), strided(tmask)
        if tuple(rgb_.shape) != (B, h, w, 3) or tuple(tgt_.shape) != (B, 3, h, w):
            raise RuntimeError(f"stage-1 objective: rgb {tuple(rgb_.shape)} / target {tuple(tgt_.shape)} do not describe {B} view(s) of {h} x {w}")
        if label_ is not None and tuple(label_.shape) == (B, h, w, 1):      # (MPMesh.render hands the label out with its channel axis)
            label_ = label_.view(B, h, w)
        for t_, shp in ((label_, (B, h, w)), (tm_, (B, h, w)), (alpha_, (B, h, w)), (asum_, (B, h, w, 2)), (ss_, (4,))):
            if t_ is not None and tuple(t_.shape) != shp:
                raise RuntimeError(f"stage-1 objective: an input of shape {tuple(t_.shape)} where {shp} is expected")
        r4 = lambda v: (v + 3) & ~3  # noqa: E731
        # one buffer: [6 doubles scratch | out 8] [g_rgb | g_label | g_alpha | g_asum | g_smooth], every part 16-byte aligned
        o_out, o_rgb = 12, 20
        o_label = o_rgb + r4(3 * n)
        o_alpha = o_label + (r4(n) if label_ is not None else 0)
        o_asum = o_alpha + (r4(n) if alpha_ is not None else 0)
        o_ss = o_asum + (r4(2 * n) if asum_ is not None else 0)
        end = o_ss + (4 if ss_ is not None else 0)
        dev = rgb_.device
        buf = torch.empty(end, dtype=torch.float32, device=dev)
        base = buf.data_ptr()
        at = lambda off, on=True: L.C.c_void_p(base + 4 * off) if on else None  # noqa: E731
        d = L.Stage1ObjectiveDesc(B, h, w, 1 if cfg[3] else 0, *[float(v) for v in cfg[4:12]], (L.C.c_float * 4)(*[float(v) for v in cfg[12:16]]))
        with torch.cuda.device(dev):
            L.check(L.lib().vl3d_stage1_objective(L.C.byref(d), L.ptr(rgb_), L.ptr(label_), L.ptr(alpha_), L.ptr(asum_), L.ptr(ss_), L.ptr(tgt_), tgt_.stride(0), tgt_.stride(1), tgt_.stride(2),
                                                  L.ptr(tm_), 0 if tm_ is None else tm_.stride(0), 0 if tm_ is None else tm_.stride(1), at(0), at(o_out), at(o_rgb), at(o_label, label_ is not None), at(o_alpha, alpha_ is not None),
                                                  at(o_asum, asum_ is not None), at(o_ss, ss_ is not None), L.stream_ptr(dev)), "vl3d_stage1_objective")
        ctx.label_shape = None if label is None else tuple(label.shape)
        ctx.buf, ctx.layout = buf, (n, B, h, w, o_rgb, o_label if label_ is not None else -1, o_alpha if alpha_ is not None else -1,
                                    o_asum if asum_ is not None else -1, o_ss if ss_ is not None else -1, end)
        parts = buf[o_out:o_out + 8]
        ctx.mark_non_differentiable(parts)
        return buf[o_out], parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        from . import _lib as L
        if g_total is None:
            return (None,) * 8
        buf = ctx.buf
        # the gradients are scaled IN PLACE (they belong to this node): it can be differentiated once, like the fused looping loss
        if getattr(ctx, "consumed", False):
            raise RuntimeError("the fused stage-1 objective keeps its gradient buffer in place: backward through it a second time needs a new forward")
        ctx.consumed = True
        n, B, h, w, o_rgb, o_label, o_alpha, o_asum, o_ss, end = ctx.layout
        g = g_total if (g_total.dtype == torch.float32 and g_total.is_contiguous()) else g_total.to(torch.float32).contiguous()
        with torch.cuda.device(buf.device):
            L.check(L.lib().vl3d_scale_inplace(end - o_rgb, L.C.c_void_p(buf.data_ptr() + 4 * o_rgb), L.ptr(g), L.stream_ptr(buf.device)), "vl3d_scale_inplace")
        cut = lambda off, m, shape: None if off < 0 else buf[off:off + m].view(shape)  # noqa: E731
        return (cut(o_rgb, 3 * n, (B, h, w, 3)), cut(o_label, n, ctx.label_shape), cut(o_alpha, n, (B, h, w)), cut(o_asum, 2 * n, (B, h, w, 2)),
                cut(o_ss, 4, (4,)), None, None, None)


class _Stage1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgbl, target, target_mask, scale_invariant):
        from . import _lib as L
        L.check_cuda(rgbl, target)
        B, C, h, w = rgbl.shape
        if C not in (3, 4) or tuple(target.shape) != (B, 3, h, w) or (C == 4 and (target_mask is None or tuple(target_mask.shape) != (B, h, w))):
            raise RuntimeError(f"stage-1 loss: rgbl {tuple(rgbl.shape)} needs target [B,3,h,w] (and target_mask [B,h,w] for the loop-mask channel)")
        r = rgbl.detach()
        if r.dtype != torch.float32 or r.stride(3) * w != r.stride(2):            # pixels of an image must be one strided run (NCHW or NHWC views are)
            r = r.to(torch.float32).contiguous()
        t = target.detach().to(torch.float32).contiguous()
        m = None if C == 3 else target_mask.detach().to(torch.float32).contiguous()
        dev = r.device
        scratch = torch.empty(3, dtype=torch.float64, device=dev)
        grad = torch.empty((B, h, w, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().vl3d_stage1_loss(B, C, h, w, L.ptr(r), r.stride(0), r.stride(1), r.stride(3), L.ptr(t), L.ptr(m), 1 if scale_invariant else 0,
                                             L.ptr(scratch[2:]), L.ptr(scratch), L.ptr(grad), L.stream_ptr(dev)), "vl3d_stage1_loss")
        ctx.grad, ctx.counts = grad, (3 * B * h * w, B * h * w)
        out = scratch[:2].to(torch.float32)
        return out[0] / ctx.counts[0], out[1] / ctx.counts[1]

    @staticmethod
    def backward(ctx, g_img, g_loop):
        g = ctx.grad
        C = g.shape[-1]
        k = torch.stack([g_img / ctx.counts[0]] * 3 + ([g_loop / ctx.counts[1]] if C == 4 else [])).to(g.dtype)
        return (g * k).permute(0, 3, 1, 2), None, None, None


def image_and_loop_loss(rgbl, target, target_mask=None, scale_invariant=True):
    """The image + loop-mask part of a stage-1 iteration's loss (train_3d.py:200-220) on MPMesh.forward's output rgbl [B,3|4,h,w]:
        loop_loss = -mean(m log l + (1 - m) log(1 - l)),  l = clamp(rgbl[:, -1], .001, .999)            (learn_loop_mask)
        img_loss  = mean((rgb * s - target)^2),  s = (exp(mean log((target + .01) / (rgb.detach() + .01))) + 3) / 4      (scale_invariant)
    -> (img_loss, loop_loss) 0-d tensors (loop_loss = 0 without the mask channel), differentiable w.r.t. rgbl, in three launches each way
    (vl3d_stage1_loss) instead of the ~60 of the torch chain."""
    if not rgbl.is_cuda:
        raise RuntimeError("videoloop3d_amd operators run on the MI355X only; there is no CPU fallback")
    return _Stage1Loss.apply(rgbl, target, target_mask, bool(scale_invariant))
