"""Fused differentiable MPI/MPV render (warp + sample + activate + composite) on the HIP kernels.

Host-side mirror of the chain MPV.py:351-454 (planar geometry) / utils_mpi.py:159-176 + 92-107 of the
reference, behind a torch.autograd.Function.  The arithmetic lives in csrc/vl3d_render.hip.
"""
from dataclasses import dataclass

import torch

from . import _lib as L


@dataclass(frozen=True)
class RenderSpec:
    """Sampling / compositing conventions (see oracle/mpi_oracle.py:RenderSpec for the CPU statement).

    Defaults = the utils_mpi convention (sigmoid -> warp_homography -> over-composite).
    `RenderSpec.mpv()` = the planar MPV.py convention (pixel centres at +0.5, hard-cut quad borders,
    sample-then-activate)."""
    pixel_center: float = 0.0
    coord_mode: str = "utils_mpi"
    scale: tuple = (1.0, 1.0)
    offset: tuple = (0.0, 0.0)
    border: str = "zeros"
    act_order: str = "pre"
    rgb_act: str = "sigmoid"
    alpha_act: str = "sigmoid"
    variant: int = 0
    uv_noise_seed: int = 0      # add_uv_noise (MPV.py:420-423, MPI.py:519-522; include/vl3d.h): 0 = off, else the seed of this call's half-texel jitter field
    # TILE-EXACT layout of a tile-culled model (include/vl3d.h "Tile-exact layout"; needs quad_keep): (th, tw) texels per quad, every quad owning
    # its border row / column as the reference's sparsified atlases store them (MPI.py:380-418).  The planes are QH th x QW tw texels; `scale`
    # / `offset` then give the LATTICE coordinate (a quad spans tw - 1 of them) and the kernels add the quad index.  (0, 0): shared borders.
    tile: tuple = (0, 0)

    @staticmethod
    def mpv(rgb_act="sigmoid", alpha_act="sigmoid", scale=(1.0, 1.0), offset=(0.0, 0.0), variant=0):
        return RenderSpec(pixel_center=0.5, coord_mode="affine", scale=scale, offset=offset, border="hardcut",
                          act_order="post", rgb_act=rgb_act, alpha_act=alpha_act, variant=variant)


def _qgrid(quad_keep, spec):
    """(QH, QW) of a quad map as the C ABI takes them: negative for the tile-exact layout."""
    QH, QW = int(quad_keep.shape[1]), int(quad_keep.shape[2])
    return (-QH, -QW) if getattr(spec, "tile", (0, 0))[0] else (QH, QW)


def _desc(stack, H, W, spec, row0, col0, cull_window=None, grad_flags=0):
    D, T, Hs, Ws, C4 = stack.shape
    assert C4 == 4, "plane stack must be (D,T,Hs,Ws,4)"
    d = L.RenderDesc()
    d.D, d.T, d.Hs, d.Ws, d.H, d.W = D, T, Hs, Ws, int(H), int(W)
    d.row0, d.col0 = int(row0), int(col0)
    d.coord_mode = L.COORD[spec.coord_mode]
    d.border_mode = L.BORDER[spec.border]
    d.act_order = L.ACT_ORDER[spec.act_order]
    d.rgb_act, d.alpha_act = L.ACT[spec.rgb_act], L.ACT[spec.alpha_act]
    d.stack_dtype = 1 if stack.dtype == torch.float16 else 0
    d.pixel_center = float(spec.pixel_center)
    d.sx, d.sy = float(spec.scale[0]), float(spec.scale[1])
    d.ox, d.oy = float(spec.offset[0]), float(spec.offset[1])
    d.variant = int(spec.variant)
    d.uv_noise_seed = int(getattr(spec, "uv_noise_seed", 0)) & 0xFFFFFFFF
    if cull_window is not None:        # the stack is the texel window (y0, x0) of a (Hs_plane, Ws_plane) plane the quad grid lies over
        d.cull_row0, d.cull_col0, d.cull_Hs, d.cull_Ws = (int(v) for v in cull_window)
    d.grad_flags = int(grad_flags)
    return d


# scratch (plan) buffer of the most recent backward: element 0 viewed as int32 is 1 when the LDS-staged
# owner-computes kernel ran, 0 when the call fell back to the atomics kernel (read by tests / diagnostics only).
LAST_BWD_SCRATCH = None


class _RenderPlanes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stack, homos, H, W, spec, row0, col0, with_reg, quad_keep=None, cull_window=None, grad_culled_unwritten=False, fused_adam=None):
        ctx.set_materialize_grads(False)      # outputs the loss does not use come back as None, not as zero-filled tensors (a fill each, and reads in the backward kernels)
        L.check_cuda(stack, homos)
        # fused_adam (optim.WindowAdam with fused_backward): `stack` is its pending window leaf and the backward takes the step itself
        # (a tile-culled model: the optimiser classifies texels with ITS quad maps, the render culls with `quad_keep`: the same map, same device)
        ctx.fused_adam = fused_adam if (fused_adam is not None and (row0, col0) == (0, 0) and stack.is_contiguous()
                                        and (quad_keep is None) == (fused_adam.quad_keep is None)
                                        and (quad_keep is None or (cull_window is not None and tuple(quad_keep.shape) == tuple(fused_adam.quad_keep.shape)))
                                        and fused_adam.fuses(stack, spec)) else None
        ctx.leaf = stack if ctx.fused_adam is not None else None
        if quad_keep is not None:
            L.check_cuda(quad_keep)
            if quad_keep.dim() != 3 or quad_keep.shape[0] != stack.shape[0]:
                raise RuntimeError(f"quad_keep must be [D,QH,QW] with D = {stack.shape[0]}, got {tuple(quad_keep.shape)}")
            from .tiles import as_u8
            quad_keep = as_u8(quad_keep)
        if stack.dtype not in (torch.float32, torch.float16):
            raise RuntimeError("plane stack must be float32 or float16 (arithmetic is fp32 either way)")
        if getattr(spec, "tile", (0, 0))[0] and quad_keep is None:
            raise RuntimeError("RenderSpec.tile (tile-exact layout) belongs to a tile-culled model: pass its quad_keep map")
        stack = stack.contiguous()
        homos = homos.detach().to(torch.float32).contiguous()
        D, T = stack.shape[:2]
        if spec.coord_mode == "affine_planes":
            if homos.shape != (D, 16):
                raise RuntimeError(f"coord_mode 'affine_planes' takes per-plane records [D,16] = [{D},16] (atlas.plane_records), got {tuple(homos.shape)}")
            if quad_keep is not None:
                raise RuntimeError("tile culling is not available with per-plane texel transforms")
        elif homos.shape != (D, 3, 3):
            raise RuntimeError(f"homos must be [D,3,3] = [{D},3,3], got {tuple(homos.shape)}")
        rgb = torch.empty((T, H, W, 3), dtype=torch.float32, device=stack.device)
        alpha = torch.empty((T, H, W), dtype=torch.float32, device=stack.device)
        ctx.nothing_to_render = T == 0 or H == 0 or W == 0
        if ctx.nothing_to_render:
            # an empty `ts` / zero-area crop: grid_sample + cumprod of the reference return empty tensors (MPV.py:425-454); the
            # ABI refuses non-positive dims, so the empty case never reaches it
            ctx.save_for_backward(stack)
            z = torch.zeros
            return (rgb, alpha, z(4, dtype=torch.float32, device=stack.device),
                    z((T, H, W, 2) if with_reg else (0,), dtype=torch.float32, device=stack.device))
        desc = _desc(stack, H, W, spec, row0, col0, cull_window if quad_keep is not None else None,
                     grad_flags=1 if (grad_culled_unwritten and quad_keep is not None) else 0)
        asum = torch.empty((T, H, W, 2), dtype=torch.float32, device=stack.device) if with_reg else None
        # (with_reg: every entry point that forms the sums clears them itself -- no second fill)
        sums = (torch.empty if with_reg else torch.zeros)(4, dtype=torch.float64, device=stack.device)
        reg_state = None
        if with_reg:
            # coverage masks / pair flags / sign words of the layer differences: written by the forward, read by the backward (include/vl3d.h)
            with torch.cuda.device(stack.device):
                reg_state = torch.empty(int(L.lib().vl3d_render_reg_state_bytes(desc)), dtype=torch.uint8, device=stack.device)
        # variant bits 12-15: 1 = keep the two-pass forward with regularisers (render, then the sums kernel) for A/B and cross-checks
        fused_reg = with_reg and ((int(spec.variant) >> 12) & 0xf) != 1
        with torch.cuda.device(stack.device):
            if fused_reg and quad_keep is None:         # render + smoothness sums in ONE sweep over the stack
                L.check(L.lib().vl3d_render_fwd_reg(desc, L.ptr(stack), L.ptr(homos), L.ptr(rgb), L.ptr(alpha), L.ptr(asum), L.ptr(sums),
                                                    L.ptr(reg_state), L.stream_ptr(stack.device)), "vl3d_render_fwd_reg")
            elif fused_reg:       # tile-culled model: the slot-by-slot regulariser kernel composites the render from the samples it takes
                L.check(L.lib().vl3d_render_fwd_reg_culled(desc, L.ptr(stack), L.ptr(homos), L.ptr(quad_keep), *_qgrid(quad_keep, spec),
                                                           L.ptr(rgb), L.ptr(alpha), L.ptr(asum), L.ptr(sums), L.ptr(reg_state),
                                                           L.stream_ptr(stack.device)), "vl3d_render_fwd_reg_culled")
            elif quad_keep is None:
                L.check(L.lib().vl3d_render_fwd(desc, L.ptr(stack), L.ptr(homos), L.ptr(rgb), L.ptr(alpha), L.ptr(asum),
                                                L.stream_ptr(stack.device)), "vl3d_render_fwd")
            else:
                ncull = int(L.lib().vl3d_render_cull_scratch_bytes(desc))
                cull = torch.empty((ncull + 3) // 4, dtype=torch.float32, device=stack.device)
                L.check(L.lib().vl3d_render_fwd_culled(desc, L.ptr(stack), L.ptr(homos), L.ptr(quad_keep), *_qgrid(quad_keep, spec), L.ptr(cull), L.ptr(rgb), L.ptr(alpha), L.ptr(asum),
                                                       L.stream_ptr(stack.device)), "vl3d_render_fwd_culled")
        ctx.save_for_backward(stack, homos, rgb, alpha)
        ctx.quad_keep = quad_keep
        ctx.spec = spec
        ctx.reg_state = reg_state
        ctx.desc = desc
        ctx.with_reg = with_reg
        if with_reg and not fused_reg:
            with torch.cuda.device(stack.device):
                if quad_keep is None:
                    L.check(L.lib().vl3d_render_reg_fwd(desc, L.ptr(stack), L.ptr(homos), L.ptr(sums), L.ptr(reg_state), L.stream_ptr(stack.device)),
                            "vl3d_render_reg_fwd")
                else:
                    L.check(L.lib().vl3d_render_reg_fwd_culled(desc, L.ptr(stack), L.ptr(homos), L.ptr(quad_keep), *_qgrid(quad_keep, spec),
                                                               L.ptr(sums), L.ptr(reg_state), L.stream_ptr(stack.device)),
                            "vl3d_render_reg_fwd_culled")
        if asum is None:
            asum = torch.zeros((0,), dtype=torch.float32, device=stack.device)
        return rgb, alpha, sums.to(torch.float32), asum

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_sums, g_asum):
        if ctx.nothing_to_render:
            return (torch.zeros_like(ctx.saved_tensors[0]),) + (None,) * 11
        stack, homos, rgb, alpha = ctx.saved_tensors
        g_reg = g_sums.to(torch.float32).contiguous() if (ctx.with_reg and g_sums is not None) else None
        g_asum = g_asum.to(torch.float32).contiguous() if (ctx.with_reg and g_asum is not None) else None
        g_rgb = g_rgb.contiguous() if g_rgb is not None else torch.zeros_like(rgb)
        g_alpha = g_alpha.contiguous() if g_alpha is not None else None
        global LAST_BWD_SCRATCH
        if ctx.fused_adam is not None:
            # backward + optimiser step in one pass over the window (vl3d_render_bwd_adam): no gradient tensor exists afterwards
            LAST_BWD_SCRATCH = ctx.fused_adam.backward_step(ctx.desc, ctx.leaf, homos, rgb, alpha, g_rgb, g_alpha, g_reg, ctx.reg_state, g_asum)
            return (None,) * 12
        g_stack = torch.empty(stack.shape, dtype=stack.dtype, device=stack.device)     # the gradient has the stack's dtype in the ABI
        with torch.cuda.device(stack.device):
            nscratch = int(L.lib().vl3d_render_bwd_scratch_bytes(ctx.desc))
            # every word the kernels read is written by the plan kernels of the same call, the header included (bwd_plan_k): nothing to clear
            scratch = torch.empty((nscratch + 3) // 4, dtype=torch.float32, device=stack.device)
            if (ctx.desc.variant & 0xf) == 1 or ctx.desc.uv_noise_seed:      # (no plan kernel will write the header: the diagnostic word reads 0)
                scratch[:16].zero_()
            qk = ctx.quad_keep
            if qk is None:
                L.check(L.lib().vl3d_render_bwd(ctx.desc, L.ptr(stack), L.ptr(homos), L.ptr(rgb), L.ptr(alpha),
                                                L.ptr(g_rgb), L.ptr(g_alpha), L.ptr(g_reg), L.ptr(ctx.reg_state), L.ptr(g_asum), L.ptr(g_stack), L.ptr(scratch), nscratch,
                                                L.stream_ptr(stack.device)), "vl3d_render_bwd")
            else:
                L.check(L.lib().vl3d_render_bwd_culled(ctx.desc, L.ptr(stack), L.ptr(homos), L.ptr(qk), *_qgrid(qk, ctx.spec),
                                                       L.ptr(rgb), L.ptr(alpha), L.ptr(g_rgb), L.ptr(g_alpha), L.ptr(g_reg), L.ptr(ctx.reg_state), L.ptr(g_asum),
                                                       L.ptr(g_stack), L.ptr(scratch), nscratch, L.stream_ptr(stack.device)),
                        "vl3d_render_bwd_culled")
        LAST_BWD_SCRATCH = scratch
        return (g_stack,) + (None,) * 11


def mask_channel_supported(stack, spec):
    """the descriptors vl3d_render_fwd_mask / _bwd_mask are built for (include/vl3d.h): the planar convention stage 1 ships."""
    return (spec.coord_mode == "affine" and spec.border == "hardcut" and spec.act_order == "post" and spec.rgb_act == "sigmoid"
            and spec.alpha_act == "sigmoid" and stack.dtype == torch.float32)


class _RenderPlanesMask(torch.autograd.Function):
    """render (+ layer regularisers) with stage 1's loop mask as a fifth composited channel (MPI.py:568-583): one forward and one backward
    sweep over the stack for colours, regularisers and label; the label's gradient reaches the mask texture only (detached weights)."""

    @staticmethod
    def forward(ctx, stack, mask, homos, H, W, spec, with_reg):
        ctx.set_materialize_grads(False)      # outputs the loss does not use come back as None, not as zero-filled tensors (a fill each, and reads in the backward kernels)
        L.check_cuda(stack, mask, homos)
        D, T, Hs, Ws, _ = stack.shape
        if tuple(mask.shape) not in ((D, T, Hs, Ws), (D, T, Hs, Ws, 1)) or mask.dtype != torch.float32:
            raise RuntimeError(f"the loop-mask texture must be float32 [D,T,Hs,Ws] = {(D, T, Hs, Ws)}, got {tuple(mask.shape)} {mask.dtype}")
        if homos.shape != (D, 3, 3):
            raise RuntimeError(f"homos must be [D,3,3] = [{D},3,3], got {tuple(homos.shape)}")
        stack, mask = stack.contiguous(), mask.contiguous()
        homos = homos.detach().to(torch.float32).contiguous()
        dev = stack.device
        rgb = torch.empty((T, H, W, 3), dtype=torch.float32, device=dev)
        alpha = torch.empty((T, H, W), dtype=torch.float32, device=dev)
        label = torch.empty((T, H, W), dtype=torch.float32, device=dev)
        desc = _desc(stack, H, W, spec, 0, 0)
        asum = torch.empty((T, H, W, 2), dtype=torch.float32, device=dev) if with_reg else None
        sums = torch.empty(4, dtype=torch.float64, device=dev) if with_reg else None      # (cleared by vl3d_render_fwd_mask)
        reg_state = None
        with torch.cuda.device(dev):
            if with_reg:
                reg_state = torch.empty(int(L.lib().vl3d_render_reg_state_bytes(desc)), dtype=torch.uint8, device=dev)
            L.check(L.lib().vl3d_render_fwd_mask(desc, L.ptr(stack), L.ptr(mask), L.ptr(homos), L.ptr(rgb), L.ptr(alpha), L.ptr(label), L.ptr(asum),
                                                 L.ptr(sums), L.ptr(reg_state), L.stream_ptr(dev)), "vl3d_render_fwd_mask")
        ctx.save_for_backward(stack, mask, homos, rgb, alpha)
        ctx.desc, ctx.reg_state, ctx.with_reg = desc, reg_state, with_reg
        z = torch.zeros((0,), dtype=torch.float32, device=dev)
        return rgb, alpha, label, (sums.to(torch.float32) if with_reg else z), (asum if with_reg else z)

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_label, g_sums, g_asum):
        stack, mask, homos, rgb, alpha = ctx.saved_tensors
        dev = stack.device
        g_reg = g_sums.to(torch.float32).contiguous() if (ctx.with_reg and g_sums is not None) else None
        g_asum = g_asum.to(torch.float32).contiguous() if (ctx.with_reg and g_asum is not None) else None
        g_rgb = g_rgb.contiguous() if g_rgb is not None else torch.zeros_like(rgb)
        g_alpha = g_alpha.contiguous() if g_alpha is not None else None
        g_label = g_label.contiguous() if g_label is not None else torch.zeros_like(alpha)
        g_stack = torch.empty_like(stack)
        g_mask = torch.empty_like(mask)
        with torch.cuda.device(dev):
            nscratch = int(L.lib().vl3d_render_bwd_scratch_bytes(ctx.desc))
            scratch = torch.empty((nscratch + 3) // 4, dtype=torch.float32, device=dev)
            if (ctx.desc.variant & 0xf) == 1:
                scratch[:16].zero_()
            L.check(L.lib().vl3d_render_bwd_mask(ctx.desc, L.ptr(stack), L.ptr(mask), L.ptr(homos), L.ptr(rgb), L.ptr(alpha), L.ptr(g_rgb), L.ptr(g_alpha),
                                                 L.ptr(g_label), L.ptr(g_reg), L.ptr(ctx.reg_state), L.ptr(g_asum), L.ptr(g_stack), L.ptr(g_mask),
                                                 L.ptr(scratch), nscratch, L.stream_ptr(dev)), "vl3d_render_bwd_mask")
        global LAST_BWD_SCRATCH
        LAST_BWD_SCRATCH = scratch
        return g_stack, g_mask, None, None, None, None, None


def render_planes_with_mask(stack, mask, homos, H, W, spec: RenderSpec = RenderSpec(), with_regularisers=False):
    """-> (rgb [T,H,W,3], alpha [T,H,W], label [T,H,W], smooth_sums[4], alpha_sums [T,H,W,2]) -- render_planes (with_regularisers: plus
    render_planes_with_regularisers' sums) and the composited loop-mask label `sum_k w_k sigmoid(sample(mask_k))` of MPI.py:568-583 from the
    same pass; `mask` [D,T,Hs,Ws] logits.  Needs mask_channel_supported(stack, spec)."""
    return _RenderPlanesMask.apply(stack, mask, homos, int(H), int(W), spec, bool(with_regularisers))


class _LabelNoise(torch.autograd.Function):
    """MPI.py:568-583 under add_uv_noise (MPI.py:519-522): the loop-mask texture sampled at the UNJITTERED positions, composited with the detached
    alphas of the JITTERED colour samples (vl3d_label_noise_fwd / _bwd: the jitter field of spec.uv_noise_seed, the colour pass's).  Gradient to the mask only."""

    @staticmethod
    def forward(ctx, mask, stack, homos, H, W, spec):
        L.check_cuda(mask, stack, homos)
        D, T, Hs, Ws, _ = stack.shape
        if tuple(mask.shape) != (D, T, Hs, Ws) or mask.dtype != torch.float32 or stack.dtype != torch.float32:
            raise RuntimeError(f"loop-mask label: float32 stack (D,T,Hs,Ws,4) and mask [D,T,Hs,Ws] = {(D, T, Hs, Ws)}, got {tuple(mask.shape)} {mask.dtype}")
        if homos.shape != (D, 3, 3):
            raise RuntimeError(f"homos must be [D,3,3] = [{D},3,3], got {tuple(homos.shape)}")
        stack, mask = stack.detach().contiguous(), mask.detach().contiguous()
        homos = homos.detach().to(torch.float32).contiguous()
        desc = _desc(stack, H, W, spec, 0, 0)
        label = torch.empty((T, H, W), dtype=torch.float32, device=stack.device)
        with torch.cuda.device(stack.device):
            L.check(L.lib().vl3d_label_noise_fwd(desc, L.ptr(stack), L.ptr(mask), L.ptr(homos), L.ptr(label), L.stream_ptr(stack.device)), "vl3d_label_noise_fwd")
        ctx.save_for_backward(stack, mask, homos)
        ctx.desc = desc
        return label

    @staticmethod
    def backward(ctx, g):
        stack, mask, homos = ctx.saved_tensors
        g = g.contiguous()
        gm = torch.empty_like(mask)
        with torch.cuda.device(stack.device):
            L.check(L.lib().vl3d_label_noise_bwd(ctx.desc, L.ptr(stack), L.ptr(mask), L.ptr(homos), L.ptr(g), L.ptr(gm), L.stream_ptr(stack.device)), "vl3d_label_noise_bwd")
        return gm, None, None, None, None, None


def loop_mask_label_with_uv_noise(mask, stack, homos, H, W, spec: RenderSpec):
    """label [T,H,W] = sum_k w_k sigmoid(sample(mask_k)) with the mask sampled at the plain positions and the weights w_k = a_k T_k from the alphas at
    the positions jittered by spec.uv_noise_seed's field (MPI.py:519-522, 568-583).  mask [D,T,Hs,Ws] logits (receives the gradient), stack detached."""
    return _LabelNoise.apply(mask, stack, homos, int(H), int(W), spec)


def render_frame_run(stack, frame0, nframes, homos, H, W, spec: RenderSpec = RenderSpec(), out=None, quad_keep=None):
    """Evaluation render (no gradient) of frames frame0 .. frame0 + nframes - 1 of the clip `stack` [D,T,Hs,Ws,4], read IN PLACE
    (vl3d_render_fwd_frames) -> (rgb [n,H,W,3], alpha [n,H,W]): `render_planes(stack[:, ts], ...)` gathers the frames first -- 571 MB per
    720p frame at D = 32 on 1.1x planes, more than the render reads.  `out`: an (rgb, alpha) pair of buffers to write into; `quad_keep`
    [D,QH,QW]: a tile-culled model (a sample inside a culled quad is not covered, as in render_planes(quad_keep=...))."""
    L.check_cuda(stack, homos)
    D, T = stack.shape[:2]
    if not stack.is_contiguous() or stack.dtype not in (torch.float32, torch.float16):
        raise RuntimeError("render_frame_run: a contiguous float32 / float16 clip [D,T,Hs,Ws,4]")
    if not (0 <= frame0 and nframes >= 1 and frame0 + nframes <= T):
        raise RuntimeError(f"render_frame_run: frames {frame0} .. {frame0 + nframes - 1} leave the clip of {T}")
    if homos.shape != (D, 3, 3):
        raise RuntimeError(f"homos must be [D,3,3] = [{D},3,3], got {tuple(homos.shape)}")
    homos = homos.detach().to(torch.float32).contiguous()
    desc = _desc(stack, H, W, spec, 0, 0)
    desc.T = int(nframes)
    if out is None:
        rgb = torch.empty((nframes, H, W, 3), dtype=torch.float32, device=stack.device)
        alpha = torch.empty((nframes, H, W), dtype=torch.float32, device=stack.device)
    else:
        rgb, alpha = out
        if tuple(rgb.shape) != (nframes, H, W, 3) or tuple(alpha.shape) != (nframes, H, W) or not rgb.is_contiguous() or not alpha.is_contiguous():
            raise RuntimeError("render_frame_run: `out` must be contiguous float32 (rgb [n,H,W,3], alpha [n,H,W])")
    with torch.cuda.device(stack.device):
        if quad_keep is None:
            L.check(L.lib().vl3d_render_fwd_frames(desc, L.ptr(stack), int(frame0), int(T), L.ptr(homos), L.ptr(rgb), L.ptr(alpha), L.stream_ptr(stack.device)),
                    "vl3d_render_fwd_frames")
        else:
            L.check_cuda(quad_keep)
            if quad_keep.dim() != 3 or quad_keep.shape[0] != D:
                raise RuntimeError(f"quad_keep must be [D,QH,QW] with D = {D}, got {tuple(quad_keep.shape)}")
            from .tiles import as_u8
            qk = as_u8(quad_keep)
            ncull = int(L.lib().vl3d_render_cull_scratch_bytes(desc))
            cull = torch.empty((ncull + 3) // 4, dtype=torch.float32, device=stack.device)
            L.check(L.lib().vl3d_render_fwd_frames_culled(desc, L.ptr(stack), int(frame0), int(T), L.ptr(homos), L.ptr(qk), *_qgrid(qk, spec),
                                                          L.ptr(cull), L.ptr(rgb), L.ptr(alpha), L.stream_ptr(stack.device)), "vl3d_render_fwd_frames_culled")
    return rgb, alpha


def render_planes(stack, homos, H, W, spec: RenderSpec = RenderSpec(), window=(0, 0), quad_keep=None, cull_window=None, grad_culled_unwritten=False,
                  fused_adam=None):
    """stack (D,T,Hs,Ws,4) pre-activation fp32 (plane 0 = nearest), homos [D,3,3] (target pixel -> plane pixel).

    Returns rgb [T,H,W,3], alpha [T,H,W].  `window=(row0,col0)` renders the H x W sub-window whose top-left
    corner is frame pixel (row0,col0) -- used for row-band sharding (equivalent to utils.py:196-200
    get_new_intrin on the target intrinsics).
    `quad_keep` [D,QH,QW] (bool/uint8, optional): tile culling map (videoloop3d_amd.tiles): a sample that falls into a culled quad
    of a plane is not covered by it, and workgroups skip the planes of which they see no kept quad (include/vl3d.h).
    `cull_window` (y0, x0, Hs_plane, Ws_plane), with quad_keep: `stack` is the texel window at (y0, x0) of a plane of that size and the
    quad grid lies over the whole plane (crop-aware training renders from a compact copy of the window, optim.WindowAdam).
    `grad_culled_unwritten` (with quad_keep): the consumer of the stack gradient never reads texels no kept quad can read (WindowAdam / TileAdam
    skip them), so the backward leaves those slots UNDEFINED instead of zero-filling them (VL3D_GRAD_CULLED_UNWRITTEN, include/vl3d.h).
    `fused_adam`: an optim.WindowAdam(fused_backward=True) whose pending window leaf `stack` is -- the backward then takes its step
    (vl3d_render_bwd_adam) and the leaf receives no gradient."""
    rgb, alpha, _, _ = _RenderPlanes.apply(stack, homos, int(H), int(W), spec, int(window[0]), int(window[1]), False, quad_keep, cull_window,
                                           bool(grad_culled_unwritten), fused_adam)
    return rgb, alpha


def render_planes_with_smoothness(stack, homos, H, W, spec: RenderSpec = RenderSpec(), window=(0, 0), quad_keep=None):
    """render_planes plus the raw sums of the layer-space smoothness regularisers (MPV.py:517-531), differentiable:
    returns (rgb, alpha, sums[4]) with sums = (sum|dx rgb|, sum|dy rgb|, sum|dx a|, sum|dy a|) over frames, planes and
    neighbouring pixel pairs of the warped+activated layers -- the [T,h,w,K,4] layer tensor is never materialised."""
    rgb, alpha, sums, _ = _RenderPlanes.apply(stack, homos, int(H), int(W), spec, int(window[0]), int(window[1]), True, quad_keep)
    return rgb, alpha, sums


def render_planes_with_regularisers(stack, homos, H, W, spec: RenderSpec = RenderSpec(), window=(0, 0), quad_keep=None, cull_window=None,
                                    grad_culled_unwritten=False, fused_adam=None):
    """(rgb, alpha, smooth_sums[4], alpha_sums[T,H,W,2]): render_planes_with_smoothness plus the per-pixel (sum_k a_k,
    sum_k a_k^2) the sparsity regulariser |a|_1/|a|_2 (MPV.py:511-515, MPI.py:599-603) is built from; all differentiable."""
    return _RenderPlanes.apply(stack, homos, int(H), int(W), spec, int(window[0]), int(window[1]), True, quad_keep, cull_window,
                               bool(grad_culled_unwritten), fused_adam)


@torch.no_grad()
def render_planes_packed(layout, pool, frames, homos, H, W, spec: RenderSpec, quad_keep, culled_alpha, out=None, frames_dev=None):
    """The forward of a PACKED tile-culled model straight from its pool (videoloop3d_amd/packed.py; the reference renders a sparsified model
    from its tile lists, MPV.py:389-449): rgb [n,H,W,3], alpha [n,H,W] for the chosen `frames` -- the bits of the culled render of the
    unpacked frames, without ever building them.  No gradient (evaluation renders; training goes through the optimiser's window copy).
    `out`: (rgb, alpha) buffers to write into; `frames_dev`: the same frame indices as an int32 device tensor (a caller rendering a long path
    uploads them once instead of per call)."""
    L.check_cuda(pool, homos, quad_keep, layout.blocks)
    if spec.coord_mode != "affine" or spec.border != "hardcut" or spec.act_order != "post":
        raise RuntimeError("a packed model renders in the planar MPV convention (RenderSpec.mpv())")
    frames = [int(t) for t in frames]
    if not frames:
        return (torch.empty((0, H, W, 3), dtype=torch.float32, device=pool.device), torch.empty((0, H, W), dtype=torch.float32, device=pool.device))
    if min(frames) < 0 or max(frames) >= layout.T:
        raise IndexError(f"frame index out of range [0, {layout.T})")
    homos = homos.detach().to(torch.float32).contiguous()
    if homos.shape != (layout.D, 3, 3):
        raise RuntimeError(f"homos must be [D,3,3] = [{layout.D},3,3], got {tuple(homos.shape)}")
    d = L.RenderDesc()
    d.D, d.T, d.Hs, d.Ws, d.H, d.W = layout.D, layout.T, layout.Hs, layout.Ws, int(H), int(W)
    d.coord_mode, d.border_mode, d.act_order = L.COORD["affine"], L.BORDER["hardcut"], L.ACT_ORDER["post"]
    d.rgb_act, d.alpha_act = L.ACT[spec.rgb_act], L.ACT[spec.alpha_act]
    d.pixel_center = float(spec.pixel_center)
    d.sx, d.sy, d.ox, d.oy = float(spec.scale[0]), float(spec.scale[1]), float(spec.offset[0]), float(spec.offset[1])
    dev = pool.device
    from .tiles import as_u8
    qk = as_u8(quad_keep)
    if frames_dev is not None:
        if frames_dev.dtype != torch.int32 or frames_dev.numel() != len(frames) or not frames_dev.is_contiguous() or frames_dev.device != dev:
            raise RuntimeError("render_planes_packed: frames_dev must be the contiguous int32 device copy of `frames`")
        ft = frames_dev
    else:
        ft = torch.tensor(frames, dtype=torch.int32).to(dev, non_blocking=True)
    if out is None:
        rgb = torch.empty((len(frames), H, W, 3), dtype=torch.float32, device=dev)
        alpha = torch.empty((len(frames), H, W), dtype=torch.float32, device=dev)
    else:
        rgb, alpha = out
        if tuple(rgb.shape) != (len(frames), H, W, 3) or tuple(alpha.shape) != (len(frames), H, W) or not rgb.is_contiguous() or not alpha.is_contiguous():
            raise RuntimeError("render_planes_packed: `out` must be contiguous float32 (rgb [n,H,W,3], alpha [n,H,W])")
    with torch.cuda.device(dev):
        L.check(L.lib().vl3d_render_fwd_packed(d, L.ptr(layout.blocks), L.ptr(pool), L.ptr(ft), len(frames), L.ptr(homos), L.ptr(qk), *_qgrid(qk, spec), float(culled_alpha), L.ptr(rgb), L.ptr(alpha), L.stream_ptr(dev)), "vl3d_render_fwd_packed")
    return rgb, alpha
