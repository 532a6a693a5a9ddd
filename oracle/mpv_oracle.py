"""CPU oracle for MPMeshVid.forward / MPMesh.forward (planar geometry).  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/MPV.py:477-556 and MPI.py:596-652 (forward), MPV.py:441-475 / MPI.py:538-594 (what render() does after the
rasteriser) on top of the pinned operator oracles (mpi_oracle / vid_oracle / atlas_oracle).

Pinning (round 4): golden G17 (tests/golden/make_golden_r04.py) holds the outputs of the reference's OWN MPI.py / MPV.py -- imported with
name-only stand-ins for the absent packages and run through a harness-side analytic rasteriser -- for dense and sparsified models of both
classes: rgb / label, every entry of `extra` (loop padding, scale-invariant gain, loss dispatch, `masked_scatter` hit-slot order and the
K / mpi_d normalisation of the smoothness terms, sparsity, density, l_smooth, d_smooth with and without edge weights / blend-weight
normalisation, background colour) and gradients w.r.t. the atlases.  tests/test_reference_modules_cpu.py checks this file against them.
Still unpinned, by necessity: the un-vendored rasteriser's pixel-centre (+0.5) and edge rule (parameters here; the harness uses the
same reading of pytorch3d).

Texture sources: `stack` is a plane stack (D,T,Hs,Ws,4) whose Hs x Ws texels span the mpi_h x mpi_w plane pixels (pitch 1 when they are
equal; the tile lattice of a sparsified reference checkpoint otherwise; `quad_keep` for its culled quads) or, with `atlas_grid_h`, the
reference's own atlas of plane cells (T,4,Ah,Aw) sampled the way MPV.py:75-81, 394-439 samples it (oracle/atlas_oracle.py).
"""
import numpy as np
import torch

from . import atlas_oracle as AO
from . import mpi_oracle as MO
from . import vid_oracle as VO


def _geometry(args, D, H, W, ref_extrin, ref_intrin, near, far, tar_extrins, tex_hw):
    ref_extrin = torch.as_tensor(ref_extrin)
    ref_intrin = torch.as_tensor(ref_intrin).float()
    mpi_h = int(args.mpi_h_scale * H) if hasattr(args, "mpi_h_scale") else tex_hw[0]
    mpi_w = int(args.mpi_w_scale * W) if hasattr(args, "mpi_w_scale") else tex_hw[1]
    planedepth = MO.make_depths(D, near, far).flip(0)                                   # MPV.py:51
    H_start, W_start = (mpi_h - H) // 2, (mpi_w - W) // 2                               # MPV.py:55
    ref_intrin_mpi = MO.get_new_intrin(ref_intrin, -H_start, -W_start)
    extrins = tar_extrins @ ref_extrin[None].inverse().to(tar_extrins.dtype)            # MPV.py:478
    return mpi_h, mpi_w, planedepth, ref_intrin_mpi, extrins


def _homos(ref_intrin_mpi, extrin, intrin, planedepth):
    D = planedepth.numel()
    eye = torch.eye(4, dtype=extrin.dtype)[None]
    normal = torch.tensor([0., 0., 1.], dtype=extrin.dtype).expand(1, D, 3)
    return MO.compute_homography(eye, ref_intrin_mpi[None].to(extrin.dtype), extrin, intrin, normal, planedepth[None].to(extrin.dtype))[0].float()


def _layers(tex, homos, h, w, args, mpi_h, mpi_w, pixel_center, atlas_grid_h, quad_keep, acts=None, tile=None, uv_noise_seed=0):
    """plane-indexed activated layers [T,h,w,D,C] + coverage [h,w,D] of a stack or (atlas_grid_h) of the reference's atlas.
    tile = (th, tw): the stack is in the TILE-EXACT layout (every quad of `quad_keep`'s grid owns a th x tw tile, border texels included --
    the reference's sparsified atlases, MPI.py:380-418; pinned by golden G19)."""
    if atlas_grid_h is not None:
        acts = (MO.ACTS[args.rgb_activate], MO.ACTS[args.alpha_activate]) if acts is None else acts
        return AO.sample_atlas_layers(tex, homos, h, w, atlas_grid_h, mpi_h, mpi_w, pixel_center, acts)
    Hs, Ws = tex.shape[2:4]
    if tile is not None:      # plane pixel -> LATTICE coordinate: a quad spans tile - 1 of them
        QH, QW = quad_keep.shape[1:]
        scale, tl = (QW * (tile[1] - 1) / max(mpi_w - 1, 1), QH * (tile[0] - 1) / max(mpi_h - 1, 1)), (int(tile[0]), int(tile[1]))
    else:
        scale, tl = ((Ws - 1) / max(mpi_w - 1, 1), (Hs - 1) / max(mpi_h - 1, 1)), (0, 0)
    spec = MO.RenderSpec(pixel_center=pixel_center, coord_mode="affine", border="hardcut", act_order="post", scale=scale, tile=tl, uv_noise_seed=int(uv_noise_seed),
                         rgb_act=args.rgb_activate if acts is None else acts[0], alpha_act=args.alpha_activate if acts is None else acts[1])
    return MO.sample_layers(tex, homos, h, w, spec, quad_keep)


def _inv_depth(homos, h, w, ref_intrin_mpi, planedepth, extrin, pixel_center):
    """1 / zbuf of the planar mesh under every pixel, [h,w,D]: the plane point under the pixel moved into the target camera (MPV.py:385)."""
    xs, ys = MO._homography_source_coords(h, w, homos.double(), pixel_center)              # plane pixels [D,h,w]
    ray = torch.inverse(ref_intrin_mpi.double()) @ torch.stack([xs, ys, torch.ones_like(xs)], -1)[..., None]
    P = ray[..., 0] * planedepth.double()[:, None, None, None]
    E = extrin.double()
    z = (P * E[2, :3]).sum(-1) + E[2, 3]
    return (1.0 / z).float().permute(1, 2, 0)


def _slots(x, cov):
    """[...,h,w,D,(C)] plane-indexed -> hit-slot order (MO.layers_to_slots) for tensors with or without a channel axis."""
    if x.dim() == 4:
        return MO.layers_to_slots(x[..., None], cov)[..., 0]
    return MO.layers_to_slots(x, cov)


def _smooth(t, D):
    """(|dx| mean + |dy| mean) * K / mpi_d of a slot-ordered layer tensor [B,h,w,K,(C)] (MPV.py:517-531, MPI.py:605-620, 639-645)."""
    K = t.shape[3]
    return ((t[:, :, :-1] - t[:, :, 1:]).abs().mean() + (t[:, :-1] - t[:, 1:]).abs().mean()) * (K / D)


def mpv_forward(stack, args, H, W, ref_extrin, ref_intrin, near, far, h, w, tar_extrins, tar_intrins, ts=None, res=None,
                losscfg=None, training=True, pixel_center=0.5, atlas_grid_h=None, quad_keep=None, tile=None):
    """stack (D,T,Hs,Ws,4), or the atlas (T,4,Ah,Aw) with atlas_grid_h.  Returns (rgb [T',3,h,w] or None, extra dict) like MPV.py:553-556."""
    D = args.mpi_d if atlas_grid_h is not None else stack.shape[0]
    T = stack.shape[0] if atlas_grid_h is not None else stack.shape[1]
    mpi_h, mpi_w, planedepth, ref_intrin_mpi, extrins = _geometry(args, D, H, W, ref_extrin, ref_intrin, near, far, tar_extrins,
                                                                  stack.shape[2:4])
    if ts is None:
        ts = torch.arange(T)
    homos = _homos(ref_intrin_mpi, extrins, tar_intrins, planedepth)
    tex = stack[ts] if atlas_grid_h is not None else stack[:, ts]
    layers, cov = _layers(tex, homos, h, w, args, mpi_h, mpi_w, pixel_center, atlas_grid_h, quad_keep, tile=tile)
    rgb, bw = MO.overcompose(layers[..., 3], layers[..., :3])
    alpha = bw.sum(-1)
    mpi = MO.layers_to_slots(layers, cov)
    if len(args.bg_color) > 0:                                                          # MPV.py:455-461
        r, g, b = map(float, args.bg_color.split('#'))
        bg = torch.tensor([r, g, b]).type_as(rgb)
        rgb = rgb * alpha[..., None] + bg[None, None, None] * (1 - alpha[..., None])
    rgb = rgb.permute(0, 3, 1, 2)
    if not training:
        return rgb, {}
    extra = {}
    rgb_pad = rgb
    if args.mpv_isloop:
        rgb_pad = torch.cat([rgb, rgb[:args.swd_patcht_size - 1]], 0)                   # MPV.py:490-492
    cfg = {k: (v[0].item() if torch.is_tensor(v) else v[0]) for k, v in losscfg.items()}
    loss_name = cfg.pop("loss_name")
    loss_gain = cfg.pop("loss_gain", 1.0)
    if args.scale_invariant:
        res_avg = res[0].mean(dim=0)
        rgb_avg = rgb.detach().mean(dim=0)
        scale = torch.exp(torch.log((res_avg + 0.01) / (rgb_avg + 0.01)).mean())
        rgb_pad = rgb_pad * ((scale + 3) / 4)
    x = rgb_pad.permute(1, 0, 2, 3)[None]
    y = res.permute(0, 2, 1, 3, 4)
    if loss_name == "gpnn":                                                             # Patch3DGPNNDirectLoss: no trimming (utils_vid.py:265-286)
        cfg.pop("dist_fn", None)
        main, _, _ = VO.gpnn_direct_loss(x, y, **cfg)
    elif loss_name == "gpnn_lm":
        cfg.pop("dist_fn", None)
        main, _, _ = VO.gpnn_loss(x, y, **cfg)
    elif loss_name == "mse":
        frm = min(x.shape[2], y.shape[2])
        main = ((x[:, :, :frm] - y[:, :, :frm]) ** 2).mean()
    else:
        main = ((x.mean(dim=2) - y.mean(dim=2)) ** 2).mean()
    extra["swd"] = main.reshape(1, -1) * loss_gain
    if args.sparsity_loss_weight > 0:
        a = mpi[..., -1]
        sp = a.norm(dim=-1, p=1) / a.norm(dim=-1, p=2).clamp_min(1e-4)
        extra["sparsity"] = (sp.mean() / np.sqrt(D) * loss_gain).reshape(1, -1)
    if args.rgb_smooth_loss_weight > 0:
        extra["rgb_smooth"] = (_smooth(mpi[..., :-1], D) * loss_gain).reshape(1, -1)
    if args.a_smooth_loss_weight > 0:
        extra["a_smooth"] = (_smooth(mpi[..., -1], D) * loss_gain).reshape(1, -1)
    if args.density_loss_weight > 0:
        extra["density"] = (alpha - 1).abs().mean().reshape(1, -1)
    if getattr(args, "d_smooth_loss_weight", 0) > 0:                                    # MPV.py:385, 463-466, 539-551
        # disp = sum_k blend_weight_k / zbuf_k; zbuf of a planar mesh = view-space depth of the plane point under the pixel
        disp = (bw * _inv_depth(homos, h, w, ref_intrin_mpi, planedepth, extrins[0], pixel_center)[None]).sum(-1)
        dg = (disp[:, 1:, :-1] - disp[:, 1:, 1:]).abs() + (disp[:, :-1, 1:] - disp[:, 1:, 1:]).abs()
        extra["d_smooth"] = dg.mean().reshape(1, -1)
    return None, extra


def mpi_forward(stack, stack_mask, args, H, W, ref_extrin, ref_intrin, near, far, h, w, tar_extrins, tar_intrins,
                training=True, pixel_center=0.5, atlas_grid_h=None, quad_keep=None, tile=None, uv_noise_seeds=None):
    """uv_noise_seeds (one per view; add_uv_noise while training, MPI.py:519-522): the COLOUR samples are jittered by that seed's field (MO.uv_jitter_field);
    the loop mask is sampled at the plain positions and composited with the jittered samples' alphas (MPI.py:568-583 reads `uvs`, not the jittered `uvs_`).
    MPMesh.forward (MPI.py:596-652) for planar geometry.  stack (D,1,Hs,Ws,4) + stack_mask (D,1,Hs,Ws) or None; with atlas_grid_h the
    reference's atlas (1,4,Ah,Aw) + atlas_mask (1,1,Ah,Aw).  Returns (rgbl [B,3|4,h,w], extra)."""
    D = args.mpi_d if atlas_grid_h is not None else stack.shape[0]
    mpi_h, mpi_w, planedepth, ref_intrin_mpi, extrins = _geometry(args, D, H, W, ref_extrin, ref_intrin, near, far, tar_extrins,
                                                                  stack.shape[2:4])
    outs, mpis, alphas, masks, disps = [], [], [], [], []
    for b in range(len(extrins)):
        homos = _homos(ref_intrin_mpi, extrins[b:b + 1], tar_intrins[b:b + 1], planedepth)
        layers, cov = _layers(stack, homos, h, w, args, mpi_h, mpi_w, pixel_center, atlas_grid_h, quad_keep, tile=tile,
                              uv_noise_seed=0 if uv_noise_seeds is None else uv_noise_seeds[b])   # 1,h,w,D,4
        rgb, bw = MO.overcompose(layers[..., 3], layers[..., :3])
        alpha = bw.sum(-1)
        if len(args.bg_color) > 0:                                                                   # MPI.py:550-556
            r, g, b_ = map(float, args.bg_color.split('#'))
            rgb = rgb * alpha[..., None] + torch.tensor([r, g, b_]).type_as(rgb)[None, None, None] * (1 - alpha[..., None])
        # depth map (MPI.py:483-485, 558-561): disparity of every hit normalised to [0,1] between far and near, blended
        bwd = bw / alpha.clamp_min(1e-10)[..., None] if getattr(args, "normalize_blendweight_fordepth", False) else bw
        inv_z = _inv_depth(homos, h, w, ref_intrin_mpi, planedepth, extrins[b], pixel_center)
        disps.append((bwd * ((inv_z - 1 / far) / (1 / near - 1 / far))[None]).sum(-1))
        if stack_mask is not None:                                                                   # MPI.py:568-583
            m = stack_mask if atlas_grid_h is not None else stack_mask[..., None]
            lab_layers, _ = _layers(m if atlas_grid_h is not None else torch.cat([m, m, m, m], -1), homos, h, w, args, mpi_h, mpi_w,
                                    pixel_center, atlas_grid_h, quad_keep,
                                    acts=(torch.sigmoid, torch.sigmoid) if atlas_grid_h is not None else ("sigmoid", "sigmoid"), tile=tile)
            lab_layers = lab_layers[..., :1]
            label, _ = MO.overcompose(layers[..., -1].detach(), lab_layers)
            rgb = torch.cat([rgb, label], dim=-1)
            masks.append(_slots(lab_layers[..., 0], cov))
        outs.append(rgb)
        mpis.append(MO.layers_to_slots(layers, cov))
        alphas.append(alpha)
    rgbl = torch.cat(outs, 0).permute(0, 3, 1, 2)
    kmax = max(m.shape[3] for m in mpis)             # the B views are rasterised in one call: K = the deepest pixel of the batch
    mpi = torch.cat([torch.nn.functional.pad(m, (0, 0, 0, kmax - m.shape[3])) for m in mpis], 0)
    alpha = torch.cat(alphas, 0)
    extra = {}
    if training:
        if args.sparsity_loss_weight > 0:
            a = mpi[..., -1]
            sp = a.norm(dim=-1, p=1) / a.norm(dim=-1, p=2).clamp_min(1e-6)
            extra["sparsity"] = (sp.mean() / np.sqrt(D)).reshape(1, -1)
        if args.rgb_smooth_loss_weight > 0:                                                          # MPI.py:605-612
            extra["rgb_smooth"] = _smooth(mpi[..., :-1], D).reshape(1, -1)
        if args.a_smooth_loss_weight > 0:                                                            # MPI.py:614-620
            extra["a_smooth"] = _smooth(mpi[..., -1], D).reshape(1, -1)
        if getattr(args, "d_smooth_loss_weight", 0) > 0:                                             # MPI.py:622-637
            disp = torch.cat(disps, 0)
            dg = (disp[:, 1:, :-1] - disp[:, 1:, 1:]).abs() + (disp[:, :-1, 1:] - disp[:, 1:, 1:]).abs()
            c = rgbl[:, :3]
            edge = (c[..., 1:, :-1] - c[..., 1:, 1:]).abs().sum(dim=1) + (c[..., :-1, 1:] - c[..., 1:, 1:]).abs().sum(dim=1)
            extra["d_smooth"] = (dg * (-edge * args.edge_scale + 1).clamp_min(0)).mean().reshape(1, -1)
        if getattr(args, "l_smooth_loss_weight", 0) > 0 and masks:                                   # MPI.py:639-645
            lm = torch.cat([torch.nn.functional.pad(m, (0, kmax - m.shape[3])) for m in masks], 0)
            extra["l_smooth"] = _smooth(lm, D).reshape(1, -1)
        if args.density_loss_weight > 0:
            extra["density"] = (alpha - 1).abs().mean().reshape(1, -1)
    return rgbl, extra
