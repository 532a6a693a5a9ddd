"""CPU oracle for MPMeshVid.forward (planar geometry).  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/MPV.py:477-556 (forward) on top of the pinned operator oracles (mpi_oracle / vid_oracle).
MPV.py itself cannot be imported here (pytorch3d, cv2, imageio, torchvision are absent) so this module is
"parity unpinned" at the pytorch3d boundary: sampling positions are pinned through compute_homography (== ray-plane
intersection), compositing through overcompose, the loss through G8; the +0.5 pixel centre and hard-cut borders are
parameters.  A third unpinned item is not a constant but a semantic: the layer tensor `mpi` the regularisers read
is HIT-SLOT indexed (slot k = the k-th nearest face the rasteriser hit at that pixel, MPV.py:386-392, 441-449, utils.py:64-69),
restated by mpi_oracle.layers_to_slots from the source; the un-vendored rasteriser's z-sort is taken to be depth order.
"""
import numpy as np
import torch

from . import mpi_oracle as MO
from . import vid_oracle as VO


def mpv_forward(stack, args, H, W, ref_extrin, ref_intrin, near, far, h, w, tar_extrins, tar_intrins, ts=None, res=None,
                losscfg=None, training=True, pixel_center=0.5):
    """stack (D,T,mpi_h,mpi_w,4).  Returns (rgb [T',3,h,w] or None, extra dict) like MPV.py:553-556."""
    D, T, mpi_h, mpi_w, _ = stack.shape
    ref_extrin = torch.as_tensor(ref_extrin)
    ref_intrin = torch.as_tensor(ref_intrin).float()
    planedepth = MO.make_depths(D, near, far).flip(0)                                   # MPV.py:51
    H_start, W_start = (mpi_h - H) // 2, (mpi_w - W) // 2                               # MPV.py:55
    ref_intrin_mpi = MO.get_new_intrin(ref_intrin, -H_start, -W_start)
    extrins = tar_extrins @ ref_extrin[None].inverse().to(tar_extrins.dtype)            # MPV.py:478
    if ts is None:
        ts = torch.arange(T)
    eye = torch.eye(4, dtype=extrins.dtype)[None]
    normal = torch.tensor([0., 0., 1.], dtype=extrins.dtype).expand(1, D, 3)
    homos = MO.compute_homography(eye, ref_intrin_mpi[None].to(extrins.dtype), extrins, tar_intrins, normal,
                                  planedepth[None].to(extrins.dtype))[0].float()
    spec = MO.RenderSpec(pixel_center=pixel_center, coord_mode="affine", border="hardcut", act_order="post",
                         rgb_act=args.rgb_activate, alpha_act=args.alpha_activate)
    rgb, alpha, bw, mpi = MO.render_planes(stack[:, ts], homos, h, w, spec, return_layers=True)
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
    if loss_name in ("gpnn", "gpnn_lm"):
        cfg.pop("dist_fn", None)
        if loss_name == "gpnn":
            cfg = {"patch_size": 7, "patcht_size": 7, "stride": 1, "stridet": 1, "rou": 0, "scaling": 0.2, **cfg}
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
        sm = mpi[..., :-1]
        denorm = sm.shape[-2] / D
        sx = (sm[:, :, :-1] - sm[:, :, 1:]).abs().mean()
        sy = (sm[:, :-1] - sm[:, 1:]).abs().mean()
        extra["rgb_smooth"] = ((sx + sy) * (loss_gain * denorm)).reshape(1, -1)
    if args.a_smooth_loss_weight > 0:
        sm = mpi[..., -1]
        denorm = sm.shape[-1] / D
        sx = (sm[:, :, :-1] - sm[:, :, 1:]).abs().mean()
        sy = (sm[:, :-1] - sm[:, 1:]).abs().mean()
        extra["a_smooth"] = ((sx + sy) * (loss_gain * denorm)).reshape(1, -1)
    if args.density_loss_weight > 0:
        extra["density"] = (alpha - 1).abs().mean().reshape(1, -1)
    if getattr(args, "d_smooth_loss_weight", 0) > 0:                                    # MPV.py:385, 463-466, 539-551
        # disp = sum_k blend_weight_k / zbuf_k; zbuf of a planar mesh = view-space depth of the plane point under the pixel
        xs, ys = MO._homography_source_coords(h, w, homos.double(), pixel_center)          # plane pixels [D,h,w]
        ray = torch.inverse(ref_intrin_mpi.double()) @ torch.stack([xs, ys, torch.ones_like(xs)], -1)[..., None]
        P = ray[..., 0] * planedepth.double()[:, None, None, None]
        E = extrins[0].double()
        z = (P * E[2, :3]).sum(-1) + E[2, 3]
        disp = (bw * (1.0 / z).float().permute(1, 2, 0)[None]).sum(-1)
        dg = (disp[:, 1:, :-1] - disp[:, 1:, 1:]).abs() + (disp[:, :-1, 1:] - disp[:, 1:, 1:]).abs()
        extra["d_smooth"] = dg.mean().reshape(1, -1)
    return None, extra


def mpi_forward(stack, stack_mask, args, H, W, ref_extrin, ref_intrin, near, far, h, w, tar_extrins, tar_intrins,
                training=True, pixel_center=0.5):
    """MPMesh.forward (MPI.py:596-652) for planar geometry.  stack (D,1,mpi_h,mpi_w,4), stack_mask (D,1,mpi_h,mpi_w) or None.
    Returns (rgbl [B,3|4,h,w], extra)."""
    D, _, mpi_h, mpi_w, _ = stack.shape
    ref_extrin = torch.as_tensor(ref_extrin)
    ref_intrin = torch.as_tensor(ref_intrin).float()
    planedepth = MO.make_depths(D, near, far).flip(0)
    H_start, W_start = (mpi_h - H) // 2, (mpi_w - W) // 2
    ref_intrin_mpi = MO.get_new_intrin(ref_intrin, -H_start, -W_start)
    extrins = tar_extrins @ ref_extrin[None].inverse().to(tar_extrins.dtype)
    spec = MO.RenderSpec(pixel_center=pixel_center, coord_mode="affine", border="hardcut", act_order="post",
                         rgb_act=args.rgb_activate, alpha_act=args.alpha_activate)
    outs, mpis, alphas = [], [], []
    for b in range(len(extrins)):
        eye = torch.eye(4, dtype=extrins.dtype)[None]
        normal = torch.tensor([0., 0., 1.], dtype=extrins.dtype).expand(1, D, 3)
        homos = MO.compute_homography(eye, ref_intrin_mpi[None].to(extrins.dtype), extrins[b:b + 1], tar_intrins[b:b + 1], normal,
                                      planedepth[None].to(extrins.dtype))[0].float()
        rgb, alpha, bw, mpi = MO.render_planes(stack, homos, h, w, spec, return_layers=True)        # mpi: 1,h,w,D,4
        if len(args.bg_color) > 0:
            r, g, b_ = map(float, args.bg_color.split('#'))
            rgb = rgb * alpha[..., None] + torch.tensor([r, g, b_]).type_as(rgb)[None, None, None] * (1 - alpha[..., None])
        if stack_mask is not None:                                                                   # MPI.py:568-583
            mspec = MO.RenderSpec(pixel_center=pixel_center, coord_mode="affine", border="hardcut", act_order="post",
                                  rgb_act="sigmoid", alpha_act="none")
            m = stack_mask[..., None]
            lab_layers = MO.render_planes(torch.cat([m, m, m, torch.zeros_like(m)], -1), homos, h, w, mspec, return_layers=True)[3]
            label, _ = MO.overcompose(mpi[..., -1].detach(), lab_layers[..., :1])
            rgb = torch.cat([rgb, label], dim=-1)
        outs.append(rgb)
        mpis.append(mpi)
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
            sm = mpi[..., :-1]
            denorm = sm.shape[-2] / D
            extra["rgb_smooth"] = (((sm[:, :, :-1] - sm[:, :, 1:]).abs().mean() + (sm[:, :-1] - sm[:, 1:]).abs().mean()) * denorm).reshape(1, -1)
        if args.a_smooth_loss_weight > 0:                                                            # MPI.py:614-620
            sm = mpi[..., -1]
            denorm = sm.shape[-1] / D
            extra["a_smooth"] = (((sm[:, :, :-1] - sm[:, :, 1:]).abs().mean() + (sm[:, :-1] - sm[:, 1:]).abs().mean()) * denorm).reshape(1, -1)
        if args.density_loss_weight > 0:
            extra["density"] = (alpha - 1).abs().mean().reshape(1, -1)
    return rgbl, extra
