"""Offline renderer on the device: the caller of the hot path in eval mode (scripts/script_render_video.py).

Mirrors /root/reference/scripts/script_render_video.py: the camera side of `dataloader.load_llff_data` it renders from -- LLFF
`poses_bounds.npy` -> axis flip, bound rescale, `recenter_poses`, the spiral of `render_path_spiral`, the intrinsics
(`dataloader.py:60-134, 205-260`) --, the view / time selection of its `--v` / `--t` switches (`:47-85`), the reference camera of the
model (`:84-87`), and the frame loop `nerf(H, W, pose, intrin, t)` in eval mode (`:129-139`).  Differences: no disk I/O besides the pose
file (frames are returned as a uint8 device tensor; writing PNG / MP4 is the caller's), and consecutive frames that share ONE camera -- the
`--v` modes render mpv_frm_num times of a fixed view -- go through the renderer as ONE call with `ts` a vector instead of one launch per
frame (the fused kernel takes T' frames of a view at once; same pixels).

The pose arithmetic is numpy like the reference's and pinned to it by golden G18 (`tests/golden/make_golden_r05.py` imports the reference's
dataloader.py with name-only stand-ins for cv2 / imageio).
"""
import os

import numpy as np
import torch


def _unit(v):
    return v / np.linalg.norm(v)


def viewmatrix(z, up, pos):
    """dataloader.py:209-215: camera-to-world 3 x 4 looking along z: columns (right, true up, forward, position)."""
    fwd = _unit(z)
    right = _unit(np.cross(up, fwd))
    return np.stack([right, _unit(np.cross(fwd, right)), fwd, pos], axis=1)


def poses_avg(poses):
    """dataloader.py:218-226: the views' average camera (mean position, summed forward / up axes); its last column is pose 0's last column
    (the (H, W, f) column of 3 x 5 poses)."""
    pos = poses[:, :3, 3].mean(axis=0)
    fwd, up = poses[:, :3, 2].sum(axis=0), poses[:, :3, 1].sum(axis=0)
    return np.concatenate([viewmatrix(fwd, up, pos), poses[0, :3, -1:]], axis=1)


def _homog(m34):
    return np.concatenate([m34, np.broadcast_to(np.array([0, 0, 0, 1.0], m34.dtype), m34.shape[:-2] + (1, 4))], axis=-2)


def recenter_poses(poses):
    """dataloader.py:235-246: every pose expressed in the average camera's frame (the (H, W, f) column untouched)."""
    out = poses + 0
    world_to_avg = np.linalg.inv(_homog(poses_avg(poses)[:3, :4]))
    out[:, :3, :4] = (world_to_avg @ _homog(poses[:, :3, :4]))[:, :3, :4]
    return out


def render_path_spiral(c2w, up, rads, focal, zrate, zdelta, rots, N):
    """dataloader.py:249-260: N cameras on a spiral around c2w, all looking at the point `focal` in front of it."""
    scale = np.append(np.asarray(rads, dtype=np.float64), 1.0)
    thetas = np.linspace(0.0, 2.0 * np.pi * rots, N + 1)[:-1]
    local = np.stack([np.cos(thetas), -np.sin(thetas), (np.cos(thetas * zrate) * zdelta) ** 2, np.ones_like(thetas)], axis=1) * scale
    centres = local @ c2w[:3, :4].T
    return np.stack([viewmatrix(np.array([0, 0, focal]) - c, up, c) for c in centres])


def _llff_cameras(table, shrink):
    """rows of a poses_bounds table [V,17] -> (camera-to-world [V,3,4] with columns (right, up, back, position), (H, W, f) per view [V,3],
    (near, far) per view [V,2]).  LLFF writes the rotation's columns as (down, right, back) (dataloader.py:79-80 reorders them); `shrink`
    divides the image size and the focal length (images loaded at 1 / factor of their resolution, dataloader.py:24-28)."""
    table = np.asarray(table)
    mats = table[:, :15].reshape(-1, 3, 5)
    c2w = np.stack([mats[:, :, 1], mats[:, :, 0], -mats[:, :, 2], mats[:, :, 3]], axis=2)
    return c2w.astype(np.float32), (mats[:, :, 4] / shrink).astype(np.float32), table[:, 15:17].astype(np.float32)


def _pinhole(hwf):
    """(H, W, f) rows [N,3] -> intrinsics [N,3,3] with the principal point at the image centre (dataloader.py:121-131)."""
    K = np.zeros((len(hwf), 3, 3), dtype=hwf.dtype)
    K[:, 0, 0] = K[:, 1, 1] = hwf[:, 2]
    K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = 0.5 * hwf[:, 1], 0.5 * hwf[:, 0], 1
    return K


def load_llff_poses(poses_bounds, factor=8, recenter=True, bd_factor=(1, 1), render_frm=120, render_scaling=1.):
    """The camera side of `dataloader.load_llff_data` (dataloader.py:60-134) without the images: `poses_bounds` = the array of
    poses_bounds.npy (or its path) -> (poses [V,3,4], intrins [V,3,3], bds [2], render_poses [N,3,4], render_intrins [N,3,3]), float32.
    Steps: LLFF axes -> (right, up, back); the scene rescaled so that the nearest depth bound is 1; optionally every pose expressed in the
    average camera's frame; a two-turn spiral of `render_frm` cameras around the average camera, looking at a point between 0.9 x the near
    and 5 x the far bound (3 : 1 in inverse depth), with radii = 0.8 x the largest camera offsets per axis."""
    if isinstance(poses_bounds, (str, os.PathLike)):
        poses_bounds = np.load(poses_bounds)
    c2w, hwf, depth_bounds = _llff_cameras(poses_bounds, 1 if factor is None else factor)
    bds = np.array([depth_bounds.min(), depth_bounds.max()], dtype=np.float32)
    unit = np.float32(1.) / bds[0]                      # the nearest bound becomes 1
    c2w[:, :, 3] *= unit
    bds *= unit
    if bd_factor is not None:
        bds *= bd_factor
    cams = np.concatenate([c2w, hwf[:, :, None]], axis=2)                  # [V,3,5]: the helpers carry the (H, W, f) column along
    if recenter:
        cams = recenter_poses(cams)
    centre = poses_avg(cams)
    look_near, look_far = bds.min() * .9, bds.max() * 5.
    w_far = .75
    spiral = render_path_spiral(centre, _unit(cams[:, :3, 1].sum(0)), rads=np.abs(cams[:, :3, 3]).max(0) * 0.8 * render_scaling,
                                focal=1. / ((1. - w_far) / look_near + w_far / look_far), zrate=.5, zdelta=look_near * .2, rots=2, N=render_frm)
    cams = cams.astype(np.float32)
    intrins = _pinhole(cams[:, :, 4])
    return cams[:, :, :4], intrins, bds, spiral.astype(np.float32), np.repeat(intrins[:1], len(spiral), axis=0)


def pose2extrin_np(pose):
    """utils.py:203-209: camera-to-world [...,3|4,4] -> world-to-camera [...,4,4]."""
    if pose.shape[-2] == 3:
        bottom = np.zeros_like(pose[..., :1, :])
        bottom[..., 3] = 1
        pose = np.concatenate([pose, bottom], axis=-2)
    return np.linalg.inv(pose)


def default_render_frames(mpv_frm_num, f=-1):
    """script_render_video.py:33: the number of spiral poses, a whole number of loops."""
    return f if f > 0 else (120 // mpv_frm_num + 1) * mpv_frm_num


def select_views_times(render_poses, render_intrins, poses, intrins, mpv_frm_num, v="", t="", test_view_idx=""):
    """script_render_video.py:47-85: which camera and which frame index every output frame takes.
    v: '' the spiral, 'r#' the #-th spiral pose held fixed, '#' the #-th training view held fixed, 'test' the first test view;
    t: '' frame i of the loop for output i, '#,#,#' those entries, '#:#[,#:#]' ranges (end excluded, descending allowed), '#' one entry.
    -> (view_poses [N,3,4], view_intrins [N,3,3], render_t int array [N])."""
    view_poses, view_intrins = render_poses.copy(), render_intrins.copy()
    render_t = np.arange(len(render_poses)) % mpv_frm_num
    if v == 'test':
        v = test_view_idx.split(',')[0]
    if len(v) > 0:
        render_t = render_t[:mpv_frm_num]
        if v[0] == 'r':
            i = int(v[1:])
            view_poses[:] = view_poses[i:i + 1]
            view_intrins[:] = render_intrins[i:i + 1]
        else:
            i = int(v)
            view_poses[:] = poses[i:i + 1]
            view_intrins[:] = intrins[i:i + 1]
    if len(t) > 0:
        if ',' in t and ':' not in t:
            render_t = render_t[list(map(int, t.split(',')))]
        elif ':' in t:
            parts = []
            for slic in t.split(','):
                start, end = list(map(int, slic.split(':')))
                parts.append(np.arange(start, end, 1 if start <= end else -1))
            render_t = np.concatenate(parts)
        else:
            render_t = render_t[[int(t)]]
    return view_poses[:len(render_t)], view_intrins[:len(render_t)], render_t


def reference_camera(poses, intrins, bds):
    """script_render_view.py:84-87 -> (ref_extrin [4,4], ref_intrin [3,3], near, far) the model is built with."""
    ref_pose = poses_avg(poses)[:, :4]
    return pose2extrin_np(ref_pose), intrins[0], float(bds.min()), float(bds.max())


def to8b(x):
    """utils.py: (255 * clip(x, 0, 1)).astype(uint8), on the device."""
    return (255 * x.clamp(0, 1)).to(torch.uint8)


def _render_frames_in_place(module, H, W, view_extrins, view_intrins, render_t, chunk):
    """render_frames for a dense model on the device: the plane homographies of every DISTINCT camera of the path are formed up front (the
    module's own `plane_homographies`, the same bits as its forward) and uploaded in ONE copy, every frame -- or run of consecutive frames of
    one camera -- is rendered where it lies in the clip (render.render_frame_run: no gather of stack[:, ts]), straight into a chunk buffer
    that is converted to uint8 once (a sparsified model with its quad map; a packed one through its block table).  None when the model is
    not one this path serves (atlas_exact / CPU)."""
    from .render import render_frame_run, render_planes_packed
    packed = getattr(module, "packed", None)
    stack = module.stack_pool.data if packed is not None else getattr(module, "stack", None)
    if (stack is None or not stack.is_cuda or not stack.is_contiguous() or module.atlas_exact or module.training
            or module.args.bg_color == "random"):      # (a random background is a draw per call: the module's own path)
        return None
    if getattr(module, "_window_opt", None) is not None:
        module._flush_deferred_updates()
    n, T, dev = len(render_t), (packed.T if packed is not None else stack.shape[1]), stack.device
    if packed is not None:      # a packed model reads its pool through the block table (vl3d_render_fwd_packed): the frame indices go up once
        from .tiles import CULLED_ALPHA
        t_dev = torch.as_tensor(render_t.astype(np.int32)).pin_memory().to(dev, non_blocking=True)
    qk = module.quad_keep.to(torch.uint8).contiguous() if (module.is_sparse and getattr(module, "quad_keep", None) is not None) else None
    ref_inv = module._on(view_extrins.device, "ref_extrin")[None, ...].inverse().to(view_extrins.dtype)
    cams, cam_of = {}, []
    for i in range(n):
        key = (view_extrins[i].numpy().tobytes(), view_intrins[i].numpy().tobytes())
        if key not in cams:
            cams[key] = (len(cams), module.plane_homographies(view_extrins[i:i + 1] @ ref_inv, view_intrins[i:i + 1]))
        cam_of.append(cams[key][0])
    homos = torch.stack([h for _, h in sorted(cams.values(), key=lambda c: c[0])]).pin_memory().to(dev, non_blocking=True)      # [cameras, D, 3, 3]
    bg = None
    if len(module.args.bg_color) > 0:                                                        # MPV.py:455-461
        bg = torch.tensor([float(v) for v in module.args.bg_color.split('#')], dtype=torch.float32, device=dev)
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
    chunk = max(1, min(int(chunk), n))
    rgb, alpha = torch.empty((chunk, H, W, 3), dtype=torch.float32, device=dev), torch.empty((chunk, H, W), dtype=torch.float32, device=dev)
    c0 = 0
    while c0 < n:
        c1 = min(n, c0 + chunk)
        i = c0
        while i < c1:
            j = i + 1      # a run: one camera, consecutive frames of the clip
            while j < c1 and cam_of[j] == cam_of[i] and render_t[j] == render_t[j - 1] + 1:
                j += 1
            t0 = int(render_t[i])
            if not (0 <= t0 and t0 + (j - i) <= T):
                raise IndexError(f"frame index {t0} .. {t0 + j - i - 1} outside the clip of {T} frames")
            if packed is not None:
                render_planes_packed(packed, stack, render_t[i:j].tolist(), homos[cam_of[i]], H, W, module.spec, qk, CULLED_ALPHA,
                                     out=(rgb[i - c0:j - c0], alpha[i - c0:j - c0]), frames_dev=t_dev[i:j])
            else:
                render_frame_run(stack, t0, j - i, homos[cam_of[i]], H, W, module.spec, out=(rgb[i - c0:j - c0], alpha[i - c0:j - c0]), quad_keep=qk)
            i = j
        m = c1 - c0
        x = rgb[:m]
        if bg is not None:
            x = x * alpha[:m, ..., None] + bg[None, None, None] * (-alpha[:m, ..., None] + 1)
        out[c0:c1] = to8b(x)
        c0 = c1
    return out


@torch.no_grad()
def render_frames(nerf, H, W, view_extrins, view_intrins, render_t, max_batch=64, in_place=True):
    """script_render_video.py:129-139: `nerf(H, W, extrin, intrin, t)` in eval mode for every output frame -> uint8 [N,H,W,3] on the
    model's device.  A dense model on the device renders every frame where it lies in the clip, with the path's homographies uploaded once
    (`_render_frames_in_place`: 720p, D = 32, T = 50 along a spiral 1430 -> several thousand frames / s; `in_place=False` keeps the loop below).
    Otherwise runs of consecutive frames with one camera are rendered by ONE call of the module with `ts` a vector (at most `max_batch` frames:
    the frames of a call are resident together)."""
    module = getattr(nerf, "module", nerf)
    was_training = module.training
    nerf.eval()
    view_extrins = torch.as_tensor(np.asarray(view_extrins), dtype=torch.float32)
    view_intrins = torch.as_tensor(np.asarray(view_intrins), dtype=torch.float32)
    render_t = np.asarray(render_t).astype(np.int64)
    # script_render_video.py:129 loops over the POSES: a `--t` range longer than the camera path renders len(view_poses) frames
    render_t = render_t[:len(view_extrins)]
    res = None
    if in_place and hasattr(module, "plane_homographies") and len(render_t) > 0:
        res = _render_frames_in_place(module, H, W, view_extrins, view_intrins, render_t, max_batch)
    if res is None:
        out, i, n = [], 0, len(render_t)
        while i < n:
            j = i + 1
            while j < n and j - i < max_batch and torch.equal(view_extrins[j], view_extrins[i]) and torch.equal(view_intrins[j], view_intrins[i]):
                j += 1
            rgb, _ = nerf(H, W, view_extrins[i:i + 1], view_intrins[i:i + 1], torch.as_tensor(render_t[i:j]))
            out.append(to8b(rgb.permute(0, 2, 3, 1)))
            i = j
        res = torch.cat(out, 0)
    if was_training:
        nerf.train()
    return res


def render_video(nerf, args, poses_bounds, ckpt=None, v="", t="", f=-1, render_scaling=1., factor=None):
    """The whole of script_render_video.evaluate() but the files: poses -> selection -> (optional) checkpoint -> frames.
    `nerf`: an MPMeshVid built with `reference_camera(...)` of the same poses; `ckpt`: a path or a loaded dict with 'network_state_dict'."""
    frm = default_render_frames(args.mpv_frm_num, f)
    poses, intrins, bds, rposes, rintr = load_llff_poses(poses_bounds, factor=factor if factor is not None else getattr(args, "factor", 1),
                                                         recenter=True, bd_factor=(getattr(args, "near_factor", 1), getattr(args, "far_factor", 1)),
                                                         render_frm=frm, render_scaling=render_scaling)
    vp, vi, rt = select_views_times(rposes, rintr, poses, intrins, args.mpv_frm_num, v, t, getattr(args, "test_view_idx", ""))
    if ckpt is not None:
        sd = torch.load(ckpt, weights_only=False) if isinstance(ckpt, (str, os.PathLike)) else ckpt
        getattr(nerf, "module", nerf).init_from_mpi(sd['network_state_dict'])
    H, W = int(round(2 * intrins[0, 1, 2])), int(round(2 * intrins[0, 0, 2]))
    return render_frames(nerf, H, W, pose2extrin_np(vp), vi, rt)
