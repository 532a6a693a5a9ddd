"""Stage-1 driver loop on the device: the caller of the hot path at T = 1 (after §8f-3).

Mirrors the training part of /root/reference/train_3d.py: the crop dataset over one still image per view (:20-95), `run_iter` (:189-250:
MPMesh.forward, the loop-mask entropy term, the scale-invariant MSE, the weighted regularisers, backward, step) and the epoch loop
(:273-318) with its per-epoch tasks -- `sparsify_faces` + a new optimiser at `sparsify_epoch` (:282-285), the quadratic ramp of the
density weight (:292-293), the per-iteration learning rate (:303-306), a checkpoint `{'epoch_i', 'network_state_dict'}` every
`i_weights` epochs (:311-318).  Differences: videos are tensors already on the device (no cv2 / DataLoader: crops are views of resident
tensors), logging is left to the caller through `on_step`, and the image + loop-mask loss runs in the fused kernel pair of
`MPI.image_and_loop_loss` (same values, tests/test_gpu_mpv.py).

`vid2img` / `compute_loopable_mask` restate the reference's cv2 arithmetic in torch (bilinear resize at half-pixel centres, the 5 x 5
Gaussian with cv2's default sigma and reflect-101 border): cv2 is absent from this image, so these two are UNPINNED restatements -- a
caller with the reference's own images / masks passes them in (`images=`, `dynmasks=`) and nothing here is recomputed.
"""
import os

import numpy as np
import torch
import torch.nn.functional as torchf

from .MPV import get_new_intrin
from .train_3dvid import generate_patchinfo, pose2extrin_torch


# OpenCV's fixed kernels for odd k <= 7 when sigma is not given (cv::getGaussianKernel's small_gaussian_tab; from memory of its source -- cv2 is
# absent from this image, so this is unpinned like everything on the OUT-OF-SCOPE data side)
_SMALL_GAUSSIAN = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
                   7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}


def _gaussian_blur(x, ksize):
    """cv2.GaussianBlur(x, (k, k), 0) on [N,C,h,w]: separable, BORDER_REFLECT_101; the kernel is OpenCV's fixed table for odd k <= 7, else
    sigma = 0.3 ((k - 1) / 2 - 1) + 0.8.  (float arithmetic: cv2 filters uint8 images in fixed point, a difference of at most a grey level.)"""
    r = ksize // 2
    if ksize in _SMALL_GAUSSIAN:
        k = torch.tensor(_SMALL_GAUSSIAN[ksize], dtype=x.dtype, device=x.device)
    else:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
        t = torch.arange(-r, r + 1, dtype=x.dtype, device=x.device)
        k = torch.exp(-(t * t) / (2 * sigma * sigma))
        k = k / k.sum()
    c = x.shape[1]
    x = torchf.pad(x, (r, r, r, r), mode="reflect")
    x = torchf.conv2d(x, k.view(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)
    return torchf.conv2d(x, k.view(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)


def compute_loopable_mask(vid, eps=15 / 255, factor=2):
    """utils.py:337-364 on a clip [F,3,h,w] in [0,1] -> bool [h,w]: pixels that both rise and fall by more than eps over the clip
    (running min / max at half resolution), labels smoothed by a 5 x 5 Gaussian, resized back, argmax == loopable."""
    h, w = vid.shape[-2:]
    down = torchf.interpolate(vid, scale_factor=1.0 / factor, mode="bilinear", align_corners=False, recompute_scale_factor=False)
    minval = maxval = down[0]
    rises = torch.zeros_like(down[0], dtype=torch.bool)
    falls = torch.zeros_like(down[0], dtype=torch.bool)
    for im in down[1:]:
        minval = torch.minimum(minval, im)
        maxval = torch.maximum(maxval, im)
        rises |= (im - minval) > eps
        falls |= (maxval - im) > eps
    unchanging = (~rises & ~falls).all(dim=0)
    unloopable = (rises ^ falls).any(dim=0)
    loopable = ~(unchanging | unloopable)
    label = torch.stack([loopable, unloopable, unchanging]).to(vid.dtype)[None] * 255
    label = torchf.interpolate(_gaussian_blur(label, 5), size=(h, w), mode="bilinear", align_corners=False)
    return label[0].argmax(dim=0) == 0


def vid2img(vid, mode="average"):
    """train_3d.py:55-83: the still image a view is trained on, from its clip [F,3,h,w] in [0,1] -> [3,h,w]."""
    if mode == "median":      # np.median (train_3d.py:58): the mean of the two middle frames of an even-length clip, not torch's lower median
        return torch.quantile(vid, 0.5, dim=0)
    if mode == "average":
        return vid.mean(dim=0)
    if mode == "first":
        return vid[0]
    if mode.startswith("dynamic"):      # emphasise the dynamics (configs/mpi_base.txt:10)
        k = mode[len("dynamic"):]
        k = 1.0 if len(k) == 0 else float(k)
        weight = (vid - vid.mean(dim=0, keepdim=True)).norm(dim=1, keepdim=True)
        weight = (k * weight + (1 - k)).clamp(1e-10, 999999)
        return (vid * weight).sum(dim=0) / weight.sum(dim=0)
    if mode.startswith("blur"):
        b = mode[len("blur"):]
        b = 11 if len(b) == 0 else int(b)
        vb = _gaussian_blur(vid, b)
        weight = ((vb - vb.mean(dim=0, keepdim=True)).norm(dim=1, keepdim=True) * 3).clamp(0.001, 3)
        return (vb * weight).sum(dim=0) / weight.sum(dim=0)
    raise RuntimeError(f"Unrecognized vid2img_mode={mode}")


class MVPatchDataset:
    """train_3d.py:20-95 on resident tensors.  `videos`: list of [F,3,h_raw,w_raw] float tensors in [0,1] (any device); `poses`
    [V,3|4,4], `intrins` [V,3,3] for the raw resolution.  Items are (w_start, h_start, pose, intrin, crop [3,ph,pw], mask [ph,pw]).
    `images` / `dynmasks`: the views' still images [3,h,w] / loopable masks [h,w] at `resize_hw`, if the caller has them already."""

    def __init__(self, resize_hw, videos, patch_size, patch_stride, poses, intrins, mode="average", images=None, dynmasks=None):
        h_raw, w_raw = videos[0].shape[-2:]
        self.h, self.w = resize_hw
        self.v = len(videos)
        self.poses = poses.clone().cpu()
        self.intrins = intrins.clone().cpu()
        self.intrins[:, :2] *= torch.tensor([self.w / w_raw, self.h / h_raw]).reshape(1, 2, 1).type_as(self.intrins)
        self.patch_h_size, self.patch_w_size = patch_size
        self.mode = mode
        if self.h * self.w < self.patch_h_size * self.patch_w_size:
            wh, pad_info = torch.tensor([[0, 0]]).long(), [0, 0, 0, 0]
            self.patch_h_size, self.patch_w_size = self.h, self.w
        else:
            wh, pad_info = generate_patchinfo(self.h, self.w, patch_size, patch_stride)
        self.patch_wh_start = wh[None].expand(self.v, -1, 2).reshape(-1, 2)
        self.view_index = np.arange(self.v)[:, None].repeat(wh.shape[0], axis=1).reshape(-1).tolist()
        self.images, self.dynmask = [], []
        for vi, vid in enumerate(videos):
            if vid.shape[-2:] != (self.h, self.w):      # cv2.resize (bilinear, no antialias) of the reference
                vid = torchf.interpolate(vid, size=(self.h, self.w), mode="bilinear", align_corners=False)
            img = images[vi] if images is not None else vid2img(vid, mode)
            ma = dynmasks[vi] if dynmasks is not None else compute_loopable_mask(vid)
            self.images.append(img.to(vid.dtype))
            self.dynmask.append(ma.to(img.dtype))
        # NB: unlike the video dataset (train_3dvid.py:52) the reference does NOT pad the stills: a crop that leaves the frame is cut short
        print(f"Dataset: generate {len(self)} patches for training, pad {pad_info} to videos")

    def __len__(self):
        return len(self.patch_wh_start)

    def __getitem__(self, item):
        w_start, h_start = (int(v) for v in self.patch_wh_start[item])
        vi = self.view_index[item]
        intrin = get_new_intrin(self.intrins[vi], h_start, w_start).float()
        crop = self.images[vi][..., h_start:h_start + self.patch_h_size, w_start:w_start + self.patch_w_size]
        mask = self.dynmask[vi][h_start:h_start + self.patch_h_size, w_start:w_start + self.patch_w_size]
        return w_start, h_start, self.poses[vi], intrin, crop, mask


def run_iter(nerf, optimizer, item, args, device):
    """train_3d.py:189-250 without the logging -> (loss, img_loss, loop_loss, {regulariser terms}) detached."""
    from .MPI import image_and_loop_loss
    _, _, pose, intrin, crop, mask = item
    b_extrin = pose2extrin_torch(pose[None].cpu())            # poses stay on the host: the module forms the plane homographies there
    b_intrin = intrin[None].cpu()
    b_rgbs = crop[None].to(device)
    b_loopmask = mask[None].to(device)
    patch_h, patch_w = b_rgbs.shape[-2:]
    if getattr(args, "add_intrin_noise", False):
        b_intrin = b_intrin.clone()
        b_intrin[:, :2, 2] += torch.rand(2).type_as(b_intrin) - 0.5     # half pixel
    nerf.train()
    if hasattr(optimizer, "acknowledge_fused_backward"):
        optimizer.acknowledge_fused_backward()
    learn_mask = bool(getattr(args, "learn_loop_mask", False))
    module = getattr(nerf, "module", nerf)
    if hasattr(module, "objective") and not getattr(args, "generic_objective", False) and getattr(module, "args", args) is not args:
        # the fused objective reads every *_loss_weight from module.args; this loop (and train()'s density ramp) reads and mutates `args`: a
        # model built from a COPY of the namespace would silently train with stale weights on one of the two paths
        raise RuntimeError("run_iter: the model was built with a different args object than the one the driver mutates; pass the same "
                           "namespace to both (or args.generic_objective = True for the reference's spelling of the objective)")
    if hasattr(module, "objective") and not getattr(args, "generic_objective", False):
        # render + every loss term + their weighted total with the scalar head fused (MPMesh.objective: same values and gradients as the
        # spelling below, tests/test_gpu_stage1_driver.py); args.generic_objective keeps the reference's spelling for A/B
        loss, img_loss, loop_loss, extra_losses = module.objective(patch_h, patch_w, b_extrin, b_intrin, b_rgbs, b_loopmask if learn_mask else None,
                                                                   scale_invariant=bool(getattr(args, "scale_invariant", False)))
    else:
        rgbl, extra = nerf(patch_h, patch_w, b_extrin, b_intrin)
        img_loss, loop_loss = image_and_loop_loss(rgbl, b_rgbs, b_loopmask if learn_mask else None,
                                                  scale_invariant=bool(getattr(args, "scale_invariant", False)))
        args_var = vars(args)
        from .train_3dvid import weighted_total
        mains = [img_loss] + ([loop_loss] if torch.is_tensor(loop_loss) else [])
        loss, _, extra_losses = weighted_total(mains, extra, lambda k: args_var.get(f"{k}_loss_weight", 0))
        img_loss, loop_loss = img_loss.detach(), (loop_loss.detach() if torch.is_tensor(loop_loss) else loop_loss)
        extra_losses = {k: v.detach() for k, v in extra_losses.items()}
    optimizer.zero_grad()
    from .train_3dvid import unit_grad
    loss.backward(unit_grad(module, loss))
    if hasattr(module, "post_backward"):
        module.post_backward()
    optimizer.step()
    return loss.detach(), img_loss, loop_loss, extra_losses


def train(nerf, args, videos, poses, intrins, H, W, device="cuda:0", on_step=None, generator=None, save_dir=None, start_epoch=0,
          images=None, dynmasks=None):
    """train_3d.py:262-318: `args.N_iters` epochs over the shuffled crops of all views.  `nerf` is the MPMesh (or an object exposing
    `.module`); returns {'iters', 'epochs', 'sparsified_at'}.  No host synchronisation inside the loop unless `on_step` reads values.
    `save_dir`: every `args.i_weights` epochs write `epoch_{epoch:04d}.tar` with the reference's keys; `start_epoch`: resume (:278-279)."""
    module = getattr(nerf, "module", nerf)
    optimizer = module.get_optimizer()
    old_density_loss_weight = float(getattr(args, "density_loss_weight", 0.0))
    dataset = MVPatchDataset((H, W), videos, (args.patch_h_size, args.patch_w_size), (args.patch_h_stride, args.patch_w_stride),
                             poses, intrins, getattr(args, "vid2img_mode", "average"), images=images, dynmasks=dynmasks)
    iter_total_step, sparsified_at = 0, None
    for epoch_i in range(int(args.N_iters)):
        if epoch_i < start_epoch:
            continue
        if epoch_i == int(getattr(args, "sparsify_epoch", -1)):
            module.sparsify_faces(erode_num=int(getattr(args, "sparsify_erode", 2)), alpha_thresh=float(getattr(args, "sparsify_alpha_thresh", 0.03)))
            optimizer = module.get_optimizer()
            sparsified_at = epoch_i
        pct = float(np.clip(epoch_i / (int(getattr(args, "density_loss_epoch", 0)) + 1), 0, 1))
        args.density_loss_weight = pct * pct * old_density_loss_weight
        for item_i in torch.randperm(len(dataset), generator=generator).tolist():           # DataLoader(dataset, 1, shuffle=True)
            if hasattr(module, "update_step"):
                module.update_step(iter_total_step)
            for (_, new_lrate), group in zip(module.get_lrate(iter_total_step), optimizer.param_groups):
                group['lr'] = new_lrate
            out = run_iter(nerf, optimizer, dataset[item_i], args, device)
            if on_step is not None:
                on_step(epoch_i, iter_total_step, *out)
            iter_total_step += 1
        if save_dir is not None and (epoch_i + 1) % max(int(getattr(args, "i_weights", 20000)), 1) == 0:
            torch.save({'epoch_i': epoch_i, 'network_state_dict': module.state_dict()}, os.path.join(save_dir, f'epoch_{epoch_i:04d}.tar'))
    args.density_loss_weight = old_density_loss_weight
    return {"iters": iter_total_step, "epochs": int(args.N_iters) - int(start_epoch), "sparsified_at": sparsified_at}
