"""Stage-2 driver loop on the device: the caller of the hot path.

Mirrors the training part of /root/reference/train_3dvid.py: the crop dataset (:22-66), the pyramid schedule (:103-119),
the per-view loss configs (:160-189) and the epoch loop with lod / optimiser re-creation / per-epoch lr / adaptive lr
(:262-290, run_iter :214-255).  Differences: videos are tensors already on the device (no cv2 / DataLoader -- the crops are
views of resident tensors), and logging / checkpoint writing are left to the caller through `on_step`.
"""
from copy import deepcopy

import numpy as np
import torch
import torch.nn.functional as torchf

from .MPV import get_new_intrin


def generate_patchinfo(H_, W_, patch_size_, patch_stride_):
    """utils.py:115-134: crop origins (w_start, h_start) covering the frame, and the right/bottom padding that makes the
    last crop fit."""
    ph, pw = patch_size_
    sh, sw = patch_stride_
    hs = np.arange(0, H_ - ph + sh, sh)
    ws = np.arange(0, W_ - pw + sw, sw)
    gw, gh = np.meshgrid(ws, hs, indexing="ij")          # w varies slowest, like np.meshgrid(h, w)[::-1] of the reference
    wh = torch.tensor(np.stack([gw, gh], axis=-1).reshape(-1, 2))
    H_pad, W_pad = int(hs.max() + ph - H_), int(ws.max() + pw - W_)
    assert sh > H_pad >= 0 and sw > W_pad >= 0, "bug occurs!"
    return wh, [0, W_pad, 0, H_pad]


def pose2extrin_torch(pose):
    """utils.py:211-219."""
    if pose.shape[-2] == 3:
        bottom = torch.zeros_like(pose[..., :1, :])
        bottom[..., 3] = 1
        pose = torch.cat([pose, bottom], dim=-2)
    return torch.inverse(pose)


def pyramid_schedule(args, H, W):
    """train_3dvid.py:103-119 -> (factors, [(h,w)], epochs per level), coarse to fine."""
    if args.pyr_minimal_dim < 0:
        stages = list(map(int, args.pyr_stage.split(','))) if len(args.pyr_stage) > 0 else []
        stages = np.array([0] + stages + [args.N_iters])
        num_epoch = list(stages[1:] - stages[:-1])
        factors = [args.pyr_factor ** i for i in list(range(len(num_epoch)))[::-1]]
    else:
        num_stage = int(np.log(args.pyr_minimal_dim / min(H, W)) / np.log(args.pyr_factor)) + 1
        factors = [args.pyr_factor ** i for i in list(range(num_stage))[::-1]]
        num_epoch = [args.pyr_num_epoch] * num_stage
    return factors, [(int(H * f), int(W * f)) for f in factors], [int(n) for n in num_epoch]


def loss_configs(args, num_views, train_view=None):
    """train_3dvid.py:160-189: one loss config per training view, the reference-view config for `loss_ref_idx`."""
    other = {"loss_name": args.loss_name, "patch_size": args.swd_patch_size, "patcht_size": args.swd_patcht_size,
             "stride": args.swd_stride, "stridet": args.swd_stridet, "alpha": args.swd_alpha, "rou": args.swd_rou,
             "scaling": args.swd_scaling, "dist_fn": args.swd_dist_fn, "macro_block": args.swd_macro_block,
             "factor": args.swd_factor}
    ref = {"loss_name": args.loss_name_ref, "loss_gain": args.swd_loss_gain_ref, "patch_size": args.swd_patch_size_ref,
           "patcht_size": args.swd_patcht_size_ref, "stride": args.swd_stride_ref, "stridet": args.swd_stridet_ref,
           "alpha": args.swd_alpha_ref, "rou": args.swd_rou_ref, "scaling": args.swd_scaling_ref,
           "dist_fn": args.swd_dist_fn_ref, "macro_block": args.swd_macro_block, "factor": args.swd_factor_ref}
    cfgs = [other] * num_views
    for i in map(int, str(args.loss_ref_idx).split(',')):
        cfgs[i] = ref
    return [cfgs[i] for i in (train_view if train_view is not None else range(num_views))]


class MVVidPatchDataset:
    """train_3dvid.py:22-66 on resident tensors.  `videos`: list of [F,3,h_raw,w_raw] float tensors in [0,1] (any device);
    `poses` [V,3|4,4], `intrins` [V,3,3] for the raw resolution.  Items are (w_start, h_start, pose, intrin, crop, cfg)."""

    def __init__(self, resize_hw, videos, patch_size, patch_stride, poses, intrins, loss_configs=None, prepare="auto", prepare_budget=0.25):
        """prepare: True / False / "auto" -- build the NN search's gram-major copy of every GPU-resident clip (utils_vid.PreparedClip:
        16*H*W*F bytes per view, about 1.33x the clip, ON TOP of the clip).  "auto" does it only while the copies of all views stay
        below `prepare_budget` of the device memory that is free right now; otherwise the loss transposes the crop it is handed, per
        iteration, as before (same results, ~0.4 ms more per iteration at 720p)."""
        h_raw, w_raw = videos[0].shape[-2:]
        self.h, self.w = resize_hw
        self.v = len(videos)
        self.poses = poses.clone()
        self.intrins = intrins.clone()
        self.intrins[:, :2] *= torch.tensor([self.w / w_raw, self.h / h_raw]).reshape(1, 2, 1).type_as(intrins)
        self.patch_h_size, self.patch_w_size = patch_size
        if self.h * self.w < self.patch_h_size * self.patch_w_size:
            wh, pad_info = torch.tensor([[0, 0]]).long(), [0, 0, 0, 0]
            self.patch_h_size, self.patch_w_size = self.h, self.w
        else:
            wh, pad_info = generate_patchinfo(self.h, self.w, patch_size, patch_stride)
        self.patch_wh_start = wh[None].expand(self.v, -1, 2).reshape(-1, 2)
        self.view_index = np.arange(self.v)[:, None].repeat(wh.shape[0], axis=1).reshape(-1).tolist()
        self.loss_configs = loss_configs
        assert len(self.loss_configs) == self.v
        self.videos = []
        for vid in videos:      # cv2.resize (bilinear, no antialias) of the reference -> interpolate(bilinear)
            if vid.shape[-2:] != (self.h, self.w):
                vid = torchf.interpolate(vid, size=(self.h, self.w), mode="bilinear", align_corners=False)
            self.videos.append(torchf.pad(vid, pad_info))
        # the captured clips are constant over a pyramid level: their layout change for the NN search happens HERE, once per view,
        # and an iteration names its crop by origin (utils_vid.PreparedClip; videos resident on the GPU only)
        self.prepared = None
        if prepare and all(v.is_cuda for v in self.videos):
            if prepare == "auto":
                need = sum(16 * v.shape[-2] * v.shape[-1] * v.shape[0] for v in self.videos)
                free, _ = torch.cuda.mem_get_info(self.videos[0].device)
                prepare = need <= prepare_budget * free
                if not prepare:
                    print(f"Dataset: prepared clips would take {need / 2**30:.2f} GiB of {free / 2**30:.1f} GiB free: the loss transposes per iteration instead")
        if prepare and all(v.is_cuda for v in self.videos):
            from .utils_vid import PreparedClip
            self.prepared = [PreparedClip(v.permute(1, 0, 2, 3)) for v in self.videos]
        print(f"Dataset: generate {len(self)} patches for training, pad {pad_info} to videos")

    def __len__(self):
        return len(self.patch_wh_start)

    def __getitem__(self, item):
        w_start, h_start = (int(v) for v in self.patch_wh_start[item])
        vi = self.view_index[item]
        intrin = get_new_intrin(self.intrins[vi], h_start, w_start).float()
        crop = self.videos[vi][..., h_start:h_start + self.patch_h_size, w_start:w_start + self.patch_w_size]
        cfg = deepcopy(self.loss_configs[vi])
        if self.prepared is not None and str(cfg.get("loss_name", "")).startswith("gpnn"):
            cfg["y_prepared"] = self.prepared[vi].crop(h_start, w_start)      # (the loss classes take it; the others ignore it via **kwargs)
        return w_start, h_start, self.poses[vi], intrin, crop, cfg


def _collate1(cfg):
    """what DataLoader(batch_size=1) does to the loss config dict (MPV.py:494 un-collates it again)."""
    return {k: [v] for k, v in cfg.items()}


_WEIGHT_CACHE = {}


def weighted_total(mains, extra, weight_of):
    """loss = sum(mains) + sum_k mean(extra[k]) * weight_of(k) over the terms with a positive weight (train_3dvid.py:231-240,
    train_3d.py:221-232) -> (loss, [main terms], {k: weighted term}) with the parts as views of ONE product.
    The reference spells this as a mean, a multiply and an add per term -- on one-element device tensors each of those is a kernel launch,
    forward and backward: ~25 launches of ~3 us per iteration behind kernels that take 50-600 us.  Here: one stack, one multiply by a cached
    device vector of the weights, one sum (three launches; two on the way back).  Same gradients (1 for the mains, weight_k for term k)."""
    names = [k for k in extra if weight_of(k) > 0]
    vals = [m.reshape(()) if m.numel() == 1 else m.mean() for m in mains]
    vals += [extra[k].reshape(()) if extra[k].numel() == 1 else extra[k].mean() for k in names]
    ws = tuple([1.0] * len(mains) + [float(weight_of(k)) for k in names])
    dev = vals[0].device
    key = (ws, str(dev), vals[0].dtype)
    w = _WEIGHT_CACHE.get(key)
    if w is None:
        if len(_WEIGHT_CACHE) > 256:
            _WEIGHT_CACHE.clear()
        w = _WEIGHT_CACHE[key] = torch.tensor(ws, dtype=vals[0].dtype, device=dev)
    terms = torch.stack(vals) * w
    n = len(mains)
    return terms.sum(), [terms[i] for i in range(n)], {k: terms[n + i] for i, k in enumerate(names)}


def unit_grad(module, loss):
    """d loss / d loss = 1 from a tensor cached on the module: `loss.backward()` fills a fresh one on the device at every call -- one launch of a
    ~30-launch iteration."""
    cache = getattr(module, "__dict__", None)
    one = cache.get("_unit_grad") if cache is not None else None
    if one is None or one.device != loss.device or one.dtype != loss.dtype or one.shape != loss.shape:
        one = torch.ones_like(loss)
        if cache is not None:
            cache["_unit_grad"] = one
    return one


def run_iter(nerf, optimizer, item, args, device):
    """train_3dvid.py:214-255 without the logging."""
    _, _, pose, intrin, crop, cfg = item
    # the pose stays on the host, where the dataset holds it: the module turns it into the plane homographies (and the crop's texel
    # window for the crop-aware optimiser) there and uploads 1 KiB -- no device round trip per iteration
    b_extrin = pose2extrin_torch(pose[None].cpu())
    b_intrin = intrin[None].cpu()
    b_rgbs = crop[None].to(device)                                     # [1,F,3,h,w]
    patch_h, patch_w = b_rgbs.shape[-2:]
    if getattr(args, "add_intrin_noise", False):
        b_intrin = b_intrin.clone()
        b_intrin[:, :2, 2] += torch.rand(2).type_as(b_intrin) - 0.5     # half pixel
    nerf.train()
    if hasattr(optimizer, "acknowledge_fused_backward"):
        optimizer.acknowledge_fused_backward()      # this loop steps once per backward: the update inside the render backward is what it wants
    module = getattr(nerf, "module", nerf)
    if hasattr(module, "objective") and not getattr(args, "generic_objective", False) and getattr(module, "args", args) is not args:
        # the fused objective reads every *_loss_weight from module.args; this loop (and train()'s density ramp) reads and mutates `args`: a
        # model built from a COPY of the namespace would silently train with stale weights on one of the two paths
        raise RuntimeError("run_iter: the model was built with a different args object than the one the driver mutates; pass the same "
                           "namespace to both (or args.generic_objective = True for the reference's spelling of the objective)")
    if hasattr(module, "objective") and not getattr(args, "generic_objective", False):
        # render + looping loss + regularisers + their weighted total, the total in one launch each way (MPMeshVid.objective: the same values
        # and gradients as the spelling below); args.generic_objective keeps the reference's spelling for A/B
        loss, swd_loss, extra_losses = module.objective(patch_h, patch_w, b_extrin, b_intrin, b_rgbs, _collate1(cfg))
    else:
        _, extra = nerf(patch_h, patch_w, b_extrin, b_intrin, res=b_rgbs, losscfg=_collate1(cfg))
        swd = extra.pop("swd")
        args_var = vars(args)
        loss, (swd_loss,), extra_losses = weighted_total([swd], extra, lambda k: args_var[f"{k}_loss_weight"])
        swd_loss, extra_losses = swd_loss.detach(), {k: v.detach() for k, v in extra_losses.items()}
    optimizer.zero_grad()
    loss.backward(unit_grad(module, loss))
    optimizer.step()
    return loss.detach(), swd_loss, extra_losses


def train(nerf, args, videos, poses, intrins, loss_cfgs, H, W, device="cuda:0", on_step=None, generator=None, save_dir=None):
    """train_3dvid.py:262-306: pyramid levels x epochs x shuffled crops.  `nerf` is the MPMeshVid (or an object exposing
    `.module`); returns the number of iterations run.  No host synchronisation inside the loop unless `on_step` reads values.
    `save_dir`: every `args.i_weights` epochs write `l{level}_epoch_{epoch:04d}.tar` with the reference's keys (:295-306);
    such a file is loaded back with `MPMeshVid.init_from_mpi(ckpt['network_state_dict'])` like the reference's scripts do."""
    module = getattr(nerf, "module", nerf)
    factors, hws, epochs = pyramid_schedule(args, H, W)
    epoch_total_step = iter_total_step = 0
    dataset = None
    for pyr_i, (factor, hw, num_epoch) in enumerate(zip(factors, hws, epochs)):
        module.lod(factor)
        optimizer = module.get_optimizer(step=0)
        dataset = None      # the previous level's resized clips and prepared copies go BEFORE the next level's are built (peak = one level)
        dataset = MVVidPatchDataset(hw, videos, (args.patch_h_size, args.patch_w_size),
                                    (args.patch_h_stride, args.patch_w_stride), poses, intrins, loss_configs=loss_cfgs,
                                    prepare=getattr(args, "prepare_clips", "auto"), prepare_budget=float(getattr(args, "prepare_clips_budget", 0.25)))
        if hasattr(module, "reserve_windows"):      # the optimiser's window buffers at the level's largest crop window, before the first epoch
            module.reserve_windows((it[4].shape[-2], it[4].shape[-1], pose2extrin_torch(it[2][None].cpu()), it[3][None].cpu())
                                   for it in (dataset[i] for i in range(len(dataset))))
        for epoch_i in range(num_epoch):
            for item_i in torch.randperm(len(dataset), generator=generator).tolist():       # DataLoader(shuffle=True)
                if hasattr(module, "update_step"):
                    module.update_step(epoch_total_step)
                name_lrates = module.get_lrate(epoch_i)
                if args.lrate_adaptive:
                    name_lrates = [(n_, lr_ / len(dataset)) for n_, lr_ in name_lrates]
                for (_, new_lrate), group in zip(name_lrates, optimizer.param_groups):
                    group['lr'] = new_lrate
                out = run_iter(nerf, optimizer, dataset[item_i], args, device)
                if on_step is not None:
                    on_step(pyr_i, epoch_i, iter_total_step, *out)
                iter_total_step += 1
            if save_dir is not None and (epoch_total_step + 1) % max(int(getattr(args, "i_weights", 1)), 1) == 0:
                import os
                torch.save({'epoch_i': epoch_i, 'epoch_total_step': epoch_total_step, 'iter_total_step': iter_total_step,
                            'pyr_i': pyr_i, 'train_factor': factor, 'hw': hw, 'network_state_dict': module.state_dict()},
                           os.path.join(save_dir, f'l{pyr_i}_epoch_{epoch_i:04d}.tar'))
            epoch_total_step += 1
    return iter_total_step
