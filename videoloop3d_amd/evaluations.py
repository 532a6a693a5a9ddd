"""Drop-in for the reference's evaluations/NNMSE.py on the HIP patch-NN kernels.

`compute_nnerr(src, tar, ...)` = completeness / coherence / loop-quality NN error of scripts/script_evaluate_ours.py:201-246:
plain per-location temporal patch NN (alpha None), then mean |NN patch of tar - patch of src|, averaged per macro block
and then over macro blocks (the reference's macro-block loop changes the result here -- it is a mean of block means --
so the block structure of evaluations/NNMSE.py:31-58 is reproduced on the per-location errors)."""
import numpy as np
import torch

from . import _lib as L
from .utils_vid import find_nn_indices, fit_patch


def compute_nnerr(src, tar, patch_size=7, stride=2, patcht_size=7, stridet=2, macro_block=65):
    """evaluations/NNMSE.py:7-58.  src, tar: [1,3,f,h,w] on the MI355X -> python float."""
    t, h, w = src.shape[-3:]

    macro_block = fit_patch(macro_block, "macro_block", patch_size, stride)
    h = fit_patch(h, "patch_height", patch_size, stride)
    w = fit_patch(w, "patch_width", patch_size, stride)
    t = fit_patch(t, "frame_num", patcht_size, stridet)
    src = src[..., :t, :h, :w]
    tar = tar[..., :h, :w]
    with torch.no_grad():
        nn, desc, xv, yv = find_nn_indices(src, tar, patch_size, patcht_size, stride, stridet, None)
        h_o, w_o, n1 = nn.shape
        err = torch.empty((h_o, w_o), dtype=torch.float32, device=xv.device)
        with torch.cuda.device(xv.device):
            L.check(L.lib().vl3d_patch_l1(desc, L.ptr(xv), L.ptr(yv), L.ptr(nn), L.ptr(err), L.stream_ptr(xv.device)),
                    "vl3d_patch_l1")
        per_loc = err.double() / (n1 * 3 * patcht_size * patch_size * patch_size)     # mean |.| of one location's patches
        macro_stride = macro_block - patch_size + stride
        lpb = macro_stride // stride                                                   # patch locations per macro block and axis
        errs = []
        for hs in np.arange(0, h - macro_block + macro_stride, macro_stride):
            for ws in np.arange(0, w - macro_block + macro_stride, macro_stride):
                b0, c0 = hs // stride, ws // stride
                errs.append(per_loc[b0:b0 + lpb, c0:c0 + lpb].mean())
        return float(torch.stack(errs).mean().item())
