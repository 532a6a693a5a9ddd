"""Drop-in mirror of the reference's utils_vid.py looping-loss operators.

Same names / arguments / error behaviour as /root/reference/utils_vid.py.  The patch search, vote-fold and
robust loss run in the HIP library (csrc/vl3d_loss.hip); patches are never materialised (no unfoldNd), and
the macro-block loop of the reference (a pure memory cap that does not change the result, utils_vid.py:323-342)
is not needed.
"""
import contextlib
import os
import warnings

import torch

from . import _lib as L


def robust_lossfun(x, rou, scale, epsilon=1e-6):
    """utils_vid.py:10-26 (elementwise; the loss classes below use the fused HIP kernel instead)."""
    if rou == 'mse':
        return x ** 2
    elif rou == 'abs':
        return x.abs()
    rou = float(rou)
    s = (x / scale) ** 2
    if rou == 0:
        return torch.log1p(s * 0.5)
    elif rou == 2:
        return 0.5 * s
    b = abs(rou - 2) + epsilon
    d = rou + epsilon if rou >= 0 else rou - epsilon
    return (b / d) * (torch.pow(s / b + 1., 0.5 * d) - 1.) * (scale * 10)


def _rho_kind(rou):
    if rou == 'mse':
        return L.RHO["mse"], 0.0
    if rou == 'abs':
        return L.RHO["abs"], 0.0
    return L.RHO["barron"], float(rou)


def extract_3Dpatches(x, patch_size, tpatch_size, stride, tstride):
    """utils_vid.py:60-69: 3-D im2col, [b,3,T,h,w] -> [b, 3*pt*ps*ps, d_out, h_out, w_out] with channel order
    (c,kt,kh,kw).  Kept for API parity (evaluations/NNMSE.py:45 uses it); the loss path never calls it."""
    b, c = x.shape[:2]
    p = x.unfold(2, tpatch_size, tstride).unfold(3, patch_size, stride).unfold(4, patch_size, stride)
    dT, dH, dW = p.shape[2:5]
    return p.permute(0, 1, 5, 6, 7, 2, 3, 4).reshape(b, c * tpatch_size * patch_size * patch_size, dT, dH, dW)


def get_NN_indices_low_memory(X, Y, alpha, chunksz, dist_fn='mse'):
    """utils_vid.py:122-142: nearest neighbour in Y for every X over materialised patches X [B,n1,...], Y [B,n2,...]
    -> long [B,n1].  `chunksz` only bounds the reference's temporary and is ignored."""
    if dist_fn != 'mse':
        raise RuntimeError("dist_fn other than 'mse' is not settable in the reference (config_parser.py:90-93)")
    L.check_cuda(X, Y)
    B, n1 = X.shape[:2]
    n2 = Y.shape[1]
    Xf = X.reshape(B, n1, -1).to(torch.float32).contiguous()
    Yf = Y.reshape(B, n2, -1).to(torch.float32).contiguous()
    if Xf.shape[2] != Yf.shape[2] or Yf.shape[0] != B:
        raise RuntimeError("X and Y must agree in batch and feature size")
    nn = torch.empty((B, n1), dtype=torch.long, device=X.device)
    with torch.cuda.device(X.device):
        L.check(L.lib().vl3d_nn_vectors(B, n1, n2, Xf.shape[2], L.ptr(Xf), L.ptr(Yf), 0 if alpha is None else 1,
                                        0.0 if alpha is None else float(alpha), L.ptr(nn), L.stream_ptr(X.device)),
                "vl3d_nn_vectors")
    return nn


def _loss_desc(x, y, patch_size, patcht_size, stride, stridet, alpha):
    """x [3,Tx,H,W], y [3,Ty,H,W] float32 CUDA tensors with unit column stride."""
    d = L.LossDesc()
    d.Tx, d.H, d.W = x.shape[1:]
    d.Ty = y.shape[1]
    d.ps, d.pt, d.stride, d.stridet = int(patch_size), int(patcht_size), int(stride), int(stridet)
    d.use_alpha = 0 if alpha is None else 1
    d.alpha = 0.0 if alpha is None else float(alpha)
    d.x_sc, d.x_st, d.x_sr = x.stride(0), x.stride(1), x.stride(2)
    d.y_sc, d.y_st, d.y_sr = y.stride(0), y.stride(1), y.stride(2)
    # kernel-variant selector for cross-checks (bits 0-3 pick one of the four patch-NN kernels, every one of them exact; bits 12-15 a
    # fold tile shape): the timing-only bits 4-7 never leave this module (the product library refuses them anyway)
    d.variant = int(KERNEL_VARIANT) & ~0xf0
    return d


# Kernel-variant selector of the loss kernels (cross-checks and A/B runs; 0 = the dispatch's own choice).  A module attribute set through
# `kernel_variant(v)`, not an environment variable: nothing outside the caller's own code changes what the product library runs.
KERNEL_VARIANT = 0


@contextlib.contextmanager
def kernel_variant(v):
    """with kernel_variant(3): ... -- run the enclosed loss calls on one specific (exact) patch-NN kernel / fold tile shape."""
    global KERNEL_VARIANT
    old, KERNEL_VARIANT = KERNEL_VARIANT, int(v)
    try:
        yield
    finally:
        KERNEL_VARIANT = old


def _as_video(v, name):
    """[1,3,T,h,w] -> [3,T,h,w] float32 view with unit column stride (copy only if needed)."""
    if v.dim() != 5 or v.shape[0] != 1 or v.shape[1] != 3:
        raise RuntimeError(f"{name} must be [1,3,T,h,w], got {tuple(v.shape)}")
    v = v[0]
    if v.dtype != torch.float32:
        v = v.float()
    if v.stride(-1) != 1:
        v = v.contiguous()
    return v


class PreparedClip:
    """A captured clip in the NN kernel's own (gram-major) form, built ONCE: y is constant training data (the reference crops it per
    iteration, train_3dvid.py:22-66, and hands the crop to the loss, MPV.py:506), so its layout change belongs to the dataset, not to
    the iteration.  `clip` [1,3,T,H,W] (or [3,T,H,W]) float32 on the GPU; `crop(h0, w0)` names a crop origin: pass it to the loss
    classes as keyword `y_prepared` NEXT TO the crop tensor itself (the vote-fold still reads the original layout through strides).
    The caller vouches that the clip's bytes do not change afterwards."""

    def __init__(self, clip):
        v = clip[0] if clip.dim() == 5 else clip
        L.check_cuda(v)
        if v.dim() != 4 or v.shape[0] != 3:
            raise RuntimeError(f"clip must be [1,3,T,H,W] or [3,T,H,W], got {tuple(clip.shape)}")
        v = v.float()
        if v.stride(-1) != 1:
            v = v.contiguous()
        self.T, self.H, self.W = (int(n) for n in v.shape[1:])
        with torch.cuda.device(v.device):
            self.gram = torch.empty(int(L.lib().vl3d_gram_major_bytes(self.T, self.H, self.W)) // 4, dtype=torch.float32, device=v.device)
            L.check(L.lib().vl3d_video_to_gram_major(L.ptr(v), v.stride(0), v.stride(1), v.stride(2), self.T, self.H, self.W, L.ptr(self.gram),
                                                     L.stream_ptr(v.device)), "vl3d_video_to_gram_major")

    def crop(self, h0, w0):
        return ClipCrop(self, int(h0), int(w0))


class ClipCrop:
    def __init__(self, clip, h0, w0):
        self.clip, self.h0, self.w0 = clip, h0, w0


class PreparedX:
    """The generated clip x in the NN kernel's form, written by the loss prologue beside the video (MPV._LoopPrologue: the render's output is
    read once for both).  `gram`: the buffer; (frames, rows, pitch): the clip it holds -- the loss may trim x to the patch grid, the form
    keeps the untrimmed pitch.  Passed to the loss classes as keyword `x_prepared` next to x itself."""

    def __init__(self, gram, frames, rows, pitch):
        self.gram, self.frames, self.rows, self.pitch = gram, int(frames), int(rows), int(pitch)


def find_nn_indices(x, y, patch_size, patcht_size, stride, stridet, alpha, y_is_constant=False, y_prepared=None, x_prepared=None):
    """Per-location temporal NN search on videos x,y [1,3,T,h,w] (already trimmed).  Returns int32 [h_o,w_o,n1].
    y_is_constant: the caller vouches that y's bytes have not changed since the previous call with the same y tensor, so its
    pixel-major copy inside the scratch may be reused (see _patchnn_scratch).
    y_prepared: a ClipCrop -- y is the crop at that origin of a PreparedClip, whose gram-major form is read in place.
    x_prepared: a PreparedX (with y_prepared) -- x's form exists already: nothing is rewritten."""
    L.check_cuda(x, y)
    xv, yv = _as_video(x.detach(), "x"), _as_video(y.detach(), "y")
    if xv.shape[2:] != yv.shape[2:]:
        raise RuntimeError("x and y must have identical spatial size (patch grids must coincide)")
    desc = _loss_desc(xv, yv, patch_size, patcht_size, stride, stridet, alpha)
    h_o = (desc.H - desc.ps) // desc.stride + 1
    w_o = (desc.W - desc.ps) // desc.stride + 1
    n1 = (desc.Tx - desc.pt) // desc.stridet + 1
    nn = torch.empty((h_o, w_o, n1), dtype=torch.int32, device=xv.device)
    if y_prepared is not None:
        c = y_prepared.clip
        if c.T != desc.Ty or c.gram.device != xv.device:
            raise RuntimeError("y_prepared does not belong to this y (frames / device)")
        rc = 3
        if x_prepared is not None and x_prepared.gram.device == xv.device and x_prepared.frames >= desc.Tx and x_prepared.rows >= desc.H and x_prepared.pitch >= desc.W:
            with torch.cuda.device(xv.device):
                rc = L.lib().vl3d_patchnn_grams(desc, L.ptr(x_prepared.gram), x_prepared.pitch, x_prepared.rows, x_prepared.frames, L.ptr(c.gram), c.W, c.H,
                                                y_prepared.h0, y_prepared.w0, L.ptr(nn), L.stream_ptr(xv.device))
            if rc == 0:
                return nn, desc, xv, yv
            if rc != 3:
                L.check(rc, "vl3d_patchnn_grams")
        with torch.cuda.device(xv.device):
            nscratch = int(L.lib().vl3d_patchnn_scratch_bytes(desc))
            scratch, _ = _patchnn_scratch(nscratch, yv, desc, xv.device, False)
            rc = L.lib().vl3d_patchnn_prepared(desc, L.ptr(xv), L.ptr(c.gram), c.W, c.H, y_prepared.h0, y_prepared.w0, L.ptr(nn), L.ptr(scratch),
                                               L.stream_ptr(xv.device))
        if rc == 0:
            return nn, desc, xv, yv
        if rc != 3:          # VL3D_EUNSUPPORTED: clip lengths outside the matrix-core kernel -> the general path below
            L.check(rc, "vl3d_patchnn_prepared")
    with torch.cuda.device(xv.device):
        nscratch = int(L.lib().vl3d_patchnn_scratch_bytes(desc))
        scratch, y_cached = _patchnn_scratch(nscratch, yv, desc, xv.device, y_is_constant)
        if y_cached:
            desc.variant |= 0x100          # the y half of the scratch still holds this very y in pixel-major form
        L.check(L.lib().vl3d_patchnn(desc, L.ptr(xv), L.ptr(yv), L.ptr(nn), L.ptr(scratch), L.stream_ptr(xv.device)),
                "vl3d_patchnn")
        desc.variant &= ~0x100
    return nn, desc, xv, yv


# One-entry cache of the NN kernel's scratch buffer.  The buffer itself is always reused when its size fits (no allocation per
# iteration); the pixel-major copy of y inside it is reused ONLY when the caller opts in with `y_is_constant=True` (the loss
# classes forward that keyword): then y's storage, version counter, view geometry, the loss configuration and the stream that
# built the copy are the key.  Without the opt-in y is re-copied on every call -- a version counter cannot see writes that
# bypass it (`y.data.copy_()`, a HIP kernel or a dataloader refilling a pinned buffer), and in the reference's training loop
# every iteration brings another crop anyway (train_3dvid.py:214-244).  The cache holds a strong reference to the keyed
# storage, so its address cannot be recycled for other data while the entry is alive.  x (the render) is always re-copied.
_SCRATCH_CACHE = {"key": None, "buf": None, "storage": None}


def _patchnn_scratch(nbytes, yv, desc, device, y_is_constant=False):
    if nbytes <= 0:
        return None, False
    stream = torch.cuda.current_stream(device).cuda_stream
    buf = _SCRATCH_CACHE["buf"]
    fits = buf is not None and buf.device == torch.device(device) and buf.numel() * 4 >= nbytes and _SCRATCH_CACHE.get("stream") == stream
    if not y_is_constant:
        if not fits:
            buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _SCRATCH_CACHE.update(key=None, buf=buf, storage=None, stream=stream)
        return buf, False
    st = yv.untyped_storage()
    key = (st.data_ptr(), yv._version, yv.storage_offset(), tuple(yv.shape), tuple(yv.stride()), str(device), nbytes, stream,
           desc.Tx, desc.ps, desc.pt, desc.stride, desc.stridet, desc.variant & 0xff)
    if fits and _SCRATCH_CACHE["key"] == key:
        return buf, True
    buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    _SCRATCH_CACHE.update(key=key, buf=buf, storage=st, stream=stream)
    return buf, False


def _nn_and_fold(x, y, patch_size, patcht_size, stride, stridet, alpha, normalize, y_is_constant=False):
    nn, desc, xv, yv = find_nn_indices(x, y, patch_size, patcht_size, stride, stridet, alpha, y_is_constant)
    s = torch.empty((1, 3, desc.Tx, desc.H, desc.W), dtype=torch.float32, device=xv.device)
    w = torch.empty((1, 1, desc.Tx, desc.H, desc.W), dtype=torch.float32, device=xv.device)
    with torch.cuda.device(xv.device):
        L.check(L.lib().vl3d_vote_fold(desc, L.ptr(yv), L.ptr(nn), L.ptr(s), L.ptr(w), 1 if normalize else 0,
                                       L.stream_ptr(xv.device)), "vl3d_vote_fold")
    return s, w, nn


def _floored(size, patch, step):
    """extent covered by the floored patch grid UnfoldNd / FoldNd use (utils_vid.py:60-69, 218-227): no warning, no trimming of
    the outputs -- the direct path keeps x's full shape and leaves the uncovered voxels empty."""
    return (size - patch) // step * step + patch if size >= patch else size


def _nn_and_fold_any_size(x, y, patch_size, patcht_size, stride, stridet, alpha, normalize, y_is_constant=False):
    """_nn_and_fold for x of ANY size, as the reference's direct path handles it: UnfoldNd floors the patch grid and FoldNd writes
    into the full x.shape, so voxels beyond the last whole patch receive no vote -- sum 0 and weight 1e-10 (after the clamp,
    utils_vid.py:228), i.e. y2x = 0 there.  The kernels run on the covered extent; the rest is filled here."""
    t, h, w = x.shape[-3:]
    tf, hf, wf = _floored(t, patcht_size, stridet), _floored(h, patch_size, stride), _floored(w, patch_size, stride)
    if (tf, hf, wf) == (t, h, w):
        return _nn_and_fold(x, y, patch_size, patcht_size, stride, stridet, alpha, normalize, y_is_constant)
    if y.shape[-2:] != x.shape[-2:]:
        raise RuntimeError("x and y must have identical spatial size (patch grids must coincide)")
    sc, wc, nn = _nn_and_fold(x[..., :tf, :hf, :wf], y[..., :hf, :wf], patch_size, patcht_size, stride, stridet, alpha, normalize,
                              y_is_constant)
    s = x.new_zeros((1, 3, t, h, w), dtype=torch.float32)
    wgt = x.new_full((1, 1, t, h, w), 1e-10, dtype=torch.float32)
    s[..., :tf, :hf, :wf] = sc
    wgt[..., :tf, :hf, :wf] = wc
    return s, wgt, nn


def FindNNpatchAndMerge(x, y, patch_size=7, patcht_size=7, stride=1, stridet=1, alpha=1e10, dist_fn='mse', **kwargs):
    """utils_vid.py:206-229 -> (y2x_sum [1,3,T,h,w], weight [1,1,T,h,w] clamped at 1e-10); x of any size (floored patch grid)."""
    if dist_fn != 'mse':
        raise RuntimeError("dist_fn other than 'mse' is not settable in the reference (config_parser.py:90-93)")
    alpha = None if alpha > 100 else alpha
    s, w, _ = _nn_and_fold_any_size(x, y, patch_size, patcht_size, stride, stridet, alpha, normalize=False,
                                    y_is_constant=bool(kwargs.get("y_is_constant", False)))
    return s, w


class _RobustMean(torch.autograd.Function):
    """loss = robust_lossfun(x - y2x, rou, scaling).mean()  (utils_vid.py:348), gradient to x only."""

    @staticmethod
    def forward(ctx, x, y2x, rou, scaling):
        L.check_cuda(x, y2x)
        xc = x.to(torch.float32).contiguous()
        yc = y2x.contiguous()
        kind, r = _rho_kind(rou)
        acc = torch.empty((), dtype=torch.float64, device=xc.device)
        n = xc.numel()
        with torch.cuda.device(xc.device):
            L.check(L.lib().vl3d_robust_fwd(n, L.ptr(xc), L.ptr(yc), kind, r, float(scaling), L.ptr(acc),
                                            L.stream_ptr(xc.device)), "vl3d_robust_fwd")
        ctx.save_for_backward(xc, yc)
        ctx.cfg = (kind, r, float(scaling), n)
        return (acc / n).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        xc, yc = ctx.saved_tensors
        kind, r, scaling, n = ctx.cfg
        g = g.to(torch.float32).contiguous()
        gx = torch.empty_like(xc)
        with torch.cuda.device(xc.device):
            L.check(L.lib().vl3d_robust_bwd(n, L.ptr(xc), L.ptr(yc), kind, r, scaling, L.ptr(g), 1.0 / n, L.ptr(gx),
                                            L.stream_ptr(xc.device)), "vl3d_robust_bwd")
        return gx, None, None, None


class _FoldRobustMean(torch.autograd.Function):
    """loss = robust_lossfun(x - fold(y, nn)/weight, rou, scaling).mean() in ONE pass over the video (vl3d_vote_fold_robust):
    the NN search (no_grad in the reference, utils_vid.py:279,322), the vote-fold, the loss sum and d loss / d x.  Returns
    (loss, y2x, weight); gradient flows to x only, through the loss.
    trim = (t, h, w): x arrives UNTRIMMED and the op works on x[..., :t, :h, :w] like the reference's slicing (utils_vid.py:307-320);
    its gradient is written straight into a zero-filled buffer of x's full shape, so the three slice backwards (a zero fill and a
    copy of the whole video each) never run."""

    @staticmethod
    def forward(ctx, x, y, patch_size, patcht_size, stride, stridet, alpha, rou, scaling, y_is_constant=False, trim=None, holder=None,
                y_prepared=None, x_prepared=None):
        ctx.set_materialize_grads(False)      # outputs the loss does not use come back as None, not as zero-filled tensors (a fill each, and reads in the backward kernels)
        xs = x if trim is None else x[..., :trim[0], :trim[1], :trim[2]]
        nn, desc, xv, yv = find_nn_indices(xs, y, patch_size, patcht_size, stride, stridet, alpha, y_is_constant, y_prepared, x_prepared)
        dev = xv.device
        gx = torch.empty(x.shape, dtype=torch.float32, device=dev)
        if tuple(xs.shape) != tuple(x.shape):
            # the gradient of the untrimmed x is zero outside the trimmed box: only that border is filled (a fill of the whole
            # video was 0.12 ms of a 720p iteration)
            gx[..., trim[0]:, :, :] = 0
            gx[..., :, trim[1]:, :] = 0
            gx[..., :, :, trim[2]:] = 0
        acc = torch.empty((), dtype=torch.float64, device=dev)
        kind, r = _rho_kind(rou)
        with torch.cuda.device(dev):
            # y2x / weight are NOT written here (767 MB of stores per 720p iteration that no caller of the reference reads:
            # utils_vid.py:345-346 caches them for `same_input`, which nothing sets); `holder` recomputes them on first access
            L.check(L.lib().vl3d_vote_fold_robust_strided(desc, L.ptr(yv), L.ptr(nn), L.ptr(xv), kind, r, float(scaling), None,
                                                          None, L.ptr(gx), gx.stride(1), gx.stride(2), gx.stride(3), L.ptr(acc),
                                                          L.stream_ptr(dev)), "vl3d_vote_fold_robust_strided")
        if holder is not None:
            holder._lazy = (desc, yv, nn)
        ctx.save_for_backward(gx)
        ctx.x_dtype = x.dtype
        # (the mean in double, rounded once to fp32 -- one launch: the division writes its fp32 result itself)
        return torch.div(acc, float(3 * desc.Tx * desc.H * desc.W), out=torch.empty((), dtype=torch.float32, device=dev))

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 14
        (gx,) = ctx.saved_tensors
        # scaled IN PLACE (the buffer belongs to this node; an out-of-place product is one more pass over the video, 0.2 ms at 720p),
        # so the node can be differentiated once
        if getattr(ctx, "consumed", False):
            raise RuntimeError("the fused looping loss keeps its gradient buffer in place: backward through it a second time needs a new forward")
        ctx.consumed = True
        gs = g.to(torch.float32).contiguous()
        if gx.numel() % 4 == 0 and gx.is_contiguous():
            with torch.cuda.device(gx.device):      # (a no-op on the device when the upstream gradient is exactly 1)
                L.check(L.lib().vl3d_scale_inplace(gx.numel(), L.ptr(gx), L.ptr(gs), L.stream_ptr(gx.device)), "vl3d_scale_inplace")
        else:
            gx.mul_(gs)
        return gx.to(ctx.x_dtype), None, None, None, None, None, None, None, None, None, None, None, None, None


def fit_patch(size, name, patch, step):
    """Largest size' <= size with (size' - patch) % step == 0, warning with the reference's text when it trims
    (utils_vid.py:307-313; evaluations/NNMSE.py uses the same rule)."""
    trimmed = (size - patch) // step * step + patch
    if trimmed != size:
        warnings.warn(f'{name} doesnot satisfy ({name} - patch_size) % stride == 0. changing {name} from {size} to {trimmed}')
    return trimmed


def _gpnn_loss(holder, x, y, patch_size, patcht_size, stride, stridet, alpha, rou, scaling, y_is_constant=False, trim=None, y_prepared=None,
               x_prepared=None):
    """NN search + vote-fold + robust mean, fused (one pass over the video for fold, loss and gradient); falls back to the
    separate kernels when a fold tile does not fit LDS, or when x does not fit the patch grid (direct path only: the LowMem class
    trims first).  Caches y2x / weight on `holder` like the reference (utils_vid.py:345-346)."""
    t, h, w = x.shape[-3:] if trim is None else trim
    fits_grid = (_floored(t, patcht_size, stridet), _floored(h, patch_size, stride), _floored(w, patch_size, stride)) == (t, h, w)
    try:
        if not fits_grid:
            raise RuntimeError("x does not fit the patch grid: unfused path")
        loss = _FoldRobustMean.apply(x, y, patch_size, patcht_size, stride, stridet, alpha, rou, scaling, y_is_constant, trim, holder, y_prepared,
                                     x_prepared)
        holder._y2x = holder._weight = None
        return loss
    except RuntimeError as e:
        if "does not fit LDS" not in str(e) and "does not fit the patch grid" not in str(e):
            raise
        if trim is not None:
            x = x[..., :t, :h, :w]
        with torch.no_grad():
            y2x, weight, _ = _nn_and_fold_any_size(x, y, patch_size, patcht_size, stride, stridet, alpha, normalize=True,
                                                   y_is_constant=y_is_constant)
        loss = _RobustMean.apply(x, y2x, rou, scaling)
    holder.last_weight, holder.last_y2x = weight, y2x
    return loss


class _LazyVotes:
    """last_y2x / last_weight of the loss classes (utils_vid.py:345-346).  The fused kernel keeps the vote average in registers; the
    two tensors are produced from the saved NN indices on first access (vl3d_vote_fold, the same kernel without the loss)."""
    _lazy = None
    _y2x = None
    _weight = None

    def _materialise(self):
        if self._y2x is None and self._lazy is not None:
            desc, yv, nn = self._lazy
            dev = yv.device
            s = torch.empty((1, 3, desc.Tx, desc.H, desc.W), dtype=torch.float32, device=dev)
            w = torch.empty((1, 1, desc.Tx, desc.H, desc.W), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                L.check(L.lib().vl3d_vote_fold(desc, L.ptr(yv), L.ptr(nn), L.ptr(s), L.ptr(w), 1, L.stream_ptr(dev)), "vl3d_vote_fold")
            self._y2x, self._weight = s, w

    @property
    def last_y2x(self):
        self._materialise()
        return self._y2x

    @last_y2x.setter
    def last_y2x(self, v):
        self._y2x, self._lazy = v, None

    @property
    def last_weight(self):
        self._materialise()
        return self._weight

    @last_weight.setter
    def last_weight(self, v):
        self._weight = v


class Patch3DGPNNDirectLoss(_LazyVotes):
    """utils_vid.py:265-286.  Like the reference's, this path takes x of ANY size: UnfoldNd floors the patch grid and FoldNd
    leaves the voxels beyond the last whole patch without a vote (y2x = 0, weight 1e-10), and the loss mean runs over all of x
    -- `loss_name='gpnn'` is the parser default and an even crop size with the default stride 2 is legal there.
    Extra keyword (ignored by the reference through **kwargs): y_is_constant=True lets the NN kernel reuse its pixel-major copy
    of y from the previous call with the same tensor."""

    def __init__(self):
        self._lazy = self._y2x = self._weight = None

    def __call__(self, x, y, mask=None, same_input=False, rou=0, scaling=0.2, **kwargs):
        if same_input:
            return _RobustMean.apply(x, self.last_y2x, rou, scaling)
        cfg = dict(patch_size=7, patcht_size=7, stride=1, stridet=1, alpha=1e10)
        cfg.update({k: v for k, v in kwargs.items() if k in cfg})
        alpha = None if cfg["alpha"] > 100 else cfg["alpha"]
        if kwargs.get("dist_fn", "mse") != "mse":
            raise RuntimeError("dist_fn other than 'mse' is not settable in the reference")
        return _gpnn_loss(self, x, y, cfg["patch_size"], cfg["patcht_size"], cfg["stride"], cfg["stridet"], alpha, rou, scaling,
                          bool(kwargs.get("y_is_constant", False)), None, kwargs.get("y_prepared"), kwargs.get("x_prepared"))


class Patch3DGPNNLowMemLoss(_LazyVotes):
    """utils_vid.py:289-349.  `macro_block` is accepted and fitted (with the reference's warning) but no
    macro-block loop is needed: the HIP path has no unfold memory to cap and the result is identical.
    One backward per forward: the fused loss keeps its gradient buffer in place (scaled by the upstream gradient when the backward runs), so
    differentiating the same loss tensor a second time (retain_graph, gradient penalties) raises -- call the loss again instead."""

    def __init__(self):
        self._lazy = self._y2x = self._weight = None

    def __call__(self, x, y, mask=None, same_input=False, macro_block=64, patch_size=7, stride=2, patcht_size=7,
                 stridet=2, rou=0, scaling=0.2, **kwargs):
        if same_input:
            weight, y2x = self.last_weight, self.last_y2x
        else:
            t, h, w = x.shape[-3:]

            macro_block = fit_patch(macro_block, "macro_block", patch_size, stride)
            h = fit_patch(h, "patch_height", patch_size, stride)
            w = fit_patch(w, "patch_width", patch_size, stride)
            t = fit_patch(t, "frame_num", patcht_size, stridet)
            y = y[..., :h, :w]
            alpha = kwargs.get("alpha", 1e10)
            alpha = None if alpha > 100 else alpha
            if kwargs.get("dist_fn", "mse") != "mse":
                raise RuntimeError("dist_fn other than 'mse' is not settable in the reference")
            # x is trimmed INSIDE the fused op (same values as slicing here, utils_vid.py:318): its gradient comes back in x's full shape
            trim = None if (t, h, w) == tuple(x.shape[-3:]) else (t, h, w)
            return _gpnn_loss(self, x, y, patch_size, patcht_size, stride, stridet, alpha, rou, scaling,
                              bool(kwargs.get("y_is_constant", False)), trim, kwargs.get("y_prepared"), kwargs.get("x_prepared"))
        return _RobustMean.apply(x, y2x, rou, scaling)


def Patch3DMSE(x, y, **kwargs):
    """utils_vid.py:437-440."""
    frm = min(x.shape[2], y.shape[2])
    return ((x[:, :, :frm] - y[:, :, :frm]) ** 2).mean()


def Patch3DAvg(x, y, **kwargs):
    """utils_vid.py:443-445."""
    return ((x.mean(dim=2) - y.mean(dim=2)) ** 2).mean()
