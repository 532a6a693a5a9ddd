"""ctypes binding of the C-ABI library (include/vl3d.h).

The HIP library is the product: there is NO CPU fallback.  Any call without the built library (or with
non-GPU tensors) raises -- see `lib()` / `check_cuda()`.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VL3D_LIB_PATH: measurement hook (A/B runs of two builds inside one GPU session, profiles/ab.sh); default = the in-tree build
LIB_PATH = os.environ.get("VL3D_LIB_PATH") or os.path.join(_HERE, "lib", "libvl3d_hip.so")

ACT = {"none": 0, "sigmoid": 1, "relu": 2, "clamp": 3, "abs": 4}
COORD = {"utils_mpi": 0, "affine": 1, "affine_planes": 2}
BORDER = {"zeros": 0, "hardcut": 1}
ACT_ORDER = {"pre": 0, "post": 1}
RHO = {"mse": 0, "abs": 1, "barron": 2}


class RenderDesc(C.Structure):
    _fields_ = [("D", C.c_int32), ("T", C.c_int32), ("Hs", C.c_int32), ("Ws", C.c_int32),
                ("H", C.c_int32), ("W", C.c_int32), ("row0", C.c_int32), ("col0", C.c_int32),
                ("coord_mode", C.c_int32), ("border_mode", C.c_int32), ("act_order", C.c_int32),
                ("rgb_act", C.c_int32), ("alpha_act", C.c_int32), ("stack_dtype", C.c_int32),
                ("pixel_center", C.c_float), ("sx", C.c_float), ("sy", C.c_float), ("ox", C.c_float),
                ("oy", C.c_float), ("variant", C.c_int32),
                ("cull_row0", C.c_int32), ("cull_col0", C.c_int32), ("cull_Hs", C.c_int32), ("cull_Ws", C.c_int32),
                ("grad_flags", C.c_int32), ("uv_noise_seed", C.c_uint32)]


class LossDesc(C.Structure):
    _fields_ = [("Tx", C.c_int32), ("Ty", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("ps", C.c_int32), ("pt", C.c_int32), ("stride", C.c_int32), ("stridet", C.c_int32),
                ("use_alpha", C.c_int32), ("alpha", C.c_float),
                ("x_sc", C.c_int64), ("x_st", C.c_int64), ("x_sr", C.c_int64),
                ("y_sc", C.c_int64), ("y_st", C.c_int64), ("y_sr", C.c_int64),
                ("variant", C.c_int32)]


_P = C.c_void_p
_I32, _I64, _F = C.c_int32, C.c_int64, C.c_float


class AdamWindow(C.Structure):
    _fields_ = [("Hs", C.c_int32), ("Ws", C.c_int32), ("y0", C.c_int32), ("x0", C.c_int32),
                ("param", _P), ("exp_avg", _P), ("exp_avg_sq", _P), ("last_step", _P), ("hist", _P),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("step", C.c_int64), ("plane_boxes", _P), ("boxes_scratch", _P),
                ("quad_keep", _P), ("quad_dyn", _P), ("QH", C.c_int32), ("QW", C.c_int32), ("class_scratch", _P),
                ("blocks", _P)]


class Stage1ObjectiveDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("scale_invariant", C.c_int32),
                ("w_img", C.c_float), ("w_loop", C.c_float), ("w_sparsity", C.c_float), ("w_density", C.c_float),
                ("w_rgb_smooth", C.c_float), ("w_a_smooth", C.c_float), ("sparsity_scale", C.c_float), ("eps", C.c_float),
                ("smooth_coef", C.c_float * 4)]


# symbol -> argtypes; every symbol include/vl3d.h declares must be listed here (tests/test_abi.py checks).
SIGNATURES = {
    "vl3d_last_error": ([], C.c_char_p),
    "vl3d_version": ([], C.c_int),
    "vl3d_render_fwd": ([C.POINTER(RenderDesc), _P, _P, _P, _P, _P, _P], C.c_int),
    "vl3d_render_fwd_frames": ([C.POINTER(RenderDesc), _P, _I32, _I32, _P, _P, _P, _P], C.c_int),
    "vl3d_render_fwd_frames_culled": ([C.POINTER(RenderDesc), _P, _I32, _I32, _P, _P, _I32, _I32, _P, _P, _P, _P], C.c_int),
    "vl3d_render_bwd_scratch_bytes": ([C.POINTER(RenderDesc)], C.c_int64),
    "vl3d_render_bwd_adam_class_bytes": ([C.POINTER(RenderDesc)], C.c_int64),
    "vl3d_render_bwd": ([C.POINTER(RenderDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P], C.c_int),
    "vl3d_render_bwd_adam": ([C.POINTER(RenderDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, C.POINTER(AdamWindow), _P], C.c_int),
    "vl3d_render_reg_state_bytes": ([C.POINTER(RenderDesc)], C.c_int64),
    "vl3d_render_fwd_reg_culled": ([C.POINTER(RenderDesc), _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _P], C.c_int),
    "vl3d_render_reg_fwd_culled": ([C.POINTER(RenderDesc), _P, _P, _P, _I32, _I32, _P, _P, _P], C.c_int),
    "vl3d_tie_static_grad": ([_I32, _I32, _I32, _I32, _P, _P, _I32, _I32, _P, _I32, _P], C.c_int),
    "vl3d_adam_step_tiles": ([_I32, _I32, _I32, _I32, _P, _P, _I32, _I32, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, _I64, _P],
                             C.c_int),
    "vl3d_adam_window_tile": ([], C.c_int32),
    "vl3d_adam_window_catchup": ([_I32] * 8 + [_P, _P, _P, _P, _P, _I32, _F, _F, _F, _P, _P, _P, _I32, _I32, _F, _I32, _P], C.c_int),
    "vl3d_adam_window_catchup_boxes": ([_I32] * 8 + [_P, _P, _P, _P, _P, _I32, _F, _F, _F, _P, _P, _P, _I32, _I32, _F, _I32, _P, _P, _P], C.c_int),
    "vl3d_adam_window_step": ([_I32] * 8 + [_P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I64, _P, _P, _I32, _I32, _I32, _P], C.c_int),
    "vl3d_adam_window_step_boxes": ([_I32] * 8 + [_P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I64, _P, _P, _I32, _I32, _I32, _P, _P, _P], C.c_int),
    "vl3d_packed_unpack_frames": ([_I32] * 4 + [_P, _P, _I32, _P, _F, _P, _P], C.c_int),
    "vl3d_render_fwd_packed": ([C.POINTER(RenderDesc), _P, _P, _P, _I32, _P, _P, _I32, _I32, _F, _P, _P, _P], C.c_int),
    "vl3d_adam_flush_older": ([_I32] * 4 + [_P, _P, _P, _P, _P, _I32, _I32, _F, _F, _F, _P, _P, _I32, _I32, _P, _P], C.c_int),
    "vl3d_adam_step_scalars": ([_F, _F, _F, _I64, C.POINTER(C.c_float), C.POINTER(C.c_float)], None),
    "vl3d_render_cull_scratch_bytes": ([C.POINTER(RenderDesc)], C.c_int64),
    "vl3d_render_fwd_culled": ([C.POINTER(RenderDesc), _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P], C.c_int),
    "vl3d_render_bwd_culled": ([C.POINTER(RenderDesc), _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P], C.c_int),
    "vl3d_render_reg_fwd": ([C.POINTER(RenderDesc), _P, _P, _P, _P, _P], C.c_int),
    "vl3d_render_fwd_reg": ([C.POINTER(RenderDesc), _P, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "vl3d_render_fwd_mask": ([C.POINTER(RenderDesc)] + [_P] * 10, C.c_int),
    "vl3d_render_bwd_mask": ([C.POINTER(RenderDesc)] + [_P] * 14 + [_I64, _P], C.c_int),
    "vl3d_label_noise_fwd": ([C.POINTER(RenderDesc), _P, _P, _P, _P, _P], C.c_int),
    "vl3d_label_noise_bwd": ([C.POINTER(RenderDesc), _P, _P, _P, _P, _P, _P], C.c_int),
    "vl3d_warp_fwd": ([_I32] * 6 + [_P, _P, _P, _P], C.c_int),
    "vl3d_warp_bwd": ([_I32] * 6 + [_P, _P, _P, _P], C.c_int),
    "vl3d_overcompose_fwd": ([_I64, _I32, _I32, _P, _P, _P, _P, _P], C.c_int),
    "vl3d_overcompose_bwd": ([_I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "vl3d_overcompose_nto0_fwd": ([_I32, _I32, _I32, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _P, _P, _P], C.c_int),
    "vl3d_overcompose_nto0_bwd": ([_I32, _I32, _I32, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _P, _P, _P, _P], C.c_int),
    "vl3d_patchnn_scratch_bytes": ([C.POINTER(LossDesc)], C.c_int64),
    "vl3d_patchnn": ([C.POINTER(LossDesc), _P, _P, _P, _P, _P], C.c_int),
    "vl3d_gram_major_bytes": ([_I32, _I32, _I32], C.c_int64),
    "vl3d_video_to_gram_major": ([_P, _I64, _I64, _I64, _I32, _I32, _I32, _P, _P], C.c_int),
    "vl3d_patchnn_prepared": ([C.POINTER(LossDesc), _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P], C.c_int),
    "vl3d_patchnn_grams": ([C.POINTER(LossDesc), _P, _I32, _I32, _I32, _P, _I32, _I32, _I32, _I32, _P, _P], C.c_int),
    "vl3d_nn_vectors": ([_I64, _I32, _I32, _I32, _P, _P, _I32, _F, _P, _P], C.c_int),
    "vl3d_patch_l1": ([C.POINTER(LossDesc), _P, _P, _P, _P, _P], C.c_int),
    "vl3d_vote_fold": ([C.POINTER(LossDesc), _P, _P, _P, _P, _I32, _P], C.c_int),
    "vl3d_vote_fold_robust": ([C.POINTER(LossDesc), _P, _P, _P, _I32, C.c_float, C.c_float, _P, _P, _P, _P, _P], C.c_int),
    "vl3d_vote_fold_robust_strided": ([C.POINTER(LossDesc), _P, _P, _P, _I32, C.c_float, C.c_float, _P, _P, _P, C.c_int64, C.c_int64,
                                       C.c_int64, _P, _P], C.c_int),
    "vl3d_scale_inplace": ([_I64, _P, _P, _P], C.c_int),
    "vl3d_loop_gain": ([_I32] * 4 + [_P, _P, _P, _P], C.c_int),
    "vl3d_loop_gain_strided": ([_I32] * 4 + [_P, _P, _I64, _I64, _I64, _P, _P], C.c_int),
    "vl3d_loop_pad_fwd": ([_I32] * 4 + [_P, _P, _P, _P], C.c_int),
    "vl3d_loop_pad_fwd_gram": ([_I32] * 4 + [_P, _P, _P, _P, _P], C.c_int),
    "vl3d_loop_pad_bwd": ([_I32] * 4 + [_P, _I64, _I64, _P, _P, _P], C.c_int),
    "vl3d_pixel_terms": ([_I64, _P, _P, _F, _P, _P, _P, _P], C.c_int),
    "vl3d_stage1_loss": ([_I32] * 4 + [_P, _I64, _I64, _I64, _P, _P, _I32, _P, _P, _P, _P], C.c_int),
    "vl3d_stage1_objective": ([C.POINTER(Stage1ObjectiveDesc)] + [_P] * 6 + [_I64] * 3 + [_P, _I64, _I64] + [_P] * 8, C.c_int),
    "vl3d_linear_head_fwd": ([_I32, C.c_uint64, _I32, _P, _P, _P, _P, _P], C.c_int),
    "vl3d_linear_head_bwd": ([_I32, _P, _P, _P, _P], C.c_int),
    "vl3d_robust_fwd": ([_I64, _P, _P, _I32, _F, _F, _P, _P], C.c_int),
    "vl3d_robust_bwd": ([_I64, _P, _P, _I32, _F, _F, _P, _F, _P, _P], C.c_int),
}

_lib = None


def sources_sha256():
    """sha256 over the files the library is compiled from (the same walk as __graft_entry__.source_hash)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) + glob.glob(os.path.join(_HERE, "csrc", "*.inc"))) \
            + [os.path.join(os.path.dirname(_HERE), "include", "vl3d.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def check_stamp(lib_path):
    """The in-tree library carries a stamp of the sources it was linked from (__graft_entry__.build): a library that does not belong to THIS
    tree -- a stale prebuilt .so beside edited kernels -- fails here, loudly, instead of answering with old code behind unchanged symbols."""
    stamp = os.path.join(os.path.dirname(lib_path), "libvl3d_hip.stamp")
    if os.path.exists(stamp) and open(stamp).read().strip() != sources_sha256():
        raise RuntimeError(f"HIP library {lib_path} was built from other sources than the ones in this tree (stamp mismatch): rebuild it with "
                           "`python -c 'import __graft_entry__ as g; g.build()'`")


def lib():
    """Load videoloop3d_amd/lib/libvl3d_hip.so (built by __graft_entry__.build()); fail loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"HIP library {LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "videoloop3d_amd has no CPU fallback.")
        if not os.environ.get("VL3D_LIB_PATH"):
            check_stamp(LIB_PATH)
        l = C.CDLL(LIB_PATH)
        # Every symbol of include/vl3d.h must resolve (tests/test_abi.py relies on it).  The one exception is explicit: an A/B run against an
        # OLDER build (VL3D_LIB_PATH + VL3D_ALLOW_MISSING_SYMBOLS=1) may lack entry points added since; they are listed on stderr once and
        # calling one fails with AttributeError.
        tolerate = bool(os.environ.get("VL3D_LIB_PATH")) and os.environ.get("VL3D_ALLOW_MISSING_SYMBOLS") == "1"
        missing = [name for name in SIGNATURES if not hasattr(l, name)]
        if missing and not tolerate:
            raise RuntimeError(f"HIP library {LIB_PATH} lacks {len(missing)} entry point(s) of include/vl3d.h ({', '.join(missing[:6])}...): it is stale or "
                               "belongs to another revision -- rebuild it (or, for an A/B against an older build, set VL3D_ALLOW_MISSING_SYMBOLS=1)")
        if missing:
            import sys
            print(f"[videoloop3d_amd] {LIB_PATH}: missing entry points tolerated (VL3D_ALLOW_MISSING_SYMBOLS=1): {', '.join(missing)}", file=sys.stderr)
        for name, (argtypes, restype) in SIGNATURES.items():
            if name in missing:
                continue
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().vl3d_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def check_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("videoloop3d_amd operators run on the MI355X only (got a CPU tensor); "
                               "there is no CPU fallback -- the CPU oracle lives in oracle/ for tests")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
