/* vl3d.h -- C ABI of the MI355X-native MPI/MPV render-and-composite + looping-loss hot path.
 *
 * One shared library (videoloop3d_amd/lib/libvl3d_hip.so, built by hipcc for gfx950) exports
 * exactly these symbols.  Plain pointers and sizes only: no torch types.  The reference
 * (limacv/VideoLoop3D) is pure Python and has no FFI; each entry point cites the reference
 * operator (file:line in /root/reference) it replaces, and INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Contract:
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates or frees
 *     and never synchronises; all work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - all tensors are dense row-major fp32 unless stated; outputs are overwritten (the library
 *     zero-fills accumulation targets itself on the same stream).
 *   - return value: VL3D_OK or an error code; never aborts.  vl3d_last_error() gives a message.
 */
#ifndef VL3D_H
#define VL3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *vl3d_stream_t; /* hipStream_t */

enum { VL3D_OK = 0, VL3D_EINVAL = 1, VL3D_ELAUNCH = 2, VL3D_EUNSUPPORTED = 3 };

/* activation table: MPI.py:21-31 (ACTIVATES); shipped configs use sigmoid for rgb and alpha
 * (configs/mpv_base.txt:29-30). */
enum { VL3D_ACT_NONE = 0, VL3D_ACT_SIGMOID = 1, VL3D_ACT_RELU = 2, VL3D_ACT_CLAMP = 3, VL3D_ACT_ABS = 4 };
/* texel-coordinate convention.  UTILS_MPI: g = p/[Ws/2,Hs/2]-1 then grid_sample(align_corners=True),
 * i.e. texel = p*(size-1)/size (utils_mpi.py:173-175).  AFFINE: texel = p*s + o (MPV.py atlas-cell UV). */
enum { VL3D_COORD_UTILS_MPI = 0, VL3D_COORD_AFFINE = 1, VL3D_COORD_AFFINE_PLANES = 2 };
/* AFFINE_PLANES (with BORDER_HARDCUT, ACT_POST, sigmoid/sigmoid): every plane has its OWN affine texel transform and quad extent --
 * the reference's atlas-cell layout, where plane p of an atlas of grid_h x grid_w cells samples at xm*pitch - (p % grid_w)/grid_w
 * with pitch = (Aw-1)/(grid_w*(mpi_w-1)) and is covered while 0 <= xm <= mpi_w-1 (MPV.py:75-81, 394-439).  `homos` is then
 * (D,16): per plane the 3x3 matrix target pixel -> TEXEL coordinate (the plane's transform already folded in), then the coverage
 * box x0, x1, y0, y1 in texel coordinates (inclusive), 3 floats of padding; desc->sx = sy = 1, ox = oy = 0. */
/* ZEROS: grid_sample padding_mode='zeros' (utils_mpi.py:174).  HARDCUT: quad extent, uncovered -> 0 after
 * activation (MPV.py:389,441-447). */
enum { VL3D_BORDER_ZEROS = 0, VL3D_BORDER_HARDCUT = 1 };
/* PRE: activate texels, then sample (sigmoid -> warp_homography chain).  POST: sample, then activate (MPV.py:425-435). */
enum { VL3D_ACT_PRE = 0, VL3D_ACT_POST = 1 };
enum { VL3D_F32 = 0, VL3D_F16 = 1 };

const char *vl3d_last_error(void);
int vl3d_version(void);

/* ------------------------------------------------------------------------------------------------
 * Fused render: per-plane homography warp + bilinear sample + activation + front-to-back over
 * composite across D planes.  Replaces the chain MPV.py:351-454 (planar geometry) ==
 * utils_mpi.py:159-176 (warp_homography) + utils_mpi.py:92-107 (overcompose).
 *
 *   stack  : (D, T, Hs, Ws, 4) pre-activation rgba, plane 0 = nearest (MPV.py:51)
 *   homos  : (D, 3, 3) fp32, target pixel -> plane pixel (utils_mpi.py:240-273), shared by all T frames
 *   rgb    : (T, H, W, 3)        alpha : (T, H, W)  = sum of blend weights (MPV.py:454)
 *   row0/col0 : offset of this output window inside the full frame (row-band sharding);
 *               pixel (y,x) of the window is frame pixel (row0+y, col0+x).
 */
typedef struct vl3d_render_desc {
    int32_t D, T, Hs, Ws;
    int32_t H, W;
    int32_t row0, col0;
    int32_t coord_mode, border_mode, act_order, rgb_act, alpha_act;
    int32_t stack_dtype;   /* VL3D_F32 | VL3D_F16 (grad_stack has the same dtype) */
    float pixel_center;    /* 0 (utils_mpi) or 0.5 (pytorch3d pixel centres) */
    float sx, sy, ox, oy;  /* VL3D_COORD_AFFINE only */
    int32_t variant;       /* kernel selector for bitwise cross-checks between EXACT kernels; 0 = default.  Bits 0-3: backward (see
                            * vl3d_render_bwd); bits 8-11: forward -- 6 = one frame per thread (default for the shipped activations and
                            * T >= 2: two frames per thread, same bits); bits 12-15: 1 = the two-pass forward with regularisers.  Bits 4-7
                            * (timing-only ablations, wrong results) exist only in a -DVL3D_VARIANTS measurement build: the product
                            * library returns VL3D_EINVAL for them. */
    /* tile culling on a WINDOW of the stack (vl3d_render_*_culled only; all 0 = the stack is the whole plane): the stack passed in is
     * the texel window [cull_row0, cull_row0+Hs) x [cull_col0, cull_col0+Ws) of a cull_Hs x cull_Ws plane, and the quad grid of
     * quad_keep lies over that whole plane. */
    int32_t cull_row0, cull_col0, cull_Hs, cull_Ws;
    /* vl3d_render_bwd_culled only.  Bit 0 (VL3D_GRAD_CULLED_UNWRITTEN): the caller never reads the gradient of texels no kept quad can
     * read (vl3d_adam_window_step / vl3d_adam_step_tiles skip them: they are no parameters), so the texels a workgroup owns on a plane it
     * skips are not written at all instead of being zero-filled -- on a 16 %-kept model that fill was a quarter of the backward's time.
     * Those slots of grad_stack are then UNDEFINED.  0 = every texel of grad_stack is written (a gradient any consumer may read).
     * A permission, not a promise: instantiations without the layer regularisers still write the zeros. */
    int32_t grad_flags;
    /* add_uv_noise of the reference (MPV.py:420-423, MPI.py:519-522; config_parser.py:48; off in every shipped configuration): while training,
     * every sample's UV is jittered by hpix * (2 rand - 1), hpix = 1 / (atlas size - 1) in normalised coordinates = HALF A TEXEL, one draw
     * per (pixel, layer), shared by all frames.  0 = off.  Non-zero: the seed of this call's jitter field -- (jx, jy) uniform in
     * [-0.5, 0.5) texels from a counter hash of (seed, plane, frame pixel): a pure function, so forward, backward and any tiling see the
     * same field (the oracle restates it: oracle/mpi_oracle.uv_jitter_field).  Coverage (hard cut, quad culling) is decided at the UNJITTERED
     * position, as the rasteriser decides it in the reference; the taps move.  VL3D_COORD_AFFINE(_PLANES) with VL3D_BORDER_HARDCUT only; the backward takes the atomics
     * kernel (a jittered tap can leave the owner-computes kernels' 1-pixel halo); vl3d_render_bwd_adam / _fwd_packed refuse it. */
    uint32_t uv_noise_seed;
} vl3d_render_desc;
enum { VL3D_GRAD_CULLED_UNWRITTEN = 1 };

/* alpha_sums (optional, may be NULL): (T,H,W,2) per pixel (sum_k a_k, sum_k a_k^2) over the planes -- the two sums the
 * sparsity regulariser (MPV.py:511-515 / MPI.py:599-603: |a|_1 / |a|_2 per pixel) is made of. */
int vl3d_render_fwd(const vl3d_render_desc *desc, const void *stack, const float *homos,
                    float *rgb, float *alpha, float *alpha_sums, vl3d_stream_t stream);
/* ... of frames frame0 .. frame0 + desc->T - 1 of a LONGER clip, read in place: `stack` is the base of a (D, T_alloc, Hs, Ws, 4) allocation,
 * desc->T the number of consecutive frames to render (an evaluation render of single frames or runs of frames -- scripts/script_render_video.py:
 * 129-139 renders one frame per camera of its path -- without gathering `stack[:, ts]` first: 571 MB per 720p frame at D = 32 on 1.1x planes). */
int vl3d_render_fwd_frames(const vl3d_render_desc *desc, const void *stack, int32_t frame0, int32_t T_alloc, const float *homos, float *rgb,
                           float *alpha, vl3d_stream_t stream);
/* ... of a tile-culled (sparsified) dense model: vl3d_render_fwd_culled's quad map and scratch, the same run of frames. */
int vl3d_render_fwd_frames_culled(const vl3d_render_desc *desc, const void *stack, int32_t frame0, int32_t T_alloc, const float *homos,
                                  const uint8_t *quad_keep, int32_t QH, int32_t QW, void *cull_scratch, float *rgb, float *alpha,
                                  vl3d_stream_t stream);

/* Backward of the above w.r.t. the stack (geometry is not differentiated: MPV.py:354).
 * rgb/alpha are the saved forward outputs; grad_alpha may be NULL (treated as 0).
 * grad_alpha_sums (optional): (T,H,W,2) gradient w.r.t. alpha_sums of the forward.
 * grad_stack (D,T,Hs,Ws,4) is overwritten; it has the dtype of the stack (fp32, or fp16 for stack_dtype == VL3D_F16 -- the
 * cfg5 shards only fit with an 8-byte gradient texel).
 * scratch: caller-owned device buffer of vl3d_render_bwd_scratch_bytes(desc) bytes (plan written and read on
 * `stream`, no host sync); with scratch == NULL the universal global-atomics kernel is used.
 * desc->variant: 0 auto (LDS-staged owner-computes kernel when its on-device feasibility plan allows, atomics
 * kernel otherwise), 1 force atomics, 3 owner-computes kernel, one frame per thread in 64 x 16-pixel regions (2: the 8-row regions of
 * round 1, no longer built, selects 3), 4 = 3 with the 3x3 gather everywhere (reference for the 2x2 gather of no-minification tiles, which
 * must equal it bit for bit), 5 = one frame per thread in 32 x 16-pixel regions (the shipped planar convention with fp32 stacks; = 3
 * elsewhere).  All of them produce the same gradient bits. */
int64_t vl3d_render_bwd_scratch_bytes(const vl3d_render_desc *desc);
int vl3d_render_bwd(const vl3d_render_desc *desc, const void *stack, const float *homos,
                    const float *rgb, const float *alpha, const float *grad_rgb, const float *grad_alpha,
                    const float *grad_reg, const void *reg_state, const float *grad_alpha_sums, float *grad_stack, void *scratch,
                    int64_t scratch_bytes, vl3d_stream_t stream);

/* Tile culling (MPI.py:288-442 "Tile Culling Algorithm": stage 2 renders only the quads that survived).  quad_keep is a device
 * byte map [D][QH][QW] over the cells of each plane's vertex grid ((Ws-1)/QW x (Hs-1)/QH texels per quad), 1 = the quad exists.
 * A sample that falls into a culled quad of a plane is NOT COVERED by that plane (the reference's mesh has no face there):
 * it contributes nothing and passes no gradient, whatever the texels hold; its layer value for the smoothness regularisers
 * is 0 like any uncovered pixel (MPV.py:441).  On top of that rule, work is skipped where a whole workgroup sees no kept
 * quad of a plane (forward: a 64-bit plane mask per workgroup walked with scalar bit scans; backward: a flag in the tile's
 * window record -- no sweep, no barrier, zeros stored to the texels it owns), which changes no result.  D <= 128.
 * cull_scratch: vl3d_render_cull_scratch_bytes(desc) bytes (forward plan, rebuilt every call, no host sync); the backward
 * keeps its plan in its own scratch. */
/* Tile-exact layout (round 6).  `sparsify_faces` cuts every kept quad out of the atlas as a tile WITH ITS OWN border row / column (MPI.py:380-418:
 * neighbouring tiles hold two copies of their common border samples), every face's UVs span exactly its tile (gen_quad_uvs, MPI.py:403-418;
 * sampled by MPV.py:394-427 / MPI.py:497-536), and stage 2 trains the two copies APART -- a static tile's copy is one texture, its dynamic
 * neighbour's copy moves per frame.  A stack on which neighbouring quads SHARE their border texels (the layout above) cannot hold such a
 * checkpoint.  Passing a NEGATIVE quad grid (-QH, -QW) to any entry point that takes one selects the tile-exact layout instead:
 *   - a plane is QH x QW tiles of th x tw texels, th = Hs / QH, tw = Ws / QW (whole tiles of at least 2 x 2 texels; with desc->cull_*: of the
 *     cull_Hs x cull_Ws plane the stack is a window of -- the window itself need not be tile aligned); texel (y, x) belongs to quad
 *     (y / th, x / tw) and to no other: its class (culled / static / dynamic) is that quad's;
 *   - desc->sx, sy, ox, oy map a plane pixel to the LATTICE coordinate L (quads sharing borders: a quad spans tw - 1 of them, quad q =
 *     floor(L / (tw - 1))); the sample's texel coordinate is L + q -- position L - q (tw - 1) in [0, tw - 1] inside tile q, whose first texel
 *     is q tw -- so its bilinear taps never leave its own tile with a non-zero weight: grid_sample(align_corners=True) on the reference's
 *     atlas between the tile's corner texel centres.  With a window, ox / oy are reduced by the window's texel origin as usual.
 *   - everything else (hard cut at the plane's extent, culled quads uncovered, hit-slot regularisers, owner-computes backward, the fused
 *     optimiser step, packed pools, add_uv_noise jittering the TILE coordinate) is unchanged: the layout is a piecewise translation of the
 *     texel coordinate.  VL3D_COORD_AFFINE + VL3D_BORDER_HARDCUT only (the reference has tiles on the planar MPV / MPI path only).
 * Pinned by golden G19 (tests/golden/make_golden_r06.py: the reference's own forward on a checkpoint whose border copies differ). */
int64_t vl3d_render_cull_scratch_bytes(const vl3d_render_desc *desc);
int vl3d_render_fwd_culled(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep,
                           int32_t QH, int32_t QW, void *cull_scratch, float *rgb, float *alpha, float *alpha_sums,
                           vl3d_stream_t stream);
int vl3d_render_bwd_culled(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep,
                           int32_t QH, int32_t QW, const float *rgb, const float *alpha, const float *grad_rgb,
                           const float *grad_alpha, const float *grad_reg, const void *reg_state, const float *grad_alpha_sums,
                           float *grad_stack, void *scratch, int64_t scratch_bytes, vl3d_stream_t stream);

/* Static tiles of a tile-culled VIDEO stack (MPV.py:235-288: one static atlas shared by all frames).  In place on the stack
 * gradient (D,T,Hs,Ws,4): texels that only static quads can read get the sum over the T frames in every frame (the T copies
 * then stay one texture under any optimiser), texels no kept quad can read get 0, texels a dynamic quad can read are left
 * alone.  quad_keep / quad_dyn: device byte maps [D][QH][QW].  mode bit 0: the gradient comes from the culled render
 * (vl3d_render_bwd_culled), which leaves exactly 0 in culled texels -- they are not rewritten.  mode bit 1: the consumer is
 * vl3d_adam_step_tiles with the same quad_dyn, which reads a static texel's gradient from frame 0 only -- the sum is written
 * to frame 0 alone (the other frames keep their per-frame values and must not be used). */
int vl3d_tie_static_grad(int32_t D, int32_t T, int32_t Hs, int32_t Ws, const uint8_t *quad_keep, const uint8_t *quad_dyn,
                         int32_t QH, int32_t QW, float *grad, int32_t mode, vl3d_stream_t stream);

/* torch.optim.Adam step (no amsgrad, no weight decay; MPV.py:199-214) on a stack parameter (D,T,Hs,Ws,4), in place on param /
 * exp_avg / exp_avg_sq, restricted to the texels a kept quad can read (quad_keep NULL: all texels).  Culled texels have zero
 * gradient and zero moments for ever, so skipping them is exact; `step` is the 1-based step count of this update.
 * (QH, QW) negative: the tile-exact layout ("Tile-exact layout" above; also for vl3d_tie_static_grad and the vl3d_adam_window_* family).
 * quad_dyn (optional, with quad_keep): texels only static quads can read are ONE parameter with T identical copies -- the
 * update is computed once from frame 0 (gradient = the frame sum vl3d_tie_static_grad leaves there; moments live in frame 0)
 * and the new value is written to all T copies. */
int vl3d_adam_step_tiles(int32_t D, int32_t T, int32_t Hs, int32_t Ws, const uint8_t *quad_keep, const uint8_t *quad_dyn, int32_t QH, int32_t QW,
                         float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float lr, float beta1, float beta2,
                         float eps, int64_t step, vl3d_stream_t stream);

/* Crop-aware Adam for the DENSE stack (csrc/vl3d_optim.hip; the optimiser of train_3dvid.py:263-290 / MPV.py:199-214).  A training
 * iteration renders one crop, so only the texels of the crop's parallax window (y0, x0, wh, ww; aligned to
 * vl3d_adam_window_tile()-texel tiles, or ending at the plane border) receive a gradient; the zero-gradient Adam steps of all
 * other texels (m <- b1 m, v <- b2 v, p <- p - lr_t m^/(sqrt(v^) + eps)) are deferred and replayed exactly, per tile, when a tile is
 * next needed.  last_step: device int32 [D][ceil(Hs/16)][ceil(Ws/16)], the step each tile is current for (0 initially).
 * hist: device float2 [steps+1], entry t = (lr_t / (1 - beta1^t), sqrt(1 - beta2^t)) as vl3d_adam_step_scalars computes them.
 * Round 6: the step functions (vl3d_adam_window_step / _boxes, vl3d_render_bwd_adam) WRITE row `step` themselves from their own lr / betas /
 * step arguments, behind the update, in the launch that marks the tiles (the table is `const` for every reader; a caller that also fills the row
 * writes the same two floats).  The caller sizes the table (step < its length) and keeps rows 0 .. step intact.
 *   vl3d_adam_window_catchup: the window's parameters as of step `upto` (steps last_step+1 .. upto replayed with g = 0).  With
 *     compact != NULL they are written to the compact (D,T,wh,ww,4) buffer the render then reads and the stack is left untouched;
 *     with compact == NULL (a flush) they are written back and the tiles marked current.
 *   vl3d_adam_window_step: replays the missed steps of the window's tiles up to step-1, then the Adam update of step `step` from the
 *     compact gradient (D,T,wh,ww,4), writes (p, m, v) and marks the tiles.  Everything outside the window stays deferred.
 * A catch-up with the full plane as the window makes the whole stack current (checkpoints, lod, evaluation renders).
 * Tile-culled models (quad_keep / quad_dyn != NULL, device byte maps [D][QH][QW]): culled texels are no parameters (the compact copy
 * shows them as (0, 0, 0, culled_alpha)); a texel only static quads can read is ONE parameter stored in frame 0 -- the compact copy
 * shows it in every frame, the step sums its compact gradient over the frames and writes frame 0 only; mirror_static bit 0 makes a
 * catch-up refresh the other frames' slots of static texels (so that the dense stack reads consistently: flush); static_tied != 0
 * tells the step that frame 0 of the gradient already holds a static texel's frame sum (vl3d_tie_static_grad ran on it).
 * mirror_static bit 1 ("lean" compact copy): the caller guarantees that `compact` holds FINITE values everywhere already (a persistent
 * buffer that was zero-filled once and has only ever been written by this call), so the slots the render cannot read with a non-zero
 * weight -- culled texels, texels outside their plane's box -- are not written: for a tile-culled model they were most of the copy. */
int32_t vl3d_adam_window_tile(void);
int vl3d_adam_window_catchup(int32_t D, int32_t T, int32_t Hs, int32_t Ws, int32_t y0, int32_t x0, int32_t wh, int32_t ww,
                             float *param, float *exp_avg, float *exp_avg_sq, int32_t *last_step, const float *hist, int32_t upto,
                             float beta1, float beta2, float eps, float *compact, const uint8_t *quad_keep, const uint8_t *quad_dyn,
                             int32_t QH, int32_t QW, float culled_alpha, int32_t mirror_static, vl3d_stream_t stream);
int vl3d_adam_window_step(int32_t D, int32_t T, int32_t Hs, int32_t Ws, int32_t y0, int32_t x0, int32_t wh, int32_t ww,
                          float *param, const float *grad_compact, float *exp_avg, float *exp_avg_sq, int32_t *last_step,
                          const float *hist, float lr, float beta1, float beta2, float eps, int64_t step, const uint8_t *quad_keep,
                          const uint8_t *quad_dyn,
                          int32_t QH, int32_t QW, int32_t static_tied, vl3d_stream_t stream);

/* The optimiser step INSIDE the render backward (train_3dvid.py:242-244: loss.backward(); optimizer.step() -- for the dense stage-2 model
 * they are one pass over the window's texels).  vl3d_render_bwd of the crop-aware iteration writes the window's compact gradient
 * (one stream) only for vl3d_adam_window_step to read it back beside (p, m, v) and write (p, m, v): 2 + 7 streams of the window.  Here the
 * owner-computes backward applies the step of `adam->step` where it would have stored a texel's complete gradient sum: 1 (taps) + 2 (m, v)
 * + 3 (p, m, v) streams, no gradient round trip.  `stack` is the compact copy of the texel window at (adam->y0, adam->x0) of the
 * (D,T,adam->Hs,adam->Ws,4) parameter -- desc->Hs x desc->Ws texels, written by vl3d_adam_window_catchup(_boxes) with upto = step - 1, so it
 * holds the parameters current for step - 1 and only the two moments are replayed (multiplications).  Window / plane_boxes / last_step /
 * hist: as vl3d_adam_window_step_boxes (hist[step] is written by this call); on return the window's tiles are marked `step` and (p, m, v) are
 * bit for bit what vl3d_render_bwd followed by vl3d_adam_window_step_boxes would have left (tests/test_gpu_optim.py).  Every texel of
 * window and box is stepped exactly once: by the tile that owns it, or -- texels no tile's gather reaches: zero gradient -- by the
 * backward's pre-pass.  grad_stack (the compact gradient, desc's dims) is still required: when the device-side plan finds the view
 * infeasible for the owner-computes kernels the atomics kernel fills it and the step kernel runs behind it, decided on the device (no
 * host synchronisation either way); it is left unwritten otherwise.
 * Tile-culled models (adam->quad_keep != NULL; the render culls with the same map, desc->cull_* = the window (y0, x0) of the (Hs, Ws) plane):
 * a DYNAMIC texel is stepped in the owner's store, a STATIC texel (one parameter for all frames) has its gradient stored to grad_stack as
 * vl3d_render_bwd_culled does and the step kernel behind the backward sums it over the frames -- static texels only, unless the plan was
 * infeasible --, culled texels are nobody's (grad_stack holds defined values in static texels only).  adam->blocks: packed storage.
 * fp32 stacks, the planar convention with the shipped activations ((affine, hardcut, post), sigmoid / sigmoid); dense models: T >= 2,
 * desc->variant 0 (the frame pairs; tile-culled models always take the one-frame tile kernel: 32-wide regions, variant 3 = 64-wide); anything else: VL3D_EUNSUPPORTED,
 * nothing launched. */
typedef struct vl3d_adam_window {
    int32_t Hs, Ws;              /* the full planes: param / exp_avg / exp_avg_sq are (D,T,Hs,Ws,4) */
    int32_t y0, x0;              /* the window [y0, y0 + desc->Hs) x [x0, x0 + desc->Ws), aligned to vl3d_adam_window_tile() */
    float *param, *exp_avg, *exp_avg_sq;
    int32_t *last_step;
    const float *hist;
    float lr, beta1, beta2, eps;
    int64_t step;
    const int32_t *plane_boxes;  /* HOST [D][4] or NULL */
    void *boxes_scratch;         /* device, 16 * D bytes; required with plane_boxes */
    /* tile-culled model (NULL quad_keep: dense): the quad maps of vl3d_adam_window_step (device byte maps [D][QH][QW]) and a device scratch
     * of vl3d_render_bwd_adam_class_bytes(desc) bytes for the texel records of the window (8 bytes per plane texel: class, the step its
     * bookkeeping tile is current for, its slot in the parameter tensors / pools -- written by the pre-pass, read by the owner's store) */
    const uint8_t *quad_keep, *quad_dyn;
    int32_t QH, QW;
    void *class_scratch;
    const int32_t *blocks;       /* PACKED storage (as vl3d_adam_window_step_boxes; needs the quad maps): param / exp_avg / exp_avg_sq are the pools */
} vl3d_adam_window;
int64_t vl3d_render_bwd_adam_class_bytes(const vl3d_render_desc *desc);
int vl3d_render_bwd_adam(const vl3d_render_desc *desc, const void *stack, const float *homos, const float *rgb, const float *alpha,
                         const float *grad_rgb, const float *grad_alpha, const float *grad_reg, const void *reg_state,
                         const float *grad_alpha_sums, float *grad_stack, void *scratch, int64_t scratch_bytes,
                         const vl3d_adam_window *adam, vl3d_stream_t stream);

/* PACKED storage of a tile-culled model (`blocks` != NULL on the three entry points below; quad maps required): the reference keeps a
 * static atlas (one frame), a dynamic atlas (T frames) and no storage for culled quads (MPI.py:364-436, MPV.py:235-288).  Here
 * param / exp_avg / exp_avg_sq are pools of 8 x 8-texel blocks (vl3d_adam_window_tile()): blocks [D][ceil(Hs/8)][ceil(Ws/8)] int32 = -1 for
 * a block no kept quad can read (not stored), else slot << 1 | dynamic, a slot being 64 texels (1 KiB); a static block owns one slot, a
 * dynamic block T consecutive ones (frame-major).  The compact window copy / gradient the render kernels work on stay dense, so the
 * kernels of the hot path are unchanged and the parameters after every step have the bits of the dense (D,T,Hs,Ws,4) model; the pool of
 * a model with 16 % of its quads kept is about a seventh of the dense stack (videoloop3d_amd/packed.py).
 *
 * The same two with per-plane boxes: plane_boxes [D][4] = (y0, y1, x0, x1) in plane texels, tile aligned, inside the window, in HOST
 * memory -- 16 bytes per plane that travel in the kernel arguments, no copy, no synchronisation (NULL, or more than 128 planes: the whole
 * window for every plane).  The window of a crop is the union of the planes' footprints; texels of the window outside
 * their own plane's box cannot be sampled in this iteration, their gradient is exactly zero and their update stays deferred like that of
 * every texel outside the window: the catch-up leaves their slots of the compact copy unwritten (never read), the step skips them. */
int vl3d_adam_window_catchup_boxes(int32_t D, int32_t T, int32_t Hs, int32_t Ws, int32_t y0, int32_t x0, int32_t wh, int32_t ww,
                                   float *param, float *exp_avg, float *exp_avg_sq, int32_t *last_step, const float *hist, int32_t upto,
                                   float beta1, float beta2, float eps, float *compact, const uint8_t *quad_keep, const uint8_t *quad_dyn,
                                   int32_t QH, int32_t QW, float culled_alpha, int32_t mirror_static, const int32_t *plane_boxes,
                                   const int32_t *blocks, vl3d_stream_t stream);
int vl3d_adam_window_step_boxes(int32_t D, int32_t T, int32_t Hs, int32_t Ws, int32_t y0, int32_t x0, int32_t wh, int32_t ww,
                                float *param, const float *grad_compact, float *exp_avg, float *exp_avg_sq, int32_t *last_step,
                                const float *hist, float lr, float beta1, float beta2, float eps, int64_t step, const uint8_t *quad_keep,
                                const uint8_t *quad_dyn, int32_t QH, int32_t QW, int32_t static_tied, const int32_t *plane_boxes,
                                const int32_t *blocks, vl3d_stream_t stream);
/* Bound on the deferral: every bookkeeping tile that has missed at least min_depth steps is replayed up to `upto`, written back and marked
 * (the others are left alone).  Run after each step it keeps what a returning crop window has to replay below min_depth steps per texel
 * -- the reference shuffles 32-72 crops x views per epoch (train_3dvid.py:263-290), so a window comes back after that many steps. */
/* Chosen frames of a PACKED model as a dense stack for an evaluation render (MPV.py:439 `atlas_dyn[ts]`): out (D,n,Hs,Ws,4); frames =
 * device int32 [n], each in [0,T); blocks without storage read (0, 0, 0, culled_alpha), static blocks their one copy in every frame. */
int vl3d_packed_unpack_frames(int32_t D, int32_t T, int32_t Hs, int32_t Ws, const int32_t *blocks, const float *pool, int32_t n,
                              const int32_t *frames, float culled_alpha, float *out, vl3d_stream_t stream);
/* The forward of MPV.py:351-454 reading the PACKED texture directly -- the reference renders a sparsified model from its static / dynamic
 * tile lists (MPV.py:389-449), never from a dense texture.  desc: D, T (frames of the model), Hs x Ws (texels of a plane), the H x W view,
 * pixel_center / sx / sy / ox / oy, activations; convention (VL3D_COORD_AFFINE, VL3D_BORDER_HARDCUT, VL3D_ACT_POST), fp32.  blocks / pool:
 * as above; frames = device int32 [n]; quad_keep [D][QH][QW] = the map the block table was built from (a sample in a culled quad is not
 * covered).  rgb (n,H,W,3), alpha (n,H,W): the bits of vl3d_render_fwd_culled on the unpacked frames.  Forward only (evaluation renders):
 * training renders from the compact window copy of the crop-aware optimiser (vl3d_adam_window_catchup). */
int vl3d_render_fwd_packed(const vl3d_render_desc *desc, const int32_t *blocks, const float *pool, const int32_t *frames, int32_t n,
                           const float *homos, const uint8_t *quad_keep, int32_t QH, int32_t QW, float culled_alpha, float *rgb,
                           float *alpha, vl3d_stream_t stream);
int vl3d_adam_flush_older(int32_t D, int32_t T, int32_t Hs, int32_t Ws, float *param, float *exp_avg, float *exp_avg_sq, int32_t *last_step,
                          const float *hist, int32_t upto, int32_t min_depth, float beta1, float beta2, float eps, const uint8_t *quad_keep,
                          const uint8_t *quad_dyn, int32_t QH, int32_t QW, const int32_t *blocks, vl3d_stream_t stream);
void vl3d_adam_step_scalars(float lr, float beta1, float beta2, int64_t step, float *lr_bc1, float *bc2s);

/* Layer-space smoothness regularisers (MPV.py:517-531 rgb_smooth / a_smooth; MPI.py:608-622) WITHOUT the materialised [T,h,w,K,4]
 * layer tensor: sums[0..3] (device doubles, overwritten) = sum over frames, layer SLOTS and neighbouring pixel pairs of |L[p]-L[q]|
 * for (x-pairs, rgb), (y-pairs, rgb), (x-pairs, alpha), (y-pairs, alpha).  L is the reference's `mpi` tensor: slot k of a pixel holds
 * the warped+activated rgba of the k-th nearest plane that COVERS it (masked_scatter over the rasteriser's z-sorted pix_to_face,
 * MPV.py:386-392, 441-449; unused slots 0) -- equal to plane k only where both pixels of a pair are covered by the same planes.
 * reg_state: caller-owned device buffer of vl3d_render_reg_state_bytes(desc) bytes, written by these forwards (per-pixel coverage
 * masks and pair flags; per (plane, frame, pixel) the signs of the four differences its layer value takes part in) and read by
 * vl3d_render_bwd, which takes the gradient w.r.t. the four sums as grad_reg = device float[4] (NULL: no regulariser term;
 * non-NULL requires the reg_state of the matching forward).  At most 128 planes. */
int64_t vl3d_render_reg_state_bytes(const vl3d_render_desc *desc);
int vl3d_render_reg_fwd(const vl3d_render_desc *desc, const void *stack, const float *homos, double *sums, void *reg_state,
                        vl3d_stream_t stream);
/* vl3d_render_fwd and vl3d_render_reg_fwd in ONE pass over the stack (dense stacks): what MPMeshVid.forward needs per training step
 * when rgb_smooth / a_smooth are on (configs/mpv_base.txt:33-34).  rgb / alpha / alpha_sums bit-identical to vl3d_render_fwd. */
int vl3d_render_fwd_reg(const vl3d_render_desc *desc, const void *stack, const float *homos, float *rgb, float *alpha,
                        float *alpha_sums, double *sums, void *reg_state, vl3d_stream_t stream);
int vl3d_render_reg_fwd_culled(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep,
                               int32_t QH, int32_t QW, double *sums, void *reg_state, vl3d_stream_t stream);
/* ... and the whole forward of a tile-culled model WITH the regularisers in one pass: rgb / alpha / alpha_sums as vl3d_render_fwd_culled
 * writes them (the same bits) and sums / reg_state as vl3d_render_reg_fwd_culled -- the slot kernel visits every covered plane of every
 * pixel nearest first, which is the order of the over-composite, so the render falls out of the samples it takes anyway. */
int vl3d_render_fwd_reg_culled(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep, int32_t QH,
                               int32_t QW, float *rgb, float *alpha, float *alpha_sums, double *sums, void *reg_state, vl3d_stream_t stream);

/* Stage 1's learned loop mask (MPI.py:115-117 `atlas_mask`, 568-583) as a FIFTH composited channel of the same pass:
 *     label(pixel) = sum_k w_k sigmoid(sample(mask_k)),   w_k = a_k T_k  the colour composite's blend weights,
 * sampled with the colour taps' positions and weights (grid_sample over the same uvs, MPI.py:569-572), and the weights DETACHED
 * (MPI.py:577-579 "detach mpi so that geometry is not related to mpi"): grad_label reaches the mask texture only.
 *   mask (D,T,Hs,Ws) logits, one float per texel of the stack;  label (T,H,W);  grad_mask (D,T,Hs,Ws) overwritten.
 * vl3d_render_fwd_mask: sums / reg_state both NULL = vl3d_render_fwd plus the label, both given = vl3d_render_fwd_reg plus the label
 * (rgb, alpha, alpha_sums, sums, reg_state exactly as those write them).  vl3d_render_bwd_mask = vl3d_render_bwd plus grad_label ->
 * grad_mask in the same sweep and gather (one frame per thread; grad_stack as vl3d_render_bwd computes it).  Built for the planar
 * convention stage 1 ships -- (VL3D_COORD_AFFINE, VL3D_BORDER_HARDCUT, VL3D_ACT_POST), sigmoid / sigmoid, fp32 stack, no quad map
 * (the reference drops the mask when it sparsifies, MPI.py:440-441); any other descriptor returns VL3D_EUNSUPPORTED and the caller
 * renders the label in a pass of its own (a stack whose channel 0 is the mask logit and channel 3 the alpha logit). */
int vl3d_render_fwd_mask(const vl3d_render_desc *desc, const void *stack, const float *mask, const float *homos, float *rgb,
                         float *alpha, float *label, float *alpha_sums, double *sums, void *reg_state, vl3d_stream_t stream);
int vl3d_render_bwd_mask(const vl3d_render_desc *desc, const void *stack, const float *mask, const float *homos, const float *rgb,
                         const float *alpha, const float *grad_rgb, const float *grad_alpha, const float *grad_label,
                         const float *grad_reg, const void *reg_state, const float *grad_alpha_sums, float *grad_stack,
                         float *grad_mask, void *scratch, int64_t scratch_bytes, vl3d_stream_t stream);

/* ... and the label alone when add_uv_noise is on (MPI.py:519-522 with :568-583): the reference jitters the COLOUR samples' UVs but samples the loop
 * mask at the UNJITTERED UVs and composites sigmoid(mask) with the detached alphas of the jittered samples -- two sampling positions per layer,
 * which the fused label channel above does not have (it refuses desc->uv_noise_seed != 0).  label (T,H,W) = sum_k w_k sigmoid(sample(mask_k, uv)),
 * w_k = a_k T_k with a_k = alpha_act(sample(alpha_k, uv + jitter_k)) x coverage(uv): the alphas the colour pass of the SAME desc->uv_noise_seed
 * composites with (same counter-hash field).  Gradient to the mask texture only (grad_mask (D,T,Hs,Ws), overwritten).  Planar convention
 * (affine, hardcut, post), fp32 stack, any alpha activation.  A completeness path (no shipped configuration sets add_uv_noise): plain kernels. */
int vl3d_label_noise_fwd(const vl3d_render_desc *desc, const float *stack, const float *mask, const float *homos, float *label, vl3d_stream_t stream);
int vl3d_label_noise_bwd(const vl3d_render_desc *desc, const float *stack, const float *mask, const float *homos, const float *grad_label,
                         float *grad_mask, vl3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Unfused operators (drop-ins for the reference's L3 functions).
 *
 * vl3d_warp_*: utils_mpi.py:159-176 warp_homography(h, w, homos[N,3,3], images[N,C,Hs,Ws]) -> [N,C,h,w]
 * with N = B*D (bilinear, zeros padding, align_corners=True under the utils_mpi normalisation). */
int vl3d_warp_fwd(int32_t N, int32_t C, int32_t Hs, int32_t Ws, int32_t h, int32_t w,
                  const float *homos, const float *images, float *out, vl3d_stream_t stream);
int vl3d_warp_bwd(int32_t N, int32_t C, int32_t Hs, int32_t Ws, int32_t h, int32_t w,
                  const float *homos, const float *grad_out, float *grad_images, vl3d_stream_t stream);

/* vl3d_overcompose_*: utils_mpi.py:92-107 overcompose(alpha[P,D], content[P,D,C]) -> rgb[P,C], blendweight[P,D];
 * front = index 0; P = B*H*W pixels. */
int vl3d_overcompose_fwd(int64_t P, int32_t D, int32_t C, const float *alpha, const float *content,
                         float *rgb, float *blendweight, vl3d_stream_t stream);
/* grad_bw may be NULL. */
int vl3d_overcompose_bwd(int64_t P, int32_t D, int32_t C, const float *alpha, const float *content,
                         const float *grad_rgb, const float *grad_bw,
                         float *grad_alpha, float *grad_content, vl3d_stream_t stream);

/* vl3d_overcompose_nto0_*: utils_mpi.py:110-132 overcomposeNto0; front = LAST plane index.
 * alpha[b,d,hw] at alpha + b*a_sb + d*a_sd + hw ; content[b,d,c,hw] at content + b*c_sb + d*c_sd + c*c_sc + hw
 * (element strides, so slices of mpi[B,D,4,H,W] need no copy).  rgb [B,C,HW]; trans [B,D,HW] (the
 * transmittance the reference returns as `blendweight` when ret_mask=True; may be NULL). */
int vl3d_overcompose_nto0_fwd(int32_t B, int32_t D, int32_t C, int64_t HW,
                              const float *alpha, int64_t a_sb, int64_t a_sd,
                              const float *content, int64_t c_sb, int64_t c_sd, int64_t c_sc,
                              float *rgb, float *trans, vl3d_stream_t stream);
/* grad_alpha [B,D,HW], grad_content [B,D,C,HW] dense. */
int vl3d_overcompose_nto0_bwd(int32_t B, int32_t D, int32_t C, int64_t HW,
                              const float *alpha, int64_t a_sb, int64_t a_sd,
                              const float *content, int64_t c_sb, int64_t c_sd, int64_t c_sc,
                              const float *grad_rgb, float *grad_alpha, float *grad_content,
                              vl3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Looping loss (utils_vid.py).  x is [3,Tx,H,W], y is [3,Ty,H,W] (batch 1, already trimmed to the
 * patch grid: utils_vid.py:307-320); element strides are given per (channel, frame, row), columns
 * are unit-stride.  Patch grid: h_o=(H-ps)/stride+1, w_o=(W-ps)/stride+1, n1=(Tx-pt)/stridet+1,
 * n2=(Ty-pt)/stridet+1. */
typedef struct vl3d_loss_desc {
    int32_t Tx, Ty, H, W;
    int32_t ps, pt, stride, stridet;
    int32_t use_alpha;     /* 0: plain NN (alpha > 100 in the reference, utils_vid.py:208) */
    float alpha;           /* utils_vid.py:133-134 normaliser offset */
    int64_t x_sc, x_st, x_sr;
    int64_t y_sc, y_st, y_sr;
    int32_t variant;       /* kernel variant selector for A/B measurements and cross-checks; 0 = default.  Bits 0-3, vl3d_patchnn: 1 strided
                            * staging (no scratch), 2 one location per workgroup, 4 the vector-ALU kernel, 6 the split-f16 matrix-core kernel
                            * (0 picks 6 wherever the clip lengths allow it -- x <= 128, y <= 192 frames -- and 4 otherwise); vl3d_vote_fold: 1 = the
                            * unstaged kernel.  Bits 4-7: refused (ablation switches of a -DVL3D_VARIANTS measurement build).  Bit 8: see vl3d_patchnn.
                            * Bit 11 (0x800), kernel 6: the workgroup-wide epilogue also without alpha.  Bits 12-15, vl3d_vote_fold*: tile shape index + 1. */
} vl3d_loss_desc;

/* bytes of device scratch vl3d_patchnn needs for this problem (0 if none). */
int64_t vl3d_patchnn_scratch_bytes(const vl3d_loss_desc *desc);

/* Per spatial location b=(by,bx): dist[i,j] = |Px_i - Py_j|^2 / (3*pt*ps*ps) (utils_vid.py:72-86),
 * optional column-min normalisation (utils_vid.py:109-119,133-134), row argmin, first minimum wins
 * (utils_vid.py:139-141).  nn is int32 [h_o, w_o, n1].  Replaces extract_3Dpatches x2 +
 * get_NN_indices_low_memory (utils_vid.py:209-216).
 * The kernel works on pixel-major copies of x and y kept in `scratch` (their layout belongs to the kernel variant: a scratch filled
 * under one variant is not valid for another).  desc->variant bit 8 (0x100): the y copy in `scratch`
 * is still valid from the previous call with the same y, configuration and scratch buffer -- skip re-building it (the
 * captured video y is constant over the iterations of a training loop; x, the render, is not). */
int vl3d_patchnn(const vl3d_loss_desc *desc, const float *x, const float *y, int32_t *nn,
                 void *scratch, vl3d_stream_t stream);

/* The captured clip y is constant training data (train_3dvid.py:22-66 crops it per iteration, MPV.py:506 hands the crop to the loss):
 * vl3d_video_to_gram_major rewrites a whole clip [3,T,H,W] (element strides per channel / frame / row, unit column stride) ONCE into
 * the NN kernel's own form (vl3d_gram_major_bytes(T, H, W) bytes), and vl3d_patchnn_prepared searches against the crop
 * (y_row0, y_col0) + (desc->H, desc->W) of that buffer (rows of y_pitch pixels, y_rows of them; desc->Ty = T) without touching y again
 * -- the y half of the per-iteration layout change (0.4 of the 2.7 ms of a 720p search) leaves the iteration.  Matrix-core kernel only
 * (x <= 128, y <= 192 frames): VL3D_EUNSUPPORTED otherwise, and the caller falls back to vl3d_patchnn.  `scratch` as for vl3d_patchnn. */
int64_t vl3d_gram_major_bytes(int32_t T, int32_t H, int32_t W);
int vl3d_video_to_gram_major(const float *y, int64_t sc, int64_t st, int64_t sr, int32_t T, int32_t H, int32_t W, float *out,
                             vl3d_stream_t stream);
int vl3d_patchnn_prepared(const vl3d_loss_desc *desc, const float *x, const float *y_gram, int32_t y_pitch, int32_t y_rows,
                          int32_t y_row0, int32_t y_col0, int32_t *nn, void *scratch, vl3d_stream_t stream);
/* ... with x in that form too (vl3d_loop_pad_fwd_gram writes it beside the video: the render's output is read once for both layouts of the
 * loss, MPV.py:484-507 -> utils_vid.py:209-216): x_gram holds an x_rows x x_pitch clip of x_frames frames, desc (Tx, H, W) names its
 * leading frames / rows / columns (the loss's trim to the patch grid, utils_vid.py:307-320).  No scratch. */
int vl3d_patchnn_grams(const vl3d_loss_desc *desc, const float *x_gram, int32_t x_pitch, int32_t x_rows, int32_t x_frames, const float *y_gram,
                       int32_t y_pitch, int32_t y_rows, int32_t y_row0, int32_t y_col0, int32_t *nn, vl3d_stream_t stream);

/* get_NN_indices_low_memory(X[B,n1,...], Y[B,n2,...], alpha, chunksz, 'mse') on MATERIALISED patches
 * (utils_vid.py:122-142; the caller evaluations/NNMSE.py:45-56 builds them with extract_3Dpatches).
 * X [B,n1,d], Y [B,n2,d] dense fp32; nn int64 [B,n1] like the reference's torch.long. */
int vl3d_nn_vectors(int64_t B, int32_t n1, int32_t n2, int32_t d, const float *X, const float *Y,
                    int32_t use_alpha, float alpha, int64_t *nn, vl3d_stream_t stream);

/* Gather the NN patches of y and vote-fold them onto x's grid (utils_vid.py:217-229):
 * sum [3,Tx,H,W] dense, weight [Tx,H,W] dense = vote count clamped at 1e-10.
 * normalize != 0 writes sum/weight (utils_vid.py:344) instead of the raw sum. */
int vl3d_vote_fold(const vl3d_loss_desc *desc, const float *y, const int32_t *nn,
                   float *sum, float *weight, int32_t normalize, vl3d_stream_t stream);

/* NN-error metric support (evaluations/NNMSE.py:45-56): err[h_o*w_o] (overwritten) = per patch location the sum over its
 * n1 patches and their 3*pt*ps*ps elements of |y patch at nn - x patch|; nn from vl3d_patchnn (use_alpha = 0). */
int vl3d_patch_l1(const vl3d_loss_desc *desc, const float *x, const float *y, const int32_t *nn, float *err,
                  vl3d_stream_t stream);

/* Fused vote-fold + robust loss (utils_vid.py:217-229 + :344 + :348 in one pass): y2x = fold(y, nn) / weight is written
 * (the loss classes cache it as last_y2x), and while it is in a register the robust loss of (x - y2x) is summed into
 * loss_sum (device double, overwritten) and grad_x (3,Tx,H,W contiguous) receives d mean(rho)/dx = rho'(x - y2x) / (3 Tx H W).
 * x uses the strides of desc.  Replaces vl3d_vote_fold(normalize=1) + vl3d_robust_fwd + vl3d_robust_bwd. */
int vl3d_vote_fold_robust(const vl3d_loss_desc *desc, const float *y, const int32_t *nn, const float *x, int32_t rho_kind,
                          float rou, float scale, float *y2x, float *weight, float *grad_x, double *loss_sum,
                          vl3d_stream_t stream);

/* The same with grad_x addressed through strides (elements): channel, frame, row; unit column stride.  The reference trims x to the
 * patch grid by slicing before the loss (utils_vid.py:307-320), so the gradient of the untrimmed x is zero outside the trimmed box:
 * the caller hands a zero-filled buffer of x's FULL shape and its strides, and the slice's backward (a zero fill and a copy per sliced
 * axis) never runs. */
/* y2x and weight may both be NULL: the vote average then stays in registers (767 MB of stores less per 720p iteration; the loss
 * classes materialise last_y2x / last_weight on first access through vl3d_vote_fold). */
int vl3d_vote_fold_robust_strided(const vl3d_loss_desc *desc, const float *y, const int32_t *nn, const float *x, int32_t rho_kind,
                                  float rou, float scale, float *y2x, float *weight, float *grad_x, int64_t gx_sc, int64_t gx_st,
                                  int64_t gx_sr, double *loss_sum, vl3d_stream_t stream);

/* x[0..n) *= *scale with `scale` a DEVICE scalar (no host sync); no memory traffic at all when it is exactly 1 -- the backward of the fused
 * looping loss: its gradient buffer exists since the forward and the upstream gradient of a loss differentiated directly is 1.  n % 4 == 0. */
int vl3d_scale_inplace(int64_t n, float *x, const float *scale, vl3d_stream_t stream);

/* The loss prologue of MPMeshVid.forward (MPV.py:484-507), from the render's NHWC output to the loss's video in three launches:
 *   rgb_pad = cat(rgb, rgb[:pad])                                       loop padding (MPV.py:490-492)
 *   scale   = (exp(mean(log((mean_f res + 0.01) / (mean_t rgb + 0.01)))) + 3) / 4     scale-invariant gain (MPV.py:499-504; rgb detached)
 *   x       = rgb_pad * scale,  handed to the loss as [1,3,T+pad,h,w]
 * vl3d_loop_gain: rgb (T,h,w,3), res (F,3,h,w) -> *log_sum (device double, zeroed by the call) = the sum of the 3 h w log ratios.
 * vl3d_loop_pad_fwd: x (3,T+pad,h,w) = gain * rgb with the first `pad` frames repeated at the end; log_sum == NULL: gain 1.
 * vl3d_loop_pad_bwd: grad_rgb (T,h,w,3) = gain * (grad_x[:, t] + grad_x[:, T + t] for t < pad); grad_x (3,T+pad,h,w) with channel /
 * frame strides gx_sc / gx_st in floats (unit column stride, rows contiguous). */
int vl3d_loop_gain(int32_t T, int32_t F, int32_t h, int32_t w, const float *rgb, const float *res, double *log_sum, vl3d_stream_t stream);
/* ... with res (F,3,h,w) read through strides in floats (r_sf frame, r_sc channel, r_sr row; unit column stride): a crop of the captured clip as it lies. */
int vl3d_loop_gain_strided(int32_t T, int32_t F, int32_t h, int32_t w, const float *rgb, const float *res, int64_t r_sf, int64_t r_sc, int64_t r_sr,
                           double *log_sum, vl3d_stream_t stream);
int vl3d_loop_pad_fwd(int32_t T, int32_t pad, int32_t h, int32_t w, const float *rgb, const double *log_sum, float *x, vl3d_stream_t stream);
/* ... writing, beside x, its form for the NN search: x_gram = vl3d_gram_major_bytes(T + pad, h, w) bytes (for vl3d_patchnn_grams). */
int vl3d_loop_pad_fwd_gram(int32_t T, int32_t pad, int32_t h, int32_t w, const float *rgb, const double *log_sum, float *x, float *x_gram,
                           vl3d_stream_t stream);
int vl3d_loop_pad_bwd(int32_t T, int32_t pad, int32_t h, int32_t w, const float *grad_x, int64_t gx_sc, int64_t gx_st, const double *log_sum,
                      float *grad_rgb, vl3d_stream_t stream);

/* Per-pixel terms of MPMesh.forward / MPMeshVid.forward after the fused render, one pass each way instead of ~30 scalar launches:
 *   sums[0] = sum_p  n1_p / max(sqrt(max(n2_p, 1e-30)), eps)      sparsity (MPI.py:599-603, MPV.py:511-515), (n1, n2) = alpha_sums (n,2) = (sum_k a_k, sum_k a_k^2)
 *   sums[1] = sum_p | alpha_p - 1 |                               density  (MPI.py:647-650, MPV.py:533-536)
 * either input may be NULL (its sum stays 0).  grad_alpha_sums (n,2) / grad_alpha (n), optional: d sums[0] / d alpha_sums, d sums[1] / d alpha
 * (unit upstream gradient, not divided by n: the caller scales).  sums: device double[2], overwritten. */
int vl3d_pixel_terms(int64_t n, const float *alpha, const float *alpha_sums, float eps, double *sums, float *grad_alpha_sums, float *grad_alpha,
                     vl3d_stream_t stream);

/* Stage 1's image + loop-mask loss (train_3d.py:200-220) on the module's output rgbl (B,C,h,w), C = 3 | 4, read through strides in floats
 * (sb batch, sc channel, sp pixel: the NHWC render output viewed as NCHW); target (B,3,h,w), target_mask (B,h,w) contiguous:
 *   s       = scale_invariant ? (exp(mean log((target + .01) / (rgb + .01))) + 3) / 4 : 1          (rgb detached in it, as in the reference)
 *   sums[0] = sum (rgb s - target)^2                                     -> img_loss  = sums[0] / (3 B h w)
 *   sums[1] = - sum (m log l + (1 - m) log(1 - l)), l = clamp(label, .001, .999)     -> loop_loss = sums[1] / (B h w)      (C == 4)
 * grad (B,h,w,C) contiguous: d sums[0] / d rgb in channels 0-2, d sums[1] / d label in channel 3 (the caller divides by the counts and
 * multiplies by the upstream gradients).  log_sum: device double scratch; sums: device double[2]; both overwritten. */
int vl3d_stage1_loss(int32_t B, int32_t C, int32_t h, int32_t w, const float *rgbl, int64_t sb, int64_t sc, int64_t sp, const float *target,
                     const float *target_mask, int32_t scale_invariant, double *log_sum, double *sums, float *grad, vl3d_stream_t stream);

/* The whole scalar head of a stage-1 iteration (train_3d.py:200-232 on MPI.py:596-652's outputs) in one sweep over the crop's pixels:
 *   img_loss, loop_loss as vl3d_stage1_loss; sparsity, density as vl3d_pixel_terms (sparsity times `sparsity_scale` = 1 / sqrt(mpi_d));
 *   rgb_smooth = coef0 s0 + coef1 s1, a_smooth = coef2 s2 + coef3 s3 from the render's four smoothness sums (MPI.py:605-619);
 *   total = w_img img + w_loop loop + w_sparsity sparsity + w_density density + w_rgb_smooth rgb_smooth + w_a_smooth a_smooth.
 * Inputs are the render's own outputs: rgb (B,h,w,3), label (B,h,w) | NULL, alpha (B,h,w) | NULL, alpha_sums (B,h,w,2) | NULL,
 * smooth_sums float[4] | NULL; target (B,3,h,w) and target_mask (B,h,w) | NULL (with label) through strides in floats (t_sb batch, t_sc channel,
 * t_sr row; m_sb, m_sr; unit column stride: a crop of a larger image needs no copy).  Outputs: out float[8] = (total, img, loop,
 * w sparsity, w density, w rgb_smooth, w a_smooth, gain) and the gradients of `total` w.r.t. every given input, FINAL for a unit upstream
 * gradient (the caller scales them by the actual one: vl3d_scale_inplace skips the pass when it is 1).  scratch: 6 device doubles. */
typedef struct vl3d_stage1_objective_desc {
    int32_t B, h, w;
    int32_t scale_invariant;
    float w_img, w_loop, w_sparsity, w_density, w_rgb_smooth, w_a_smooth;
    float sparsity_scale, eps;
    float smooth_coef[4];
} vl3d_stage1_objective_desc;
int vl3d_stage1_objective(const vl3d_stage1_objective_desc *desc, const float *rgb, const float *label, const float *alpha, const float *alpha_sums,
                          const float *smooth_sums, const float *target, int64_t t_sb, int64_t t_sc, int64_t t_sr, const float *target_mask,
                          int64_t m_sb, int64_t m_sr, double *scratch, float *out, float *grad_rgb,
                          float *grad_label, float *grad_alpha, float *grad_alpha_sums, float *grad_smooth, vl3d_stream_t stream);

/* The weighted total of a stage-2 iteration (train_3dvid.py:230-240: loss = swd + sum_k weight_k term_k) over n <= 16 device scalars
 * v = (*main_term, rest[0..n-2]) with device coefficients coef[n]; term i belongs to group (groups >> 4 i) & 15 of `ngroups` <= 16 (a smoothness
 * mean is two of the render's four sums): out[0] = sum_i coef_i v_i, out[1 + g] = the sum of group g's terms (one launch instead of a stack, a
 * multiply and a sum on one-element tensors and their autograd mirrors).  Backward: grad_terms[i] = coef_i * *grad_total. */
int vl3d_linear_head_fwd(int32_t n, uint64_t groups, int32_t ngroups, const float *main_term, const float *rest, const float *coef, float *out,
                         vl3d_stream_t stream);
int vl3d_linear_head_bwd(int32_t n, const float *coef, const float *grad_total, float *grad_terms, vl3d_stream_t stream);

/* robust_lossfun (utils_vid.py:10-26) fused with the mean (utils_vid.py:348).
 * kind: 0 'mse', 1 'abs', 2 general Barron with float rou (rou==0 and rou==2 special-cased as the reference).
 * loss_sum: device double, overwritten with sum over n elements of rho(x - y2x). */
enum { VL3D_RHO_MSE = 0, VL3D_RHO_ABS = 1, VL3D_RHO_BARRON = 2 };
int vl3d_robust_fwd(int64_t n, const float *x, const float *y2x, int32_t kind, float rou, float scale,
                    double *loss_sum, vl3d_stream_t stream);
/* grad_x[i] = rho'(x[i]-y2x[i]) * (*grad_out) * inv_n ; grad_out is a DEVICE scalar (no host sync). */
int vl3d_robust_bwd(int64_t n, const float *x, const float *y2x, int32_t kind, float rou, float scale,
                    const float *grad_out, float inv_n, float *grad_x, vl3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VL3D_H */
