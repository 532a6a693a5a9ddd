// C ABI of the fused MPI/MPV render (include/vl3d.h): argument checks, scratch layout and the dispatch to the convention that was
// compiled for (coord_mode, border_mode, act_order) -- one translation unit each, vl3d_render_c*.hip.  Kernels: vl3d_render_core.h.
//
// Replaces (reference, /root/reference): MPV.py:351-454 (planar geometry) ==
// utils_mpi.py:159-176 (warp_homography) + utils_mpi.py:92-107 (overcompose), and their autograd.
#include <string>
#include "vl3d_render_core.h"

using vl3d_render_detail::RenderArgs;

namespace {

// conventions a caller can reach (every other combination of the three enums returns VL3D_EUNSUPPORTED):
//   (UTILS_MPI, ZEROS,   PRE )  the reference's utils_mpi chain                       sigmoid/sigmoid, none/none, none/sigmoid
//   (AFFINE,    HARDCUT, POST)  the reference's MPV.py planar convention              all nine activation pairs
//   (AFFINE_PLANES, HARDCUT, POST)  MPV.py atlas-cell sampling: per-plane texel transform + quad extent   sigmoid/sigmoid
//   (UTILS_MPI, ZEROS,   POST), (UTILS_MPI, HARDCUT, PRE)  cross-check conventions     sigmoid/sigmoid
int dispatch(bool bwd, const vl3d_render_desc *d, const RenderArgs &a, hipStream_t s) {
    using namespace vl3d_render_detail;
    const int c = d->coord_mode, b = d->border_mode, o = d->act_order;
    if (c == VL3D_COORD_UTILS_MPI && b == VL3D_BORDER_ZEROS && o == VL3D_ACT_PRE) return conv_utils_zeros_pre(bwd, d, a, s);
    if (c == VL3D_COORD_UTILS_MPI && b == VL3D_BORDER_ZEROS && o == VL3D_ACT_POST) return conv_utils_zeros_post(bwd, d, a, s);
    if (c == VL3D_COORD_UTILS_MPI && b == VL3D_BORDER_HARDCUT && o == VL3D_ACT_PRE) return conv_utils_hardcut_pre(bwd, d, a, s);
    if (c == VL3D_COORD_AFFINE_PLANES && b == VL3D_BORDER_HARDCUT && o == VL3D_ACT_POST) return conv_affine_planes_hardcut_post(bwd, d, a, s);
    if (c == VL3D_COORD_AFFINE && b == VL3D_BORDER_HARDCUT && o == VL3D_ACT_POST) {
        if (d->rgb_act == VL3D_ACT_SIGMOID && d->alpha_act == VL3D_ACT_SIGMOID) return conv_affine_hardcut_post_sig(bwd, d, a, s);
        return conv_affine_hardcut_post_other(bwd, d, a, s);
    }
    vl3d_set_error("unsupported (coord_mode, border_mode, act_order): built are (utils_mpi, zeros, pre|post), (utils_mpi, hardcut, pre), "
                   "(affine, hardcut, post)");
    return VL3D_EUNSUPPORTED;
}

int check_desc(const vl3d_render_desc *d) {
    VL3D_REQUIRE(d != nullptr, "null render desc");
    if (vl3d_check_variant(d->variant) != VL3D_OK) return VL3D_EINVAL;
    VL3D_REQUIRE(d->D > 0 && d->T > 0 && d->Hs > 0 && d->Ws > 0 && d->H > 0 && d->W > 0, "non-positive render dims");
    VL3D_REQUIRE((int64_t)d->Hs * d->Ws < (1ll << 31), "plane too large for 32-bit texel index");
    VL3D_REQUIRE(d->Hs < (1 << 24) && d->Ws < (1 << 24), "plane side too large (24-bit row arithmetic)");
    VL3D_REQUIRE(d->stack_dtype == VL3D_F32 || d->stack_dtype == VL3D_F16, "stack_dtype must be VL3D_F32 or VL3D_F16");
    VL3D_REQUIRE(d->coord_mode != VL3D_COORD_AFFINE_PLANES || (d->sx == 1.0f && d->sy == 1.0f && d->ox == 0.0f && d->oy == 0.0f),
                 "VL3D_COORD_AFFINE_PLANES: the texel transforms live in the per-plane records (sx = sy = 1, ox = oy = 0)");
    VL3D_REQUIRE(d->stack_dtype == VL3D_F32 || (d->rgb_act == VL3D_ACT_SIGMOID && d->alpha_act == VL3D_ACT_SIGMOID),
                 "fp16 plane stacks are implemented for the shipped (sigmoid, sigmoid) activations only");
    // the packed fp16 tap load fetches texels x0 and x0+1 of a row with one 16-byte read (load_taps2): a row needs two texels
    VL3D_REQUIRE(d->stack_dtype == VL3D_F32 || d->Ws >= 2, "fp16 plane stacks need Ws >= 2");
    VL3D_REQUIRE(d->uv_noise_seed == 0 || (d->border_mode == VL3D_BORDER_HARDCUT && d->coord_mode != VL3D_COORD_UTILS_MPI),
                 "uv_noise_seed (add_uv_noise, MPV.py:420-423) belongs to the planar MPV / MPI convention: affine coordinates, VL3D_BORDER_HARDCUT");
    return VL3D_OK;
}

RenderArgs make_args(const vl3d_render_desc *d) {
    RenderArgs a{};
    a.D = d->D; a.T = d->T; a.Hs = d->Hs; a.Ws = d->Ws; a.H = d->H; a.W = d->W;
    a.Tstride = d->T;
    a.row0 = d->row0; a.col0 = d->col0;
    a.pc = d->pixel_center; a.sx = d->sx; a.sy = d->sy; a.ox = d->ox; a.oy = d->oy;
    a.uv_seed = d->uv_noise_seed;
    return a;
}

}  // namespace

// the quad grid is laid over the whole plane; the stack may be a texel window of it (desc->cull_*: crop-aware training renders
// from a compact copy of the window the crop can reach, videoloop3d_amd/optim.py)
static void set_cull_geometry(RenderArgs &a, const vl3d_render_desc *desc, int32_t QH, int32_t QW) {
    const bool win = desc->cull_Hs > 0 && desc->cull_Ws > 0;
    a.q_Hs = win ? desc->cull_Hs : desc->Hs;
    a.q_Ws = win ? desc->cull_Ws : desc->Ws;
    a.q_x0 = win ? (float)desc->cull_col0 : 0.0f;
    a.q_y0 = win ? (float)desc->cull_row0 : 0.0f;
    a.q_th = a.q_tw = 0;
    if (QH < 0 && QW < 0) {
        // TILE-EXACT layout (include/vl3d.h): the plane is |QH| x |QW| tiles of th x tw texels, every quad owning its border row / column; the
        // homography + (sx, ox) give LATTICE coordinates (a quad spans tw - 1 of them), the kernels add the quad index (make_taps_i)
        a.QH = -QH; a.QW = -QW;
        a.q_th = a.q_Hs / a.QH; a.q_tw = a.q_Ws / a.QW;
        a.q_inv_cw = 1.0f / (float)(a.q_tw > 1 ? a.q_tw - 1 : 1);
        a.q_inv_ch = 1.0f / (float)(a.q_th > 1 ? a.q_th - 1 : 1);
        return;
    }
    a.QH = QH; a.QW = QW;
    a.q_inv_cw = (float)QW / (float)(a.q_Ws > 1 ? a.q_Ws - 1 : 1);
    a.q_inv_ch = (float)QH / (float)(a.q_Hs > 1 ? a.q_Hs - 1 : 1);
}

static int check_cull(const vl3d_render_desc *desc, const uint8_t *quad_keep, int32_t QH, int32_t QW) {
    if (!quad_keep) return VL3D_OK;
    VL3D_REQUIRE(desc->coord_mode != VL3D_COORD_AFFINE_PLANES, "tile culling is not available with per-plane texel transforms");
    VL3D_REQUIRE((QH > 0 && QW > 0) || (QH < 0 && QW < 0), "tile culling: empty quad grid (both positive, or both negative for the tile-exact layout)");
    if (QH < 0) {
        const int pH = desc->cull_Hs > 0 ? desc->cull_Hs : desc->Hs, pW = desc->cull_Ws > 0 ? desc->cull_Ws : desc->Ws;
        VL3D_REQUIRE(pH % (-QH) == 0 && pW % (-QW) == 0 && pH / (-QH) >= 2 && pW / (-QW) >= 2,
                     "tile-exact layout: the plane must be |QH| x |QW| whole tiles of at least 2 x 2 texels");
        VL3D_REQUIRE(desc->coord_mode == VL3D_COORD_AFFINE && desc->border_mode == VL3D_BORDER_HARDCUT,
                     "tile-exact layout: the planar MPV / MPI convention only (VL3D_COORD_AFFINE, VL3D_BORDER_HARDCUT)");
    }
    VL3D_REQUIRE((desc->cull_Hs == 0 && desc->cull_Ws == 0) ||
                     (desc->cull_row0 >= 0 && desc->cull_col0 >= 0 && desc->cull_row0 + desc->Hs <= desc->cull_Hs && desc->cull_col0 + desc->Ws <= desc->cull_Ws),
                 "tile culling: the stack window (cull_row0, cull_col0) + (Hs, Ws) leaves the plane (cull_Hs, cull_Ws)");
    VL3D_REQUIRE(desc->D <= 128, "tile culling supports at most 128 planes");
    return VL3D_OK;
}

extern "C" int64_t vl3d_render_cull_scratch_bytes(const vl3d_render_desc *desc) {
    if (!desc || desc->H <= 0 || desc->W <= 0) return 0;
    // two 64-bit plane masks per forward workgroup, sized for the smallest workgroup any forward variant uses (64 x 4 pixels)
    return (int64_t)((desc->W + 63) / 64) * ((desc->H + 3) / 4) * 16;
}

static int render_fwd_impl(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep, int32_t QH,
                           int32_t QW, void *cull_scratch, float *rgb, float *alpha, float *alpha_sums, vl3d_stream_t stream, int32_t frame0 = 0,
                           int32_t T_alloc = 0) {
    int rc = check_desc(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(stack && homos && rgb && alpha, "null pointer passed to vl3d_render_fwd");
    rc = check_cull(desc, quad_keep, QH, QW);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(!quad_keep || cull_scratch, "tile culling: the forward needs vl3d_render_cull_scratch_bytes() of scratch");
    RenderArgs a = make_args(desc);
    a.stack = (const float *)stack; a.homos = homos; a.rgb = rgb; a.alpha = alpha; a.asum = alpha_sums;
    a.quad_keep = quad_keep; a.QH = QH; a.QW = QW;
    set_cull_geometry(a, desc, QH, QW); a.cull_masks = (const unsigned long long *)cull_scratch;
    a.fwd_variant = (desc->variant >> 8) & 0xf;
    a.ablate = (desc->variant >> 4) & 0xf;
    VL3D_REQUIRE((int64_t)desc->Hs * desc->Ws * 16 < (1ll << 32), "frame too large for 32-bit byte offsets");
    a.g_f16 = desc->stack_dtype == VL3D_F16;
    if (T_alloc > 0) {      // vl3d_render_fwd_frames: frames frame0 .. frame0 + T - 1 of a (D, T_alloc, Hs, Ws, 4) allocation, read in place
        VL3D_REQUIRE(frame0 >= 0 && frame0 + desc->T <= T_alloc, "vl3d_render_fwd_frames: the run of frames leaves the clip");
        a.stack = reinterpret_cast<const float *>(reinterpret_cast<const char *>(stack) + (size_t)frame0 * desc->Hs * desc->Ws * (a.g_f16 ? 8 : 16));
        a.Tstride = T_alloc;
    }
    rc = dispatch(false, desc, a, (hipStream_t)stream);
    if (rc != VL3D_OK) return rc;
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_render_fwd_frames(const vl3d_render_desc *desc, const void *stack, int32_t frame0, int32_t T_alloc, const float *homos,
                                      float *rgb, float *alpha, vl3d_stream_t stream) {
    VL3D_REQUIRE(T_alloc > 0, "vl3d_render_fwd_frames: T_alloc must be the clip length of the stack allocation");
    return render_fwd_impl(desc, stack, homos, nullptr, 0, 0, nullptr, rgb, alpha, nullptr, stream, frame0, T_alloc);
}

extern "C" int vl3d_render_fwd_frames_culled(const vl3d_render_desc *desc, const void *stack, int32_t frame0, int32_t T_alloc, const float *homos,
                                             const uint8_t *quad_keep, int32_t QH, int32_t QW, void *cull_scratch, float *rgb, float *alpha,
                                             vl3d_stream_t stream) {
    VL3D_REQUIRE(T_alloc > 0, "vl3d_render_fwd_frames_culled: T_alloc must be the clip length of the stack allocation");
    VL3D_REQUIRE(quad_keep != nullptr, "vl3d_render_fwd_frames_culled: null quad map (vl3d_render_fwd_frames renders a dense model)");
    return render_fwd_impl(desc, stack, homos, quad_keep, QH, QW, cull_scratch, rgb, alpha, nullptr, stream, frame0, T_alloc);
}

extern "C" int vl3d_render_fwd(const vl3d_render_desc *desc, const void *stack, const float *homos,
                               float *rgb, float *alpha, float *alpha_sums, vl3d_stream_t stream) {
    return render_fwd_impl(desc, stack, homos, nullptr, 0, 0, nullptr, rgb, alpha, alpha_sums, stream);
}

extern "C" int vl3d_render_fwd_culled(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep,
                                      int32_t QH, int32_t QW, void *cull_scratch, float *rgb, float *alpha, float *alpha_sums,
                                      vl3d_stream_t stream) {
    return render_fwd_impl(desc, stack, homos, quad_keep, QH, QW, cull_scratch, rgb, alpha, alpha_sums, stream);
}

// scratch layout: per-plane records | one int4 window per (tile, plane), sized for the smallest tile interior any variant
// uses (60 x 6) | (256-byte aligned) owner table, one uint16 per (plane, texel)
static int64_t owner_table_off(const vl3d_render_desc *desc) {
    // window records: the largest tile count of any backward kernel (owned pixels per tile: 62 x 14 / 60 x 12 for the one-frame tile
    // kernel without / with regularisers, 30 x 14 / 28 x 12 for the frame pairs, 62 x 6 for the flat one-frame regions of a single frame)
    auto ntiles = [&](int iw, int ih) { return (int64_t)((desc->W + iw - 1) / iw) * ((desc->H + ih - 1) / ih); };
    int64_t tiles = ntiles(62, 14);
    for (const auto &t : {ntiles(60, 12), ntiles(30, 14), ntiles(28, 12), ntiles(62, 6)}) tiles = t > tiles ? t : tiles;
    const int64_t b = (int64_t)plan_win_off(desc->D) * sizeof(float) + tiles * desc->D * 16;
    return (b + 255) & ~(int64_t)255;
}

extern "C" int64_t vl3d_render_bwd_scratch_bytes(const vl3d_render_desc *desc) {
    if (!desc || desc->D <= 0 || desc->H <= 0 || desc->W <= 0) return 0;
    // + padding: the gather's unconditional prefetch reads up to 16 rows + 64 texels past a window's corner
    return owner_table_off(desc) + ((int64_t)desc->D * desc->Hs * desc->Ws + 16 * (int64_t)desc->Ws + 64) * 2;
}

static int render_reg_fwd_impl(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep, int32_t QH,
                               int32_t QW, double *sums, void *reg_state, vl3d_stream_t stream);

// reg_state: flags [H][W] u8 | coverage masks [H][W][2] u64 | sign words [ceil(D/4)][T][H][W][4] u16 | patch words (same shape) (256-byte aligned parts)
struct RegLayout { int64_t masks, signs, patch, total; };
static RegLayout reg_layout(const vl3d_render_desc *d) {
    auto up = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
    const int64_t px = (int64_t)d->H * d->W, words = (int64_t)((d->D + 3) / 4) * d->T * px * 8;      // groups of four planes
    RegLayout l;
    l.masks = up(px);
    l.signs = l.masks + up(px * 16);
    l.patch = l.signs + up(words);
    l.total = l.patch + up(words);
    return l;
}
static void set_reg_state(RenderArgs &a, const vl3d_render_desc *d, const void *reg_state) {
    char *b = reinterpret_cast<char *>(const_cast<void *>(reg_state));
    const RegLayout l = reg_layout(d);
    a.reg_flags = reinterpret_cast<unsigned char *>(b);
    a.reg_masks = reinterpret_cast<unsigned long long *>(b + l.masks);
    a.reg_signs = reinterpret_cast<unsigned short *>(b + l.signs);
    a.reg_patch = reinterpret_cast<unsigned short *>(b + l.patch);
}
extern "C" int64_t vl3d_render_reg_state_bytes(const vl3d_render_desc *desc) {
    if (!desc || desc->D <= 0 || desc->T <= 0 || desc->H <= 0 || desc->W <= 0) return 0;
    return reg_layout(desc).total;
}

extern "C" int vl3d_render_fwd_reg(const vl3d_render_desc *desc, const void *stack, const float *homos, float *rgb, float *alpha,
                                   float *alpha_sums, double *sums, void *reg_state, vl3d_stream_t stream) {
    int rc = check_desc(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(stack && homos && rgb && alpha && sums && reg_state, "null pointer passed to vl3d_render_fwd_reg");
    VL3D_REQUIRE(desc->D <= 128, "the layer regularisers support at most 128 planes (coverage masks)");
    VL3D_REQUIRE((int64_t)desc->Hs * desc->Ws * 16 < (1ll << 32), "frame too large for 32-bit byte offsets");
    RenderArgs a = make_args(desc);
    a.stack = (const float *)stack; a.homos = homos; a.rgb = rgb; a.alpha = alpha; a.asum = alpha_sums; a.reg_sums = sums;
    set_reg_state(a, desc, reg_state);
    a.g_f16 = desc->stack_dtype == VL3D_F16;
    a.reg_fwd = 2;
    // (the sums are cleared by reg_masks_k, the first kernel of the regulariser forward)
    rc = dispatch(false, desc, a, (hipStream_t)stream);
    if (rc != VL3D_OK) return rc;
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

// ---- stage 1's loop mask as a fifth composited channel (MPI.py:115-117, 568-583) ---------------------------------------------------
static int check_mask_desc(const vl3d_render_desc *d, const char *who) {
    if (d->coord_mode == VL3D_COORD_AFFINE && d->border_mode == VL3D_BORDER_HARDCUT && d->act_order == VL3D_ACT_POST &&
        d->rgb_act == VL3D_ACT_SIGMOID && d->alpha_act == VL3D_ACT_SIGMOID && d->stack_dtype == VL3D_F32)
        return VL3D_OK;
    vl3d_set_error((std::string(who) + ": the loop-mask channel is built for the planar convention stage 1 ships -- (affine, hardcut, post), "
                   "sigmoid / sigmoid, fp32 stack (MPI.py:452-594); render the label in a pass of its own otherwise").c_str());
    return VL3D_EUNSUPPORTED;
}

extern "C" int vl3d_render_fwd_mask(const vl3d_render_desc *desc, const void *stack, const float *mask, const float *homos, float *rgb,
                                    float *alpha, float *label, float *alpha_sums, double *sums, void *reg_state, vl3d_stream_t stream) {
    int rc = check_desc(desc);
    if (rc != VL3D_OK) return rc;
    rc = check_mask_desc(desc, "vl3d_render_fwd_mask");
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(stack && mask && homos && rgb && alpha && label, "null pointer passed to vl3d_render_fwd_mask");
    VL3D_REQUIRE(desc->uv_noise_seed == 0, "vl3d_render_fwd_mask: add_uv_noise jitters the colour samples only (MPI.py:519-522, 568-572): render the label in a pass of its own");
    VL3D_REQUIRE((sums == nullptr) == (reg_state == nullptr), "vl3d_render_fwd_mask: sums and reg_state come together (both NULL: no layer regularisers)");
    VL3D_REQUIRE(!sums || desc->D <= 128, "the layer regularisers support at most 128 planes (coverage masks)");
    VL3D_REQUIRE((int64_t)desc->Hs * desc->Ws * 16 < (1ll << 32), "frame too large for 32-bit byte offsets");
    RenderArgs a = make_args(desc);
    a.stack = (const float *)stack; a.homos = homos; a.rgb = rgb; a.alpha = alpha; a.asum = alpha_sums;
    a.mask = mask; a.label = label;
    a.fwd_variant = (desc->variant >> 8) & 0xf;
    if (sums) {
        a.reg_sums = sums;
        set_reg_state(a, desc, reg_state);
        a.reg_fwd = 2;
        // (the sums are cleared by reg_masks_k, the first kernel of the regulariser forward)
    }
    rc = dispatch(false, desc, a, (hipStream_t)stream);
    if (rc != VL3D_OK) return rc;
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_render_bwd_mask(const vl3d_render_desc *desc, const void *stack, const float *mask, const float *homos, const float *rgb,
                                    const float *alpha, const float *grad_rgb, const float *grad_alpha, const float *grad_label,
                                    const float *grad_reg, const void *reg_state, const float *grad_alpha_sums, float *grad_stack,
                                    float *grad_mask, void *scratch, int64_t scratch_bytes, vl3d_stream_t stream) {
    int rc = check_desc(desc);
    if (rc != VL3D_OK) return rc;
    rc = check_mask_desc(desc, "vl3d_render_bwd_mask");
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(stack && mask && homos && rgb && alpha && grad_rgb && grad_label && grad_stack && grad_mask, "null pointer passed to vl3d_render_bwd_mask");
    VL3D_REQUIRE(!grad_reg || reg_state, "vl3d_render_bwd_mask: grad_reg needs the reg_state the forward with regularisers filled");
    VL3D_REQUIRE((int64_t)desc->Hs * desc->Ws * 16 < (1ll << 32), "frame too large for 32-bit byte offsets");
    RenderArgs a = make_args(desc);
    if (grad_reg) set_reg_state(a, desc, reg_state);
    a.stack = (const float *)stack; a.homos = homos;
    a.rgb = const_cast<float *>(rgb); a.alpha = const_cast<float *>(alpha);
    a.g_rgb = grad_rgb; a.g_alpha = grad_alpha; a.g_reg = grad_reg; a.g_asum = grad_alpha_sums; a.g_stack = grad_stack;
    a.mask = mask; a.g_label = grad_label; a.g_mask = grad_mask;
    VL3D_REQUIRE(desc->uv_noise_seed == 0, "vl3d_render_bwd_mask: add_uv_noise jitters the colour samples only (MPI.py:519-522, 568-572): render the label in a pass of its own");
    const bool want_tile = (desc->variant & 0xf) != 1 && scratch != nullptr && scratch_bytes >= vl3d_render_bwd_scratch_bytes(desc);
    a.gather9 = (desc->variant & 0xf) == 4;
    a.owner4 = (desc->variant & 0xf) != 3 && (desc->variant & 0xf) != 4;
    if (want_tile) {
        a.plan = (const float *)scratch;
        a.owner = reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(scratch) + owner_table_off(desc));
        // flat 64 x 8 regions (two 512-thread workgroups per CU at this instantiation's 128 registers) unless variant 3 / 4 ask for the 16 rows:
        // stage-1 iterations +3.3 % at the reference's crop, +2 % for a 720p frame (profiles/r05d_s1_mask_rows.txt), same bits
        a.tile_rows = ((desc->variant & 0xf) == 3 || (desc->variant & 0xf) == 4) ? 16 : 8;
    } else {
        const size_t texels = (size_t)desc->D * desc->T * desc->Hs * desc->Ws;
        VL3D_HIP(hipMemsetAsync(grad_stack, 0, texels * 16, (hipStream_t)stream));
        VL3D_HIP(hipMemsetAsync(grad_mask, 0, texels * 4, (hipStream_t)stream));
    }
    rc = dispatch(true, desc, a, (hipStream_t)stream);
    if (rc != VL3D_OK) return rc;
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_render_reg_fwd(const vl3d_render_desc *desc, const void *stack, const float *homos, double *sums, void *reg_state,
                                   vl3d_stream_t stream) {
    return render_reg_fwd_impl(desc, stack, homos, nullptr, 0, 0, sums, reg_state, stream);
}

extern "C" int vl3d_render_reg_fwd_culled(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep,
                                          int32_t QH, int32_t QW, double *sums, void *reg_state, vl3d_stream_t stream) {
    return render_reg_fwd_impl(desc, stack, homos, quad_keep, QH, QW, sums, reg_state, stream);
}

extern "C" int vl3d_render_fwd_reg_culled(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep, int32_t QH,
                                          int32_t QW, float *rgb, float *alpha, float *alpha_sums, double *sums, void *reg_state,
                                          vl3d_stream_t stream) {
    int rc = check_desc(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(stack && homos && quad_keep && rgb && alpha && sums && reg_state, "null pointer passed to vl3d_render_fwd_reg_culled");
    VL3D_REQUIRE(desc->D <= 128, "the layer regularisers support at most 128 planes (coverage masks)");
    rc = check_cull(desc, quad_keep, QH, QW);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE((int64_t)desc->Hs * desc->Ws * 16 < (1ll << 32), "frame too large for 32-bit byte offsets");
    RenderArgs a = make_args(desc);
    a.stack = (const float *)stack; a.homos = homos; a.rgb = rgb; a.alpha = alpha; a.asum = alpha_sums; a.reg_sums = sums;
    a.quad_keep = quad_keep; a.QH = QH; a.QW = QW;
    set_cull_geometry(a, desc, QH, QW);
    set_reg_state(a, desc, reg_state);
    // (the sums are cleared by reg_masks_k, the first kernel of the regulariser forward)
    a.reg_fwd = 3;
    a.g_f16 = desc->stack_dtype == VL3D_F16;
    rc = dispatch(false, desc, a, (hipStream_t)stream);
    if (rc != VL3D_OK) return rc;
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

static int render_reg_fwd_impl(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep, int32_t QH,
                               int32_t QW, double *sums, void *reg_state, vl3d_stream_t stream) {
    int rc = check_desc(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(stack && homos && sums && reg_state, "null pointer passed to vl3d_render_reg_fwd");
    VL3D_REQUIRE(desc->D <= 128, "the layer regularisers support at most 128 planes (coverage masks)");
    rc = check_cull(desc, quad_keep, QH, QW);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE((int64_t)desc->Hs * desc->Ws * 16 < (1ll << 32), "frame too large for 32-bit byte offsets");
    RenderArgs a = make_args(desc);
    a.stack = (const float *)stack; a.homos = homos; a.reg_sums = sums;
    a.quad_keep = quad_keep; a.QH = QH; a.QW = QW;
    set_cull_geometry(a, desc, QH, QW);
    set_reg_state(a, desc, reg_state);
    // (the sums are cleared by reg_masks_k, the first kernel of the regulariser forward)
    a.reg_fwd = 1;
    a.g_f16 = desc->stack_dtype == VL3D_F16;
    rc = dispatch(false, desc, a, (hipStream_t)stream);
    if (rc != VL3D_OK) return rc;
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

static int render_bwd_impl(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep, int32_t QH,
                           int32_t QW, const float *rgb, const float *alpha, const float *grad_rgb,
                           const float *grad_alpha, const float *grad_reg, const void *reg_state, const float *grad_alpha_sums,
                           float *grad_stack, void *scratch, int64_t scratch_bytes, vl3d_stream_t stream);

extern "C" int vl3d_render_bwd(const vl3d_render_desc *desc, const void *stack, const float *homos,
                               const float *rgb, const float *alpha, const float *grad_rgb,
                               const float *grad_alpha, const float *grad_reg, const void *reg_state, const float *grad_alpha_sums,
                               float *grad_stack, void *scratch, int64_t scratch_bytes, vl3d_stream_t stream) {
    return render_bwd_impl(desc, stack, homos, nullptr, 0, 0, rgb, alpha, grad_rgb, grad_alpha, grad_reg, reg_state, grad_alpha_sums, grad_stack,
                           scratch, scratch_bytes, stream);
}

extern "C" int vl3d_render_bwd_culled(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep,
                                      int32_t QH, int32_t QW, const float *rgb, const float *alpha, const float *grad_rgb,
                                      const float *grad_alpha, const float *grad_reg, const void *reg_state, const float *grad_alpha_sums,
                                      float *grad_stack, void *scratch, int64_t scratch_bytes, vl3d_stream_t stream) {
    return render_bwd_impl(desc, stack, homos, quad_keep, QH, QW, rgb, alpha, grad_rgb, grad_alpha, grad_reg, reg_state, grad_alpha_sums,
                           grad_stack, scratch, scratch_bytes, stream);
}

static int render_bwd_impl(const vl3d_render_desc *desc, const void *stack, const float *homos, const uint8_t *quad_keep, int32_t QH,
                           int32_t QW, const float *rgb, const float *alpha, const float *grad_rgb,
                           const float *grad_alpha, const float *grad_reg, const void *reg_state, const float *grad_alpha_sums,
                           float *grad_stack, void *scratch, int64_t scratch_bytes, vl3d_stream_t stream) {
    int rc = check_desc(desc);
    if (rc != VL3D_OK) return rc;
    rc = check_cull(desc, quad_keep, QH, QW);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(stack && homos && rgb && alpha && grad_rgb && grad_stack, "null pointer passed to vl3d_render_bwd");
    VL3D_REQUIRE(!grad_reg || reg_state, "vl3d_render_bwd: grad_reg needs the reg_state the forward with regularisers filled");
    RenderArgs a = make_args(desc);
    if (grad_reg) set_reg_state(a, desc, reg_state);
    a.stack = (const float *)stack; a.homos = homos;
    a.rgb = const_cast<float *>(rgb); a.alpha = const_cast<float *>(alpha);
    a.g_rgb = grad_rgb; a.g_alpha = grad_alpha; a.g_reg = grad_reg; a.g_asum = grad_alpha_sums; a.g_stack = grad_stack;
    a.quad_keep = quad_keep; a.QH = QH; a.QW = QW;
    set_cull_geometry(a, desc, QH, QW);
    a.g_f16 = desc->stack_dtype == VL3D_F16;
    a.grad_culled_unwritten = (quad_keep && (desc->grad_flags & VL3D_GRAD_CULLED_UNWRITTEN)) ? 1 : 0;
    // variant: 0 auto (tile kernel when its on-device plan says feasible, else atomics), 1 force atomics,
    //          3 tile kernel (16-row regions, one frame per thread), 4 = 3 without the 2x2 gather, 2 = the tile kernel in flat 64 x 8
    //          regions (round 1 measured them 18.9 vs 16.8 ms at cfg3 and dropped them; round 5 brought them back for ONE frame, T = 1,
    //          where 2520 half-size workgroups on 1024 slots beat 1092 on 512: 0.380 vs 0.394 ms, same bits -- the default there)
    // (add_uv_noise: a jittered tap can leave the 1-pixel halo the owner-computes kernels stage -- the atomics kernel takes the call)
    const bool want_tile = (desc->variant & 0xf) != 1 && scratch != nullptr && scratch_bytes >= vl3d_render_bwd_scratch_bytes(desc) &&
                           desc->uv_noise_seed == 0;
    a.ablate = (desc->variant >> 4) & 0xf;
    a.gather9 = (desc->variant & 0xf) == 4;
    a.owner4 = (desc->variant & 0xf) != 3 && (desc->variant & 0xf) != 4;
    if (want_tile) {
        a.plan = (const float *)scratch;
        a.owner = reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(scratch) + owner_table_off(desc));
        const int bv = desc->variant & 0xf;
        a.tile_rows = bv == 0 ? 17 : (bv == 5 ? 18 : (bv == 2 ? 8 : 16));     // 17: 16 rows, frame-pair kernels / flat one-frame regions allowed; 18: 32-wide one-frame regions; 8: flat 64 x 8 one-frame regions
    } else {
        a.plan = nullptr;
        a.tile_rows = 0;
        const size_t bytes = (size_t)desc->D * desc->T * desc->Hs * desc->Ws * (desc->stack_dtype == VL3D_F16 ? 8 : 16);
        VL3D_HIP(hipMemsetAsync(grad_stack, 0, bytes, (hipStream_t)stream));
    }
    rc = dispatch(true, desc, a, (hipStream_t)stream);
    if (rc != VL3D_OK) return rc;
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

// ---- the optimiser step inside the backward (include/vl3d.h) ---------------------------------------------------------------------------
extern "C" int64_t vl3d_render_bwd_adam_class_bytes(const vl3d_render_desc *desc) {
    if (!desc || desc->D <= 0 || desc->Hs <= 0 || desc->Ws <= 0) return 0;
    // one 8-byte record per (plane, texel) of the compact window, padded like the owner table (the gather's unconditional prefetch)
    return ((int64_t)desc->D * desc->Hs * desc->Ws + 16 * (int64_t)desc->Ws + 64) * 8;
}

extern "C" int vl3d_render_bwd_adam(const vl3d_render_desc *desc, const void *stack, const float *homos, const float *rgb, const float *alpha,
                                    const float *grad_rgb, const float *grad_alpha, const float *grad_reg, const void *reg_state,
                                    const float *grad_alpha_sums, float *grad_stack, void *scratch, int64_t scratch_bytes,
                                    const vl3d_adam_window *adam, vl3d_stream_t stream) {
    int rc = check_desc(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(adam != nullptr, "vl3d_render_bwd_adam: null adam window");
    if (desc->uv_noise_seed) {
        vl3d_set_error("vl3d_render_bwd_adam: add_uv_noise takes the atomics backward (vl3d_render_bwd(_culled) + vl3d_adam_window_step)");
        return VL3D_EUNSUPPORTED;
    }
    VL3D_REQUIRE(stack && homos && rgb && alpha && grad_rgb && grad_stack && scratch, "null pointer passed to vl3d_render_bwd_adam");
    VL3D_REQUIRE(!grad_reg || reg_state, "vl3d_render_bwd_adam: grad_reg needs the reg_state the forward with regularisers filled");
    const uint8_t *qk = adam->quad_keep;
    const int bv = desc->variant & 0xf;
    if (!(desc->coord_mode == VL3D_COORD_AFFINE && desc->border_mode == VL3D_BORDER_HARDCUT && desc->act_order == VL3D_ACT_POST &&
          desc->rgb_act == VL3D_ACT_SIGMOID && desc->alpha_act == VL3D_ACT_SIGMOID && desc->stack_dtype == VL3D_F32 &&
          (qk ? (bv == 0 || bv == 3 || bv == 5) : (desc->T >= 2 && bv == 0)))) {
        vl3d_set_error("vl3d_render_bwd_adam: built for the stage-2 iteration -- (affine, hardcut, post), sigmoid / sigmoid, fp32 stack, "
                       "T >= 2 and variant 0 for a dense model; use vl3d_render_bwd(_culled) + vl3d_adam_window_step otherwise");
        return VL3D_EUNSUPPORTED;
    }
    VL3D_REQUIRE(scratch_bytes >= vl3d_render_bwd_scratch_bytes(desc), "vl3d_render_bwd_adam: scratch smaller than vl3d_render_bwd_scratch_bytes()");
    VL3D_REQUIRE(adam->param && adam->exp_avg && adam->exp_avg_sq && adam->last_step && adam->hist && adam->step >= 1 &&
                     adam->step < (1ll << 31), "vl3d_render_bwd_adam: null pointer / bad step in the adam window");
    VL3D_REQUIRE((int64_t)desc->Hs * desc->Ws * 16 < (1ll << 32) && (int64_t)adam->Hs * adam->Ws * 16 < (1ll << 32),
                 "frame too large for 32-bit byte offsets");
    if (qk) {
        rc = check_cull(desc, qk, adam->QH, adam->QW);
        if (rc != VL3D_OK) return rc;
        VL3D_REQUIRE(adam->class_scratch, "vl3d_render_bwd_adam: a tile-culled model needs class_scratch (vl3d_render_bwd_adam_class_bytes())");
        VL3D_REQUIRE(adam->step < (1ll << 29), "vl3d_render_bwd_adam: tile-culled models keep the step in 29 bits of the texel records");
    } else {
        VL3D_REQUIRE(!adam->blocks, "vl3d_render_bwd_adam: the packed layout belongs to a tile-culled model (quad maps)");
    }
    if (qk) {
        VL3D_REQUIRE(desc->cull_Hs == adam->Hs && desc->cull_Ws == adam->Ws && desc->cull_row0 == adam->y0 && desc->cull_col0 == adam->x0,
                     "vl3d_render_bwd_adam: desc->cull_* must name the optimiser's window (y0, x0) of its (Hs, Ws) planes");
    }
    hipStream_t s = (hipStream_t)stream;
    // window / box checks and the box table on the device (the tail with a NULL gradient does exactly that)
    rc = vl3d_adam_window_step_tail(desc->D, desc->T, adam->Hs, adam->Ws, adam->y0, adam->x0, desc->Hs, desc->Ws, adam->param, nullptr,
                                    adam->exp_avg, adam->exp_avg_sq, adam->last_step, adam->hist, adam->lr, adam->beta1, adam->beta2, adam->eps,
                                    adam->step, qk, adam->quad_dyn, adam->QH, adam->QW, adam->plane_boxes, adam->blocks, nullptr, adam->boxes_scratch, s);
    if (rc != VL3D_OK) return rc;
    RenderArgs a = make_args(desc);
    if (grad_reg) set_reg_state(a, desc, reg_state);
    a.stack = (const float *)stack; a.homos = homos;
    a.rgb = const_cast<float *>(rgb); a.alpha = const_cast<float *>(alpha);
    a.g_rgb = grad_rgb; a.g_alpha = grad_alpha; a.g_reg = grad_reg; a.g_asum = grad_alpha_sums; a.g_stack = grad_stack;
    a.quad_keep = qk; a.QH = adam->QH; a.QW = adam->QW;
    set_cull_geometry(a, desc, adam->QH, adam->QW);
    a.grad_culled_unwritten = qk ? 1 : 0;
    a.plan = (const float *)scratch;
    a.owner = reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(scratch) + owner_table_off(desc));
    a.tile_rows = qk ? (bv == 3 ? 16 : 18) : 17;      // 16 / 18: the one-frame tile kernel (64- / 32-wide regions, tile-culled models), 17: the frame pairs
    const double bc1 = 1.0 - pow((double)adam->beta1, (double)adam->step), bc2 = 1.0 - pow((double)adam->beta2, (double)adam->step);
    const int ts = vl3d_adam::TS;
    a.ad.p = reinterpret_cast<float4 *>(adam->param); a.ad.m = reinterpret_cast<float4 *>(adam->exp_avg);
    a.ad.v = reinterpret_cast<float4 *>(adam->exp_avg_sq);
    a.ad.last_step = adam->last_step;
    a.ad.boxes = (adam->plane_boxes && desc->D <= 128) ? reinterpret_cast<const int4 *>(adam->boxes_scratch) : nullptr;
    a.ad.y0 = adam->y0; a.ad.x0 = adam->x0; a.ad.Hs = adam->Hs; a.ad.Ws = adam->Ws;
    a.ad.tiles_y = (adam->Hs + ts - 1) / ts; a.ad.tiles_x = (adam->Ws + ts - 1) / ts;
    a.ad.step = (int)adam->step;
    a.ad.lr_bc1 = (float)((double)adam->lr / bc1); a.ad.beta1 = adam->beta1; a.ad.beta2 = adam->beta2; a.ad.eps = adam->eps;
    a.ad.bc2s = (float)sqrt(bc2);
    a.ad.quad_dyn = qk ? adam->quad_dyn : nullptr;
    a.ad.cls = qk ? reinterpret_cast<uint2 *>(adam->class_scratch) : nullptr;
    a.ad.blocks = adam->blocks;
    rc = dispatch(true, desc, a, s);
    if (rc != VL3D_OK) return rc;
    // behind the backward, under the plan's device-side flag: nothing more (dense, feasible) / the static texels (tile-culled, feasible) / the
    // whole window from the atomics kernel's gradient (infeasible view); the tiles are marked either way
    return vl3d_adam_window_step_tail(desc->D, desc->T, adam->Hs, adam->Ws, adam->y0, adam->x0, desc->Hs, desc->Ws, adam->param, grad_stack,
                                      adam->exp_avg, adam->exp_avg_sq, adam->last_step, adam->hist, adam->lr, adam->beta1, adam->beta2, adam->eps,
                                      adam->step, qk, adam->quad_dyn, adam->QH, adam->QW, adam->plane_boxes, adam->blocks,
                                      reinterpret_cast<const int *>(scratch), nullptr, s);
}
