// Render straight from the PACKED texture of a tile-culled model (include/vl3d.h "Packed storage"; videoloop3d_amd/packed.py).
//
// The reference renders a sparsified model from its tile lists: every face of a kept quad carries UVs into the static atlas (one frame)
// or the dynamic atlas (one texture per frame), culled quads have no face at all (MPV.py:389-449).  The packed form of this package holds
// the same three kinds of texels in a pool of 8 x 8-texel blocks behind a block table [D][Hs/8][Ws/8] (-1 | slot << 1 | dynamic).  This
// kernel is the forward of MPV.py:351-454 reading THAT: per pixel and plane the base tap and the tent weights of the dense kernels
// (make_taps_i), per tap one table entry (a scalar-cache-friendly 4-byte load, shared by the 64 texels of a block) and one 16-byte texel
// from the pool -- frame t of a dynamic block, the one copy of a static block, (0, 0, 0, culled_alpha) where no block is stored --
// blended, activated and composited by the dense kernels' own shade2 / composite sequence: the image is the dense culled render's,
// bit for bit, without the (D, n, Hs, Ws, 4) stack an evaluation render used to unpack first.
// Forward only: training renders from the compact window copy the crop-aware optimiser's catch-up builds anyway (csrc/vl3d_optim.hip).
#include "vl3d_render_core.h"

using vl3d_render_detail::RenderArgs;

namespace {

constexpr int TSB = 8;      // block side (vl3d_adam_window_tile())

struct PackedSrc {
    const int *blocks;           // [D][tiles_y][tiles_x]
    const float4 *pool;
    const int *frames;           // [n] frame indices into the model's T frames
    int tiles_y, tiles_x;
    float culled_alpha;
};

__device__ __forceinline__ f4 packed_texel(const PackedSrc &p, const int *__restrict__ bplane, int x, int y, int frame) {
    const int e = bplane[(y / TSB) * p.tiles_x + x / TSB];
    if (e < 0) return f4{0.f, 0.f, 0.f, p.culled_alpha};
    const size_t o = (size_t)(e >> 1) * (TSB * TSB) + (size_t)((y % TSB) * TSB + (x % TSB)) + ((e & 1) ? (size_t)frame * (TSB * TSB) : 0);
    const float4 v = p.pool[o];
    return f4{v.x, v.y, v.z, v.w};
}

template <int RACT, int AACT>
__global__ __launch_bounds__(256) void render_fwd_packed_k(RenderArgs a, PackedSrc p, int tiles_x, int tiles_y) {
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_x = b % tiles_x, rest = b / tiles_x;
    const int tile_y = rest % tiles_y, ti = rest / tiles_y;
    const int x = tile_x * 64 + (threadIdx.x & 63);
    const int y = tile_y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const int frame = p.frames[ti];
    const float px = (float)(a.col0 + x) + a.pc, py = (float)(a.row0 + y) + a.pc;
    float Tr = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, A = 0.f;
    for (int d = 0; d < a.D; ++d) {
        float h[VL3D_HN];
        load_uniform(a.homos + VL3D_HS * d, h);
        const TapsI ti_ = make_taps_i<VL3D_COORD_AFFINE, VL3D_BORDER_HARDCUT>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, plane_cull(a, d));
        // an uncovered plane (outside the quad extent or inside a culled quad) leaves the composite state untouched bit for bit
        // (w = a T = 0: every accumulator adds +0, T is multiplied by 1): its taps are not fetched
        if (ti_.cov == 0.0f) continue;
        const int *bplane = p.blocks + (size_t)d * p.tiles_y * p.tiles_x;
        const int x1 = min(ti_.x0 + 1, a.Ws - 1), y1 = min(ti_.y0 + 1, a.Hs - 1);       // (a 1-texel axis: the second tap's weight is 0)
        f4 v[4];
        v[0] = packed_texel(p, bplane, ti_.x0, ti_.y0, frame);
        v[1] = packed_texel(p, bplane, x1, ti_.y0, frame);
        v[2] = packed_texel(p, bplane, ti_.x0, y1, frame);
        v[3] = packed_texel(p, bplane, x1, y1, frame);
        Taps2 t2;
        t2.w = ti_.w; t2.cov = ti_.cov; t2.tx = ti_.tx; t2.ty = ti_.ty; t2.off = 0;
        const f4 o = shade2<VL3D_ACT_POST, RACT, AACT>(t2, v);
        const float w = o.w * Tr;
        cr += w * o.x; cg += w * o.y; cb += w * o.z; A += w;
        Tr *= (1.0f - o.w);
    }
    const size_t pix = ((size_t)ti * a.H + y) * a.W + x;
    a.rgb[pix * 3 + 0] = cr; a.rgb[pix * 3 + 1] = cg; a.rgb[pix * 3 + 2] = cb;
    a.alpha[pix] = A;
}

template <int RACT, int AACT>
void launch_packed(const RenderArgs &a, const PackedSrc &p, int n, hipStream_t s) {
    const int tiles_x = (a.W + 63) / 64, tiles_y = (a.H + 3) / 4;
    hipLaunchKernelGGL((render_fwd_packed_k<RACT, AACT>), dim3((unsigned)(tiles_x * tiles_y * n)), dim3(256), 0, s, a, p, tiles_x, tiles_y);
}

}  // namespace

// MPV.py:351-454 for a packed tile-culled model, forward only (evaluation renders of chosen frames, MPV.py:439 `atlas_dyn[ts]`).
// desc: D, T (frames of the MODEL), Hs, Ws (texels of a plane), H, W, row0/col0, pixel_center, sx/sy/ox/oy, rgb_act/alpha_act; the planar
// MPV convention (VL3D_COORD_AFFINE, VL3D_BORDER_HARDCUT, VL3D_ACT_POST) is the only one a packed model renders with.
// blocks [D][ceil(Hs/8)][ceil(Ws/8)], pool: packed.PackedLayout; frames [n] device int32; quad_keep [D][QH][QW] (required: the block table
// is built from it); rgb (n,H,W,3), alpha (n,H,W).
extern "C" int vl3d_render_fwd_packed(const vl3d_render_desc *desc, const int32_t *blocks, const float *pool, const int32_t *frames, int32_t n,
                                      const float *homos, const uint8_t *quad_keep, int32_t QH, int32_t QW, float culled_alpha, float *rgb,
                                      float *alpha, vl3d_stream_t stream) {
    VL3D_REQUIRE(desc != nullptr, "null render desc");
    if (vl3d_check_variant(desc->variant) != VL3D_OK) return VL3D_EINVAL;
    VL3D_REQUIRE(desc->D > 0 && desc->D <= 128 && desc->T > 0 && desc->Hs > 0 && desc->Ws > 0 && desc->H > 0 && desc->W > 0 && n > 0,
                 "vl3d_render_fwd_packed: non-positive dims (or more than 128 planes)");
    VL3D_REQUIRE(blocks && pool && frames && homos && quad_keep && rgb && alpha, "vl3d_render_fwd_packed: null pointer");
    VL3D_REQUIRE((QH > 0 && QW > 0) || (QH < 0 && QW < 0 && desc->Hs % (-QH) == 0 && desc->Ws % (-QW) == 0 && desc->Hs / (-QH) >= 2 && desc->Ws / (-QW) >= 2),
                 "vl3d_render_fwd_packed: bad quad grid (tile-exact layout: whole tiles of at least 2 x 2 texels)");
    VL3D_REQUIRE(desc->coord_mode == VL3D_COORD_AFFINE && desc->border_mode == VL3D_BORDER_HARDCUT && desc->act_order == VL3D_ACT_POST,
                 "vl3d_render_fwd_packed: the planar MPV convention (affine, hardcut, post) only");
    VL3D_REQUIRE(desc->stack_dtype == VL3D_F32, "vl3d_render_fwd_packed: the pool holds fp32 texels");
    VL3D_REQUIRE((int64_t)desc->H * desc->W * n < (1ll << 40), "vl3d_render_fwd_packed: output too large");
    RenderArgs a{};
    a.D = desc->D; a.T = desc->T; a.Hs = desc->Hs; a.Ws = desc->Ws; a.H = desc->H; a.W = desc->W;
    a.row0 = desc->row0; a.col0 = desc->col0;
    a.pc = desc->pixel_center; a.sx = desc->sx; a.sy = desc->sy; a.ox = desc->ox; a.oy = desc->oy;
    a.homos = homos; a.rgb = rgb; a.alpha = alpha;
    a.quad_keep = quad_keep; a.QH = QH; a.QW = QW;
    a.q_Hs = desc->Hs; a.q_Ws = desc->Ws; a.q_x0 = 0.0f; a.q_y0 = 0.0f;
    if (QH < 0) {      // tile-exact layout (include/vl3d.h): |QH| x |QW| tiles, each quad owning its border texels
        a.QH = -QH; a.QW = -QW;
        a.q_th = desc->Hs / a.QH; a.q_tw = desc->Ws / a.QW;
        a.q_inv_cw = 1.0f / (float)(a.q_tw - 1);
        a.q_inv_ch = 1.0f / (float)(a.q_th - 1);
    } else {
        a.q_inv_cw = (float)QW / (float)(a.q_Ws > 1 ? a.q_Ws - 1 : 1);
        a.q_inv_ch = (float)QH / (float)(a.q_Hs > 1 ? a.q_Hs - 1 : 1);
    }
    PackedSrc p{blocks, reinterpret_cast<const float4 *>(pool), frames, (desc->Hs + TSB - 1) / TSB, (desc->Ws + TSB - 1) / TSB, culled_alpha};
    hipStream_t s = (hipStream_t)stream;
#define VL3D_CASE(R, A)                                                   \
    if (desc->rgb_act == R && desc->alpha_act == A) {                     \
        launch_packed<R, A>(a, p, n, s);                                  \
        VL3D_CHECK_LAUNCH();                                              \
        return VL3D_OK;                                                   \
    }
    VL3D_CASE(VL3D_ACT_SIGMOID, VL3D_ACT_SIGMOID)
    VL3D_CASE(VL3D_ACT_NONE, VL3D_ACT_NONE)
    VL3D_CASE(VL3D_ACT_NONE, VL3D_ACT_SIGMOID)
    VL3D_CASE(VL3D_ACT_CLAMP, VL3D_ACT_SIGMOID)
    VL3D_CASE(VL3D_ACT_RELU, VL3D_ACT_SIGMOID)
    VL3D_CASE(VL3D_ACT_ABS, VL3D_ACT_SIGMOID)
    VL3D_CASE(VL3D_ACT_CLAMP, VL3D_ACT_CLAMP)
    VL3D_CASE(VL3D_ACT_SIGMOID, VL3D_ACT_CLAMP)
    VL3D_CASE(VL3D_ACT_NONE, VL3D_ACT_CLAMP)
#undef VL3D_CASE
    vl3d_set_error("vl3d_render_fwd_packed: unsupported (rgb_act, alpha_act) pair");
    return VL3D_EUNSUPPORTED;
}
