// Fused MPI/MPV render for gfx950: per-plane homography warp + bilinear sample + activation +
// front-to-back over-composite across D planes, forward and backward.
//
// Replaces (reference, /root/reference): MPV.py:351-454 (planar geometry) ==
// utils_mpi.py:159-176 (warp_homography) + utils_mpi.py:92-107 (overcompose), and their autograd.
//
// Data layout in HBM: plane stack (D,T,Hs,Ws,4) fp32 -- one texel = one 16-byte rgba vector, so a
// wave of 64 consecutive output pixels reads ~65 consecutive texels (1 KiB, fully coalesced) per tap
// row.  Output rgb (T,H,W,3), alpha (T,H,W).
//
// Variant 0 (this file, v1): one thread per output pixel, planes walked front-to-back in registers,
// taps fetched straight through the vector L1 (texture-cache style; the 4x tap redundancy between
// neighbouring pixels is absorbed by L1/L2).  Backward is a single front-to-back sweep that uses the
// saved forward outputs:  sum_{j>k} w_j q_j = (G.C + gA.A) - sum_{j<=k} w_j q_j  (SURVEY §9.3), and
// scatters into grad_stack with hardware fp32 atomics.
#include "vl3d_common.h"

namespace {

struct RenderArgs {
    const float *stack;
    const float *homos;
    float *rgb;
    float *alpha;
    const float *g_rgb;
    const float *g_alpha;
    float *g_stack;
    int D, T, Hs, Ws, H, W, row0, col0;
    float pc, sx, sy, ox, oy;
};

struct Taps {
    int idx[4];     // texel index (y*Ws+x) of each tap, -1 if out of range
    float w[4];     // bilinear weights
    int x0, y0;     // top-left tap
    bool covered;   // plane contributes at this pixel
};

template <int COORD, int BORDER>
__device__ __forceinline__ Taps make_taps(const float *__restrict__ h, float px, float py, int Hs, int Ws,
                                          float sx, float sy, float ox, float oy) {
    Taps t;
    // p = H (x, y, 1), perspective divide (utils_mpi.py:171-172)
    float X = h[0] * px + h[1] * py + h[2];
    float Y = h[3] * px + h[4] * py + h[5];
    float Z = h[6] * px + h[7] * py + h[8];
    float xs = X / Z, ys = Y / Z;
    float tx = texel_coord<COORD>(xs, (float)Ws / 2.0f, (float)(Ws - 1), sx, ox);
    float ty = texel_coord<COORD>(ys, (float)Hs / 2.0f, (float)(Hs - 1), sy, oy);
    bool cov;
    if constexpr (BORDER == VL3D_BORDER_HARDCUT)
        cov = (tx >= 0.0f) && (tx <= (float)(Ws - 1)) && (ty >= 0.0f) && (ty <= (float)(Hs - 1));
    else
        cov = (tx > -1.0f) && (tx < (float)Ws) && (ty > -1.0f) && (ty < (float)Hs);  // some tap in range
    t.covered = cov;   // NaN/inf coordinates compare false -> uncovered
    if (!cov) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { t.idx[i] = -1; t.w[i] = 0.0f; }
        t.x0 = t.y0 = 0;
        return t;
    }
    float fx0 = floorf(tx), fy0 = floorf(ty);
    float fx = tx - fx0, fy = ty - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    t.x0 = x0; t.y0 = y0;
    bool xl = x0 >= 0, xr = x0 + 1 < Ws, yt = y0 >= 0, yb = y0 + 1 < Hs;
    int base = y0 * Ws + x0;
    t.idx[0] = (xl && yt) ? base : -1;
    t.idx[1] = (xr && yt) ? base + 1 : -1;
    t.idx[2] = (xl && yb) ? base + Ws : -1;
    t.idx[3] = (xr && yb) ? base + Ws + 1 : -1;
    t.w[0] = (1.0f - fx) * (1.0f - fy);
    t.w[1] = fx * (1.0f - fy);
    t.w[2] = (1.0f - fx) * fy;
    t.w[3] = fx * fy;
    return t;
}

__device__ __forceinline__ float4 ld_texel(const float *plane, int idx) {
    return idx >= 0 ? *reinterpret_cast<const float4 *>(plane + (size_t)idx * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// sample + activate one plane at one pixel -> c (rgb), a.  Keeps what backward needs.
template <int ORDER, int RACT, int AACT>
__device__ __forceinline__ void shade(const float *plane, const Taps &t, float4 &out, float4 &pre, float4 tapv[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) tapv[i] = ld_texel(plane, t.idx[i]);
    if constexpr (ORDER == VL3D_ACT_POST) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v.x += t.w[i] * tapv[i].x; v.y += t.w[i] * tapv[i].y;
            v.z += t.w[i] * tapv[i].z; v.w += t.w[i] * tapv[i].w;
        }
        pre = v;
        out = make_float4(act_fwd<RACT>(v.x), act_fwd<RACT>(v.y), act_fwd<RACT>(v.z), act_fwd<AACT>(v.w));
    } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (t.idx[i] >= 0) {   // out-of-range taps contribute 0 (not act(0)): zeros padding of the ACTIVATED image
                v.x += t.w[i] * act_fwd<RACT>(tapv[i].x); v.y += t.w[i] * act_fwd<RACT>(tapv[i].y);
                v.z += t.w[i] * act_fwd<RACT>(tapv[i].z); v.w += t.w[i] * act_fwd<AACT>(tapv[i].w);
            }
        }
        pre = v;
        out = v;
    }
}

constexpr int TILE_X = 64, TILE_Y = 4;

template <int COORD, int BORDER, int ORDER, int RACT, int AACT>
__global__ __launch_bounds__(TILE_X *TILE_Y) void render_fwd_k(RenderArgs a) {
    const int x = blockIdx.x * TILE_X + (threadIdx.x & (TILE_X - 1));
    const int y = blockIdx.y * TILE_Y + (threadIdx.x / TILE_X);
    const int t = blockIdx.z;
    if (x >= a.W || y >= a.H) return;
    const float px = (float)(a.col0 + x) + a.pc, py = (float)(a.row0 + y) + a.pc;
    const size_t frame = (size_t)a.Hs * a.Ws * 4;
    const float *plane = a.stack + (size_t)t * frame;
    const size_t plane_stride = (size_t)a.T * frame;
    float Tr = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, A = 0.f;
    for (int d = 0; d < a.D; ++d, plane += plane_stride) {
        Taps tp = make_taps<COORD, BORDER>(a.homos + 9 * d, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy);
        if (!tp.covered) continue;
        float4 o, pre, tv[4];
        shade<ORDER, RACT, AACT>(plane, tp, o, pre, tv);
        float w = o.w * Tr;             // blend weight a_k * T_k (utils_mpi.py:100-104)
        cr += w * o.x; cg += w * o.y; cb += w * o.z; A += w;
        Tr *= (1.0f - o.w);
    }
    const size_t pix = ((size_t)t * a.H + y) * a.W + x;
    a.rgb[pix * 3 + 0] = cr; a.rgb[pix * 3 + 1] = cg; a.rgb[pix * 3 + 2] = cb;
    a.alpha[pix] = A;
}

template <int COORD, int BORDER, int ORDER, int RACT, int AACT>
__global__ __launch_bounds__(TILE_X *TILE_Y) void render_bwd_k(RenderArgs a) {
    const int x = blockIdx.x * TILE_X + (threadIdx.x & (TILE_X - 1));
    const int y = blockIdx.y * TILE_Y + (threadIdx.x / TILE_X);
    const int t = blockIdx.z;
    if (x >= a.W || y >= a.H) return;
    const float px = (float)(a.col0 + x) + a.pc, py = (float)(a.row0 + y) + a.pc;
    const size_t frame = (size_t)a.Hs * a.Ws * 4;
    const size_t plane_stride = (size_t)a.T * frame;
    const float *plane = a.stack + (size_t)t * frame;
    float *gplane = a.g_stack + (size_t)t * frame;
    const size_t pix = ((size_t)t * a.H + y) * a.W + x;
    const float Gr = a.g_rgb[pix * 3 + 0], Gg = a.g_rgb[pix * 3 + 1], Gb = a.g_rgb[pix * 3 + 2];
    const float gA = a.g_alpha ? a.g_alpha[pix] : 0.0f;
    // S = sum_k w_k q_k with q_k = G.c_k + gA  ==  G.C + gA*A from the saved forward outputs
    const float S = Gr * a.rgb[pix * 3 + 0] + Gg * a.rgb[pix * 3 + 1] + Gb * a.rgb[pix * 3 + 2] + gA * a.alpha[pix];
    float Tr = 1.0f, P = 0.0f;
    for (int d = 0; d < a.D; ++d, plane += plane_stride, gplane += plane_stride) {
        Taps tp = make_taps<COORD, BORDER>(a.homos + 9 * d, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy);
        if (!tp.covered) continue;
        float4 o, pre, tv[4];
        shade<ORDER, RACT, AACT>(plane, tp, o, pre, tv);
        const float q = Gr * o.x + Gg * o.y + Gb * o.z + gA;
        const float w = o.w * Tr;
        P += w * q;
        const float om = 1.0f - o.w;
        // dL/da_k = T_k q_k - (sum_{j>k} w_j q_j)/(1-a_k); everything behind a fully opaque plane has zero weight
        const float behind = (om > 1e-12f) ? (S - P) / om : 0.0f;
        float4 go = make_float4(w * Gr, w * Gg, w * Gb, Tr * q - behind);   // grad wrt activated (c, a)
        Tr *= om;
        if constexpr (ORDER == VL3D_ACT_POST) {
            float4 gv = make_float4(go.x * act_bwd<RACT>(pre.x, o.x), go.y * act_bwd<RACT>(pre.y, o.y),
                                    go.z * act_bwd<RACT>(pre.z, o.z), go.w * act_bwd<AACT>(pre.w, o.w));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (tp.idx[i] >= 0) {
                    float *g = gplane + (size_t)tp.idx[i] * 4;
                    atomicAdd(g + 0, tp.w[i] * gv.x); atomicAdd(g + 1, tp.w[i] * gv.y);
                    atomicAdd(g + 2, tp.w[i] * gv.z); atomicAdd(g + 3, tp.w[i] * gv.w);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (tp.idx[i] >= 0) {
                    float *g = gplane + (size_t)tp.idx[i] * 4;
                    const float4 s = tv[i];
                    atomicAdd(g + 0, tp.w[i] * go.x * act_bwd<RACT>(s.x, act_fwd<RACT>(s.x)));
                    atomicAdd(g + 1, tp.w[i] * go.y * act_bwd<RACT>(s.y, act_fwd<RACT>(s.y)));
                    atomicAdd(g + 2, tp.w[i] * go.z * act_bwd<RACT>(s.z, act_fwd<RACT>(s.z)));
                    atomicAdd(g + 3, tp.w[i] * go.w * act_bwd<AACT>(s.w, act_fwd<AACT>(s.w)));
                }
            }
        }
    }
}

// ---- dispatch over the compile-time conventions -------------------------------------------------------
template <bool BWD, int COORD, int BORDER, int ORDER, int RACT, int AACT>
void launch(const RenderArgs &a, hipStream_t s) {
    dim3 grid((a.W + TILE_X - 1) / TILE_X, (a.H + TILE_Y - 1) / TILE_Y, a.T), block(TILE_X * TILE_Y);
    if constexpr (BWD)
        hipLaunchKernelGGL((render_bwd_k<COORD, BORDER, ORDER, RACT, AACT>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((render_fwd_k<COORD, BORDER, ORDER, RACT, AACT>), grid, block, 0, s, a);
}

template <bool BWD, int COORD, int BORDER, int ORDER>
int dispatch_act(const vl3d_render_desc *d, const RenderArgs &a, hipStream_t s) {
    // activation pairs kept by the product: (sigmoid,sigmoid) shipped; (none,none) for pre-activated stacks;
    // (sigmoid|none|clamp|relu|abs) x (sigmoid|none|clamp) otherwise
#define VL3D_CASE(R, A)                                              \
    if (d->rgb_act == R && d->alpha_act == A) {                      \
        launch<BWD, COORD, BORDER, ORDER, R, A>(a, s);               \
        return VL3D_OK;                                              \
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
    vl3d_set_error("unsupported (rgb_act, alpha_act) pair");
    return VL3D_EUNSUPPORTED;
}

template <bool BWD>
int dispatch(const vl3d_render_desc *d, const RenderArgs &a, hipStream_t s) {
#define VL3D_CASE(C, B, O)                                                         \
    if (d->coord_mode == C && d->border_mode == B && d->act_order == O)            \
        return dispatch_act<BWD, C, B, O>(d, a, s);
    VL3D_CASE(VL3D_COORD_UTILS_MPI, VL3D_BORDER_ZEROS, VL3D_ACT_PRE)
    VL3D_CASE(VL3D_COORD_UTILS_MPI, VL3D_BORDER_ZEROS, VL3D_ACT_POST)
    VL3D_CASE(VL3D_COORD_UTILS_MPI, VL3D_BORDER_HARDCUT, VL3D_ACT_PRE)
    VL3D_CASE(VL3D_COORD_UTILS_MPI, VL3D_BORDER_HARDCUT, VL3D_ACT_POST)
    VL3D_CASE(VL3D_COORD_AFFINE, VL3D_BORDER_ZEROS, VL3D_ACT_PRE)
    VL3D_CASE(VL3D_COORD_AFFINE, VL3D_BORDER_ZEROS, VL3D_ACT_POST)
    VL3D_CASE(VL3D_COORD_AFFINE, VL3D_BORDER_HARDCUT, VL3D_ACT_PRE)
    VL3D_CASE(VL3D_COORD_AFFINE, VL3D_BORDER_HARDCUT, VL3D_ACT_POST)
#undef VL3D_CASE
    vl3d_set_error("unsupported (coord_mode, border_mode, act_order)");
    return VL3D_EUNSUPPORTED;
}

int check_desc(const vl3d_render_desc *d) {
    VL3D_REQUIRE(d != nullptr, "null render desc");
    VL3D_REQUIRE(d->D > 0 && d->T > 0 && d->Hs > 0 && d->Ws > 0 && d->H > 0 && d->W > 0, "non-positive render dims");
    VL3D_REQUIRE((int64_t)d->Hs * d->Ws < (1ll << 31), "plane too large for 32-bit texel index");
    VL3D_REQUIRE(d->stack_dtype == VL3D_F32, "only fp32 plane stacks are implemented in this round");
    return VL3D_OK;
}

RenderArgs make_args(const vl3d_render_desc *d) {
    RenderArgs a{};
    a.D = d->D; a.T = d->T; a.Hs = d->Hs; a.Ws = d->Ws; a.H = d->H; a.W = d->W;
    a.row0 = d->row0; a.col0 = d->col0;
    a.pc = d->pixel_center; a.sx = d->sx; a.sy = d->sy; a.ox = d->ox; a.oy = d->oy;
    return a;
}

}  // namespace

extern "C" int vl3d_render_fwd(const vl3d_render_desc *desc, const void *stack, const float *homos,
                               float *rgb, float *alpha, vl3d_stream_t stream) {
    int rc = check_desc(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(stack && homos && rgb && alpha, "null pointer passed to vl3d_render_fwd");
    RenderArgs a = make_args(desc);
    a.stack = (const float *)stack; a.homos = homos; a.rgb = rgb; a.alpha = alpha;
    rc = dispatch<false>(desc, a, (hipStream_t)stream);
    if (rc != VL3D_OK) return rc;
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_render_bwd(const vl3d_render_desc *desc, const void *stack, const float *homos,
                               const float *rgb, const float *alpha, const float *grad_rgb,
                               const float *grad_alpha, float *grad_stack, vl3d_stream_t stream) {
    int rc = check_desc(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(stack && homos && rgb && alpha && grad_rgb && grad_stack, "null pointer passed to vl3d_render_bwd");
    RenderArgs a = make_args(desc);
    a.stack = (const float *)stack; a.homos = homos;
    a.rgb = const_cast<float *>(rgb); a.alpha = const_cast<float *>(alpha);
    a.g_rgb = grad_rgb; a.g_alpha = grad_alpha; a.g_stack = grad_stack;
    const size_t bytes = (size_t)desc->D * desc->T * desc->Hs * desc->Ws * 4 * sizeof(float);
    VL3D_HIP(hipMemsetAsync(grad_stack, 0, bytes, (hipStream_t)stream));
    rc = dispatch<true>(desc, a, (hipStream_t)stream);
    if (rc != VL3D_OK) return rc;
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}
