// The Adam arithmetic of the plane stack's optimiser (torch.optim.Adam without amsgrad / weight decay; train_3dvid.py:263-290, MPV.py:199-214),
// shared by the optimiser kernels (vl3d_optim.hip) and the render backward's fused owner store (vl3d_render_core.h, render_bwd_pair_k<ADAM>):
// ONE spelling, every product and sum an explicit fma / mul, so that a step taken inside the backward, a step taken by the step kernel and
// a deferred step replayed later give the same bits whatever code surrounds them (no contraction is left to the compiler's choice).
#pragma once
#include "vl3d_common.h"

namespace vl3d_adam {

constexpr int TS = 8;       // side of the optimiser's bookkeeping tiles in texels (vl3d_adam_window_tile)

// exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
__device__ __forceinline__ void adam_moments(float &mm, float &vv, float gg, float beta1, float beta2) {
    mm = __builtin_fmaf(beta1, mm, (1.0f - beta1) * gg);
    vv = __builtin_fmaf(beta2, vv, ((1.0f - beta2) * gg) * gg);
}

// One Adam step of one value.  The square root and the two quotients are the hardware's v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the
// correctly rounded expansions (~40 instructions per value and step): the deferred zero-gradient steps are REPLAYED through this very
// function -- on the reference's schedule a returning window replays 2-4 steps per texel, twice per iteration, and with the IEEE forms
// that arithmetic, not the 7 memory streams, set the time of the catch-up and step kernels (2.5 + 3.5 ms against 1.0 + 2.0 at depth ~1).
// Every path (dense step, window step, catch-up, flush, the render backward's fused store) goes through it, so deferring stays bit-identical
// to not deferring; against torch.optim.Adam the update term differs by <= 3 ulp per step (tests/test_gpu_optim.py: <= 2e-6 on the
// parameters after 40 steps).
__device__ __forceinline__ void adam_upd(float &pp, float gg, float &mm, float &vv, float lr_bc1, float beta1, float beta2, float eps, float bc2s) {
    adam_moments(mm, vv, gg, beta1, beta2);
    const float den = __builtin_fmaf(__builtin_amdgcn_sqrtf(vv), __builtin_amdgcn_rcpf(bc2s), eps);      // sqrt(exp_avg_sq)/sqrt(bc2) + eps
    pp = __builtin_fmaf(-lr_bc1, mm * __builtin_amdgcn_rcpf(den), pp);                                    // param.addcdiv_(exp_avg, den, value=-lr/bc1)
}

__device__ __forceinline__ void adam_upd4(float4 &pp, const float4 gg, float4 &mm, float4 &vv, float lr_bc1, float beta1, float beta2, float eps,
                                          float bc2s) {
    adam_upd(pp.x, gg.x, mm.x, vv.x, lr_bc1, beta1, beta2, eps, bc2s);
    adam_upd(pp.y, gg.y, mm.y, vv.y, lr_bc1, beta1, beta2, eps, bc2s);
    adam_upd(pp.z, gg.z, mm.z, vv.z, lr_bc1, beta1, beta2, eps, bc2s);
    adam_upd(pp.w, gg.w, mm.w, vv.w, lr_bc1, beta1, beta2, eps, bc2s);
}

// the zero-gradient steps from+1 .. upto of one texel, in registers (the dense update's operations with g = 0, in its order)
__device__ __forceinline__ void replay(float4 &pp, float4 &mm, float4 &vv, const float2 *__restrict__ hist, int from, int upto, float beta1,
                                       float beta2, float eps) {
    for (int s = from + 1; s <= upto; ++s) {
        const float2 h = hist[s];         // uniform: (lr / bc1, sqrt(bc2)) of step s
        adam_upd4(pp, make_float4(0.f, 0.f, 0.f, 0.f), mm, vv, h.x, beta1, beta2, eps, h.y);
    }
}

// ... of the two moments alone (the same operations replay() applies to them): for a caller that already holds the replayed PARAMETER --
// the compact window copy the catch-up wrote for the render is exactly that -- and needs no square root, no reciprocal and no step table
__device__ __forceinline__ void replay_moments(float4 &mm, float4 &vv, int from, int upto, float beta1, float beta2) {
    for (int s = from + 1; s <= upto; ++s) {
        adam_moments(mm.x, vv.x, 0.0f, beta1, beta2);
        adam_moments(mm.y, vv.y, 0.0f, beta1, beta2);
        adam_moments(mm.z, vv.z, 0.0f, beta1, beta2);
        adam_moments(mm.w, vv.w, 0.0f, beta1, beta2);
    }
}

// tile-culled models (quad maps keep / dyn [D][QH][QW], MPI.py:288-442): 0 = culled texel (no kept quad can read it: no parameter),
// 1 = dynamic (a parameter per frame), 2 = static (only static quads can read it: ONE parameter, living in frame 0 -- the reference's
// static atlas, MPV.py:235-288).  Same classification as tiles.quad_to_texel_mask / adam_tiles_k.  keep == NULL: everything dynamic.
// th, tw != 0: the TILE-EXACT layout (include/vl3d.h: a negative quad grid at the ABI) -- the plane is QH x QW tiles of th x tw texels, each quad owning
// its border row / column (the reference's sparsified atlases, MPI.py:380-418): a texel belongs to exactly one quad and has that quad's class.
struct Quads { const unsigned char *keep, *dyn; int QH, QW, th, tw; };
// the ABI's (QH, QW) -> Quads: a negative grid selects the tile-exact layout of a plane of Hs x Ws = |QH| th x |QW| tw texels
__host__ inline Quads make_quads(const unsigned char *keep, const unsigned char *dyn, int QH, int QW, int Hs, int Ws) {
    Quads q{keep, keep ? dyn : nullptr, QH < 0 ? -QH : QH, QW < 0 ? -QW : QW, 0, 0};
    if (keep && QH < 0 && QW < 0) { q.th = Hs / q.QH; q.tw = Ws / q.QW; }
    return q;
}
// (what a caller's quad grid must satisfy: both signs equal; tile-exact planes are whole tiles of at least 2 x 2 texels)
__host__ inline bool quad_grid_ok(int QH, int QW, int Hs, int Ws) {
    if (QH > 0 && QW > 0) return true;
    return QH < 0 && QW < 0 && Hs % (-QH) == 0 && Ws % (-QW) == 0 && Hs / (-QH) >= 2 && Ws / (-QW) >= 2;
}
__device__ __forceinline__ int texel_class(const Quads &q, int d, int x, int y, int Hs, int Ws) {
    if (!q.keep) return 1;
    if (q.th) {      // uniform: tile-exact layout, one quad per texel ((i + 1/2) / t in fp32 is exact to the tile for i < 2^22)
        const int qy = min((int)(((float)y + 0.5f) * (1.0f / (float)q.th)), q.QH - 1), qx = min((int)(((float)x + 0.5f) * (1.0f / (float)q.tw)), q.QW - 1);
        const size_t i = ((size_t)d * q.QH + qy) * q.QW + qx;
        if (!q.keep[i]) return 0;
        return (!q.dyn || q.dyn[i]) ? 1 : 2;
    }
    const int ylo = quad_index(y - 1, Hs, q.QH), yhi = quad_index(y + 1, Hs, q.QH), xlo = quad_index(x - 1, Ws, q.QW), xhi = quad_index(x + 1, Ws, q.QW);
    const unsigned char *k = q.keep + (size_t)d * q.QH * q.QW;
    if (!(k[ylo * q.QW + xlo] | k[ylo * q.QW + xhi] | k[yhi * q.QW + xlo] | k[yhi * q.QW + xhi])) return 0;
    if (!q.dyn) return 1;
    const unsigned char *m = q.dyn + (size_t)d * q.QH * q.QW;
    return (m[ylo * q.QW + xlo] | m[ylo * q.QW + xhi] | m[yhi * q.QW + xlo] | m[yhi * q.QW + xhi]) ? 1 : 2;
}

}  // namespace vl3d_adam

// ---- the render backward's fused owner store (vl3d_render_bwd_adam) ----------------------------------------------------------------
// What the owner-computes backward needs to apply the step where it would have stored the gradient: the (D,T,Hs,Ws,4) parameter / moment
// tensors of the full plane stack, the bookkeeping of the crop-aware optimiser (per-tile step table) and the scalars of THIS step.  The
// stack the backward renders from is the compact copy of the texel window at (y0, x0), holding the parameters current for step - 1.
struct vl3d_adam_epilogue {
    float4 *p, *m, *v;             // NULL p: no fused step (the backward stores the gradient)
    const int *last_step;          // [D][tiles_y][tiles_x]
    const int4 *boxes;             // device [D] x (y0, y1, x0, x1) in plane texels (a plane's texels outside its box stay deferred) or NULL
    int y0, x0, Hs, Ws, tiles_y, tiles_x, step;
    float lr_bc1, beta1, beta2, eps, bc2s;
    // tile-culled models: the quad maps (classification of a texel: vl3d_adam::texel_class) and an 8-byte record per texel of the compact window
    // [D][desc->Hs][desc->Ws] (+ the owner table's padding), written by the backward's pre-pass (frame independent) and read by the gather next
    // to its owner entry: .x = class | step << 2 -- class 0 culled / 1 dynamic (stepped in the owner's store) / 2 static (its gradient is
    // stored: the step kernel sums it over the frames) / 3 outside its plane's box; step = what the texel's bookkeeping tile is current for --,
    // .y = the texel's 16-byte slot inside a frame of p / m / v (dense tensors or packed pools; dynamic texels only)
    const unsigned char *quad_dyn;
    uint2 *cls;                    // NULL: dense model (every texel dynamic)
    // PACKED storage (vl3d_adam_window_step_boxes' `blocks`): p / m / v are pools of 8 x 8-texel blocks, blocks [D][tiles_y][tiles_x] = -1 or
    // slot << 1 | dynamic; a dynamic block owns T consecutive slots (frame-major).  NULL: the dense (D,T,Hs,Ws,4) tensors.
    const int *blocks;
};

// (internal, vl3d_optim.hip) the tail of vl3d_render_bwd_adam: the window step from the compact gradient for what the backward did not step
// itself -- static texels of a tile-culled model always; everything when the device-side plan of the backward said "infeasible"
// (*plan_ok == 0: the atomics kernel produced the gradient) -- then the tile marks.  Called BEFORE the render kernels with
// grad_compact == NULL: window / box checks and the box table onto the device (boxes_dev).
__attribute__((visibility("hidden"))) int vl3d_adam_window_step_tail(int32_t D, int32_t T, int32_t Hs, int32_t Ws, int32_t y0, int32_t x0, int32_t wh,
                                                                     int32_t ww, float *param, const float *grad_compact, float *exp_avg,
                                                                     float *exp_avg_sq, int32_t *last_step, const float *hist, float lr, float beta1,
                                                                     float beta2, float eps, int64_t step, const uint8_t *quad_keep,
                                                                     const uint8_t *quad_dyn, int32_t QH, int32_t QW, const int32_t *plane_boxes,
                                                                     const int32_t *blocks, const int *plan_ok, void *boxes_dev, hipStream_t stream);
