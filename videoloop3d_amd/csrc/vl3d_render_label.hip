// Stage 1's composited loop-mask label when add_uv_noise is on (include/vl3d.h vl3d_label_noise_fwd / _bwd).
//
// MPI.py:519-522 jitters the COLOUR samples' UVs by half a texel while training; MPI.py:568-583 then samples the loop-mask texture at the
// UNJITTERED UVs (`uvs`, not the local `uvs_`) and composites sigmoid(mask) with the DETACHED alphas of the jittered colour samples:
//     label = sum_k w_k sigmoid(sample(mask_k, uv))          w_k = a_k T_k,  a_k = act(sample(alpha_k, uv + jitter_k)) x coverage(uv)
// -- two sampling positions per layer.  The fused label channel of the render (vl3d_render_fwd_mask) and the label pass over a (mask, ., ., alpha)
// buffer have ONE, so this combination gets a kernel pair of its own: per pixel and plane the alpha taps at the jittered position (the field of
// desc->uv_noise_seed, the one the colour pass of the same call draws: uv_jitter), the mask taps at the plain one, coverage at the plain one
// (the rasteriser's, in the reference).  The gradient reaches the mask texture only (alphas detached, MPI.py:577-579): global atomics at the four
// unjittered taps.  Off in every shipped configuration (config_parser.py:48): built for completeness of MPI.py's semantics, not for speed.
#include <string>
#include "vl3d_render_core.h"

using vl3d_render_detail::RenderArgs;

namespace {

__device__ __forceinline__ float act_any(int act, float v) {
    switch (act) {
        case VL3D_ACT_SIGMOID: return act_fwd<VL3D_ACT_SIGMOID>(v);
        case VL3D_ACT_RELU: return act_fwd<VL3D_ACT_RELU>(v);
        case VL3D_ACT_CLAMP: return act_fwd<VL3D_ACT_CLAMP>(v);
        case VL3D_ACT_ABS: return act_fwd<VL3D_ACT_ABS>(v);
        default: return v;
    }
}

// BWD = false: label (T,H,W).  BWD = true: g_mask (D,T,Hs,Ws) += g_label w_k sigmoid'(m_k) x tap weights (zero-filled by the caller of the kernel).
template <bool BWD>
__global__ __launch_bounds__(256) void label_noise_k(RenderArgs a, int alpha_act) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), t = blockIdx.z;
    if (x >= a.W || y >= a.H) return;
    const float px = (float)(a.col0 + x) + a.pc, py = (float)(a.row0 + y) + a.pc;
    const size_t frame = (size_t)a.Hs * a.Ws, pix = ((size_t)t * a.H + y) * a.W + x;
    const TapStep st = make_tap_step<false>(a.Hs, a.Ws);
    const float gL = BWD ? a.g_label[pix] : 0.0f;
    float Tr = 1.0f, lab = 0.0f;
    for (int d = 0; d < a.D; ++d) {
        float h[VL3D_HN];
        load_uniform(a.homos + VL3D_HS * d, h);
        const QuadCull noq = QuadCull{nullptr, 0, 0, 0.f, 0.f, 0.f, 0.f, 0};
        const Taps2 tu = make_taps2<VL3D_COORD_AFFINE, VL3D_BORDER_HARDCUT>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, noq, UvNoise{0u, 0});
        if (tu.cov == 0.0f) continue;      // (the plane does not cover the pixel: a = 0, the composite does not move)
        const Taps2 tj = make_taps2<VL3D_COORD_AFFINE, VL3D_BORDER_HARDCUT>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, noq, UvNoise{a.uv_seed, d});
        const char *plane = reinterpret_cast<const char *>(a.stack) + ((size_t)d * a.T + t) * frame * 16;
        f4 v[4];
        load_taps2<false>(plane, tj, st, v);
        // (associated like shade2's blend: the alpha the colour pass of the same call composites with)
        const float a_pre = v[0].w * tj.w[0] + (v[1].w * tj.w[1] + (v[2].w * tj.w[2] + v[3].w * tj.w[3]));
        const float al = act_any(alpha_act, a_pre);
        float mt[4];
        const float *mplane = a.mask + ((size_t)d * a.T + t) * frame;
        load_mask_taps(mplane, tu.off, st, mt);
        const float m = act_fwd<VL3D_ACT_SIGMOID>(mask_blend(mt, tu.w));
        const float w = al * Tr;
        if constexpr (BWD) {
            const float gm = gL * w * (m * (1.0f - m));
            float *gmp = a.g_mask + ((size_t)d * a.T + t) * frame + (tu.off >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tu.w[i] != 0.0f) atomicAdd(gmp + ((i & 1) ? (st.dx >> 4) : 0u) + ((i & 2) ? (st.dy >> 4) : 0u), gm * tu.w[i]);
        } else {
            lab = fmaf(w, m, lab);
        }
        Tr *= (1.0f - al);
    }
    if constexpr (!BWD) a.label[pix] = lab;
}

int check(const vl3d_render_desc *d, const char *who) {
    VL3D_REQUIRE(d != nullptr, "null render desc");
    if (vl3d_check_variant(d->variant) != VL3D_OK) return VL3D_EINVAL;
    VL3D_REQUIRE(d->D > 0 && d->T > 0 && d->Hs > 0 && d->Ws > 0 && d->H > 0 && d->W > 0 && d->T <= 65535, "non-positive render dims");
    VL3D_REQUIRE((int64_t)d->Hs * d->Ws * 16 < (1ll << 32), "frame too large for 32-bit byte offsets");
    if (!(d->coord_mode == VL3D_COORD_AFFINE && d->border_mode == VL3D_BORDER_HARDCUT && d->act_order == VL3D_ACT_POST && d->stack_dtype == VL3D_F32)) {
        vl3d_set_error((std::string(who) + ": the planar MPI convention only -- (affine, hardcut, post), fp32 stack (MPI.py:452-594)").c_str());
        return VL3D_EUNSUPPORTED;
    }
    return VL3D_OK;
}

RenderArgs args_of(const vl3d_render_desc *d) {
    RenderArgs a{};
    a.D = d->D; a.T = d->T; a.Hs = d->Hs; a.Ws = d->Ws; a.H = d->H; a.W = d->W; a.Tstride = d->T;
    a.row0 = d->row0; a.col0 = d->col0;
    a.pc = d->pixel_center; a.sx = d->sx; a.sy = d->sy; a.ox = d->ox; a.oy = d->oy;
    a.uv_seed = d->uv_noise_seed;
    return a;
}

}  // namespace

extern "C" int vl3d_label_noise_fwd(const vl3d_render_desc *desc, const float *stack, const float *mask, const float *homos, float *label,
                                    vl3d_stream_t stream) {
    int rc = check(desc, "vl3d_label_noise_fwd");
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(stack && mask && homos && label, "null pointer passed to vl3d_label_noise_fwd");
    RenderArgs a = args_of(desc);
    a.stack = stack; a.mask = mask; a.homos = homos; a.label = label;
    hipLaunchKernelGGL(label_noise_k<false>, dim3((desc->W + 63) / 64, (desc->H + 3) / 4, desc->T), dim3(256), 0, (hipStream_t)stream, a, desc->alpha_act);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_label_noise_bwd(const vl3d_render_desc *desc, const float *stack, const float *mask, const float *homos, const float *grad_label,
                                    float *grad_mask, vl3d_stream_t stream) {
    int rc = check(desc, "vl3d_label_noise_bwd");
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(stack && mask && homos && grad_label && grad_mask, "null pointer passed to vl3d_label_noise_bwd");
    RenderArgs a = args_of(desc);
    a.stack = stack; a.mask = mask; a.homos = homos; a.g_label = grad_label; a.g_mask = grad_mask;
    VL3D_HIP(hipMemsetAsync(grad_mask, 0, (size_t)desc->D * desc->T * desc->Hs * desc->Ws * sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(label_noise_k<true>, dim3((desc->W + 63) / 64, (desc->H + 3) / 4, desc->T), dim3(256), 0, (hipStream_t)stream, a, desc->alpha_act);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}
