// The scalar terms around the render in a training iteration, fused (include/vl3d.h "Per-pixel terms" / "Stage-1 image loss").
//
// After the fused render a stage-1 iteration (train_3d.py:189-236, MPI.py:596-652) still formed, in ~100 tiny torch launches each way,
//   sparsity  = mean_p ( sum_k a_k / max(sqrt(sum_k a_k^2), eps) )            MPI.py:599-603 / MPV.py:511-515 (from the render's alpha sums)
//   density   = mean_p | alpha_p - 1 |                                        MPI.py:647-650 / MPV.py:533-536
//   loop_loss = - mean_p ( m log l + (1 - m) log(1 - l) ),  l = clamp(label, .001, .999)        train_3d.py:200-209
//   img_loss  = mean ( (rgb * s - target)^2 ),  s = (exp(mean log((target + .01) / (rgb.detach() + .01))) + 3) / 4      train_3d.py:213-220
// -- 0.35-0.5 ms of a 2.2 ms iteration at 720p, more than the forward render.  Here each group is one pass over the pixels that forms the
// sums (double accumulators, one atomic per workgroup) AND writes the gradient for a unit upstream gradient; the autograd wrappers scale
// it by the actual upstream scalars.  All HBM-bound streaming kernels over a few MB.
#include "vl3d_common.h"

namespace {

__device__ __forceinline__ void block_sum_add(float v, double *out, float *red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (double)((red[0] + red[1]) + (red[2] + red[3])));
    __syncthreads();
}

// sums[0] += sum_p ratio_p, sums[1] += sum_p |alpha_p - 1|;  g_asum (n,2) / g_alpha (n): d(sum)/d(input) (NOT yet divided by n)
__global__ __launch_bounds__(256) void pixel_terms_k(int64_t n, const float *__restrict__ alpha, const float2 *__restrict__ asum, float eps,
                                                     double *__restrict__ sums, float2 *__restrict__ g_asum, float *__restrict__ g_alpha) {
    __shared__ float red[4];
    float sp = 0.f, dn = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (asum) {
            const float2 s = asum[i];
            // n2 = sqrt(clamp_min(sum a^2, 1e-30)) keeps the square root's gradient finite where no plane covers the pixel (MPV.sparsity_ratio)
            const bool tiny = s.y < 1e-30f;
            const float n2 = sqrtf(tiny ? 1e-30f : s.y);
            const bool floor_ = n2 < eps;
            const float den = floor_ ? eps : n2;
            sp += s.x / den;
            if (g_asum) g_asum[i] = make_float2(1.0f / den, (floor_ || tiny) ? 0.0f : -s.x / (2.0f * n2 * n2 * n2));
        }
        if (alpha) {
            const float d = alpha[i] - 1.0f;
            dn += fabsf(d);
            if (g_alpha) g_alpha[i] = d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f);
        }
    }
    if (asum) block_sum_add(sp, sums, red);
    if (alpha) block_sum_add(dn, sums + 1, red);
}

// rgbl: (B, C, h, w) addressed through strides (the NHWC render output viewed as NCHW); target (B,3,h,w) contiguous
__global__ __launch_bounds__(256) void stage1_gain_k(int B, int64_t hw, const float *__restrict__ rgbl, int64_t sb, int64_t sc, int64_t sp,
                                                     const float *__restrict__ target, double *__restrict__ log_sum) {
    __shared__ float red[4];
    float v = 0.f;
    const int64_t n = (int64_t)B * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / hw, p = i - b * hw;
        const float *r = rgbl + b * sb + p * sp, *t = target + b * 3 * hw + p;
        v += __logf((t[0] + 0.01f) / (r[0] + 0.01f)) + __logf((t[hw] + 0.01f) / (r[sc] + 0.01f)) + __logf((t[2 * hw] + 0.01f) / (r[2 * sc] + 0.01f));
    }
    block_sum_add(v, log_sum, red);
}

// sums[0] += sum (rgb s - target)^2 over B*3*hw, sums[1] += - sum (m log l + (1-m) log(1-l)) over B*hw;
// grad (B,h,w,C) contiguous: channels 0-2 = 2 (rgb s - target) s, channel 3 = -(m / l - (1 - m) / (1 - l)) inside the clamp, 0 outside (NOT yet / counts)
__global__ __launch_bounds__(256) void stage1_loss_k(int B, int C, int64_t hw, const float *__restrict__ rgbl, int64_t sb, int64_t sc, int64_t sp,
                                                     const float *__restrict__ target, const float *__restrict__ tmask, const double *__restrict__ log_sum,
                                                     double *__restrict__ sums, float *__restrict__ grad) {
    __shared__ float red[4];
    const int64_t n = (int64_t)B * hw;
    const float s = log_sum ? (__expf((float)(*log_sum / (double)(3 * n))) + 3.0f) * 0.25f : 1.0f;
    float im = 0.f, lp = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / hw, p = i - b * hw;
        const float *r = rgbl + b * sb + p * sp, *t = target + b * 3 * hw + p;
        float *g = grad + i * C;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = r[c * sc] * s - t[c * hw];
            im = fmaf(d, d, im);
            g[c] = 2.0f * d * s;
        }
        if (C == 4) {
            const float l0 = r[3 * sc], m = tmask[i];
            const float l = fminf(fmaxf(l0, 0.001f), 0.999f);
            lp -= m * __logf(l) + (1.0f - m) * __logf(1.0f - l);
            g[3] = (l0 >= 0.001f && l0 <= 0.999f) ? -(m / l - (1.0f - m) / (1.0f - l)) : 0.0f;
        }
    }
    block_sum_add(im, sums, red);
    if (C == 4) block_sum_add(lp, sums + 1, red);
}

// ---- the whole scalar head of a stage-1 iteration in ONE pass (vl3d_stage1_objective) ------------------------------------------------------
// A stage-1 iteration at the reference's crop is ~0.45 ms of GPU work; with the terms above it still went through ~45 one-element torch
// launches between the forward render and the backward render (means, weights, stack / sum, their autograd mirrors), each 10-20 us of HOST
// time: the iteration was host bound at 0.74 ms (profiles/r05d_seq_s1.txt).  Here every per-pixel term of train_3d.py:200-232 is formed in one
// sweep -- sums AND the gradients of the weighted total, final (weights and counts folded in) -- and the last workgroup to finish combines
// the sums into the total and its parts: memset + gain + this kernel on the way in, one (skipped when the upstream gradient is 1) scaling
// launch on the way back.
struct S1Obj {
    int B, has_label, has_gain;
    int64_t hw, t_sb, t_sc, t_sr, m_sb, m_sr;      // target (B,3,h,w) / target_mask (B,h,w) strides in floats (unit column stride)
    int w;
    const float *rgb, *label, *alpha, *smooth, *target, *tmask;
    const float2 *asum;
    float w_img, w_loop, w_sp, w_den, w_rs, w_as, sp_scale, eps;
    float coef[4];
    double *scratch;            // [0] log-ratio sum, [1] img, [2] loop, [3] sparsity, [4] density, [5] (as unsigned) the finished-workgroup ticket
    float *out;                 // [8]: total, img, loop, w sparsity, w density, w rgb_smooth, w a_smooth, gain
    float *g_rgb, *g_label, *g_alpha, *g_smooth;
    float2 *g_asum;
};

__global__ __launch_bounds__(256) void stage1_objective_gain_k(int64_t n, int64_t hw, int w, const float *__restrict__ rgb, const float *__restrict__ target,
                                                               int64_t t_sb, int64_t t_sc, int64_t t_sr, double *__restrict__ log_sum) {
    __shared__ float red[4];
    float v = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / hw, p = i - b * hw, y = p / w, x = p - y * w;
        const float *r = rgb + i * 3, *t = target + b * t_sb + y * t_sr + x;
        v += __logf((t[0] + 0.01f) / (r[0] + 0.01f)) + __logf((t[t_sc] + 0.01f) / (r[1] + 0.01f)) + __logf((t[2 * t_sc] + 0.01f) / (r[2] + 0.01f));
    }
    block_sum_add(v, log_sum, red);
}

__global__ __launch_bounds__(256) void stage1_objective_k(S1Obj a) {
    __shared__ float red[4];
    __shared__ unsigned last;
    const int64_t n = (int64_t)a.B * a.hw;
    const float s = a.has_gain ? (__expf((float)(a.scratch[0] / (double)(3 * n))) + 3.0f) * 0.25f : 1.0f;
    const float k_img = a.w_img / (float)(3 * n), k_loop = a.w_loop / (float)n, k_sp = a.w_sp * a.sp_scale / (float)n, k_den = a.w_den / (float)n;
    float im = 0.f, lp = 0.f, sp = 0.f, dn = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / a.hw, p = i - b * a.hw, y = p / a.w, x = p - y * a.w;
        const float *r = a.rgb + i * 3, *t = a.target + b * a.t_sb + y * a.t_sr + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = r[c] * s - t[c * a.t_sc];
            im = fmaf(d, d, im);
            a.g_rgb[i * 3 + c] = (2.0f * d * s) * k_img;
        }
        if (a.has_label) {
            const float l0 = a.label[i], m = a.tmask[b * a.m_sb + y * a.m_sr + x];
            const float l = fminf(fmaxf(l0, 0.001f), 0.999f);
            lp -= m * __logf(l) + (1.0f - m) * __logf(1.0f - l);
            a.g_label[i] = (l0 >= 0.001f && l0 <= 0.999f) ? -(m / l - (1.0f - m) / (1.0f - l)) * k_loop : 0.0f;
        }
        if (a.asum) {      // (pixel_terms_k)
            const float2 q = a.asum[i];
            const bool tiny = q.y < 1e-30f;
            const float n2 = sqrtf(tiny ? 1e-30f : q.y);
            const bool floor_ = n2 < a.eps;
            const float den = floor_ ? a.eps : n2;
            sp += q.x / den;
            a.g_asum[i] = make_float2(k_sp / den, (floor_ || tiny) ? 0.0f : (-q.x / (2.0f * n2 * n2 * n2)) * k_sp);
        }
        if (a.alpha) {
            const float d = a.alpha[i] - 1.0f;
            dn += fabsf(d);
            a.g_alpha[i] = d > 0.f ? k_den : (d < 0.f ? -k_den : 0.0f);
        }
    }
    block_sum_add(im, a.scratch + 1, red);
    if (a.has_label) block_sum_add(lp, a.scratch + 2, red);
    if (a.asum) block_sum_add(sp, a.scratch + 3, red);
    if (a.alpha) block_sum_add(dn, a.scratch + 4, red);
    // the last workgroup to get here combines the sums (its atomics and everyone else's are visible behind the fences)
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(reinterpret_cast<unsigned *>(a.scratch + 5), 1u);
    }
    __syncthreads();
    if (last != gridDim.x - 1 || threadIdx.x != 0) return;
    __threadfence();
    double sc[5];
#pragma unroll
    for (int i = 1; i < 5; ++i) sc[i] = __hip_atomic_load(a.scratch + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (other XCDs' atomics)
    const float img = (float)(sc[1] / (double)(3 * n)), loop = a.has_label ? (float)(sc[2] / (double)n) : 0.0f;
    const float spars = a.asum ? (float)(sc[3] / (double)n) * a.sp_scale * a.w_sp : 0.0f;
    const float dens = a.alpha ? (float)(sc[4] / (double)n) * a.w_den : 0.0f;
    float rs = 0.0f, as = 0.0f;
    if (a.smooth) {
        rs = (a.coef[0] * a.smooth[0] + a.coef[1] * a.smooth[1]) * a.w_rs;
        as = (a.coef[2] * a.smooth[2] + a.coef[3] * a.smooth[3]) * a.w_as;
        a.g_smooth[0] = a.coef[0] * a.w_rs; a.g_smooth[1] = a.coef[1] * a.w_rs;
        a.g_smooth[2] = a.coef[2] * a.w_as; a.g_smooth[3] = a.coef[3] * a.w_as;
    }
    a.out[0] = a.w_img * img + a.w_loop * loop + spars + dens + rs + as;
    a.out[1] = img; a.out[2] = loop; a.out[3] = spars; a.out[4] = dens; a.out[5] = rs; a.out[6] = as; a.out[7] = s;
}

// total = sum_i coef_i v_i over n <= 16 device scalars (v_0 = *main, v_1.. = rest[0..n-2]); term i belongs to group (groups >> 4 i) & 15:
// out[0] = total, out[1 + g] = the sum of group g's terms -- the weighted total of a stage-2 iteration (train_3dvid.py:230-240: swd +
// weight_k mean_k, a smoothness mean being two of the render's four sums) in one launch; its backward is coef * g in one more.
__global__ __launch_bounds__(64) void linear_head_fwd_k(int n, unsigned long long groups, int ngroups, const float *__restrict__ main_,
                                                        const float *__restrict__ rest, const float *__restrict__ coef, float *__restrict__ out) {
    if (threadIdx.x != 0) return;
    float acc[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    float tot = 0.f;
    for (int i = 0; i < n; ++i) {
        const float t = coef[i] * (i == 0 ? *main_ : rest[i - 1]);
        const int g = (int)((groups >> (4 * i)) & 15ull);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] += (k == g) ? t : 0.f;
        tot += t;
    }
    out[0] = tot;
    for (int g = 0; g < ngroups; ++g) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v = (k == g) ? acc[k] : v;
        out[1 + g] = v;
    }
}
__global__ __launch_bounds__(64) void linear_head_bwd_k(int n, const float *__restrict__ coef, const float *__restrict__ g, float *__restrict__ out) {
    if ((int)threadIdx.x < n) out[threadIdx.x] = coef[threadIdx.x] * *g;
}

unsigned grid_for(int64_t n) {
    const int64_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int vl3d_pixel_terms(int64_t n, const float *alpha, const float *alpha_sums, float eps, double *sums, float *grad_alpha_sums,
                                float *grad_alpha, vl3d_stream_t stream) {
    VL3D_REQUIRE(n > 0 && sums && (alpha || alpha_sums), "vl3d_pixel_terms: bad arguments");
    VL3D_REQUIRE((!grad_alpha || alpha) && (!grad_alpha_sums || alpha_sums), "vl3d_pixel_terms: a gradient buffer without its input");
    hipStream_t s = (hipStream_t)stream;
    VL3D_HIP(hipMemsetAsync(sums, 0, 2 * sizeof(double), s));
    hipLaunchKernelGGL(pixel_terms_k, dim3(grid_for(n)), dim3(256), 0, s, n, alpha, reinterpret_cast<const float2 *>(alpha_sums), eps, sums,
                       reinterpret_cast<float2 *>(grad_alpha_sums), grad_alpha);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_stage1_loss(int32_t B, int32_t C, int32_t h, int32_t w, const float *rgbl, int64_t sb, int64_t sc, int64_t sp, const float *target,
                                const float *target_mask, int32_t scale_invariant, double *log_sum, double *sums, float *grad, vl3d_stream_t stream) {
    VL3D_REQUIRE(B > 0 && (C == 3 || C == 4) && h > 0 && w > 0 && rgbl && target && sums && grad, "vl3d_stage1_loss: bad arguments");
    VL3D_REQUIRE(C == 3 || target_mask, "vl3d_stage1_loss: a loop-mask channel needs its target");
    VL3D_REQUIRE(!scale_invariant || log_sum, "vl3d_stage1_loss: the scale-invariant gain needs the log_sum scratch");
    hipStream_t s = (hipStream_t)stream;
    const int64_t hw = (int64_t)h * w, n = (int64_t)B * hw;
    VL3D_HIP(hipMemsetAsync(sums, 0, 2 * sizeof(double), s));
    if (scale_invariant) {
        VL3D_HIP(hipMemsetAsync(log_sum, 0, sizeof(double), s));
        hipLaunchKernelGGL(stage1_gain_k, dim3(grid_for(n)), dim3(256), 0, s, B, hw, rgbl, sb, sc, sp, target, log_sum);
    }
    hipLaunchKernelGGL(stage1_loss_k, dim3(grid_for(n)), dim3(256), 0, s, B, C, hw, rgbl, sb, sc, sp, target, target_mask,
                       scale_invariant ? log_sum : nullptr, sums, grad);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_stage1_objective(const vl3d_stage1_objective_desc *d, const float *rgb, const float *label, const float *alpha,
                                     const float *alpha_sums, const float *smooth_sums, const float *target, int64_t t_sb, int64_t t_sc, int64_t t_sr,
                                     const float *target_mask, int64_t m_sb, int64_t m_sr, double *scratch, float *out, float *grad_rgb, float *grad_label, float *grad_alpha, float *grad_alpha_sums,
                                     float *grad_smooth, vl3d_stream_t stream) {
    VL3D_REQUIRE(d && d->B > 0 && d->h > 0 && d->w > 0 && rgb && target && scratch && out && grad_rgb, "vl3d_stage1_objective: bad arguments");
    VL3D_REQUIRE(!label == !target_mask && !label == !grad_label, "vl3d_stage1_objective: the loop-mask label comes with its target and its gradient buffer");
    VL3D_REQUIRE(!alpha == !grad_alpha && !alpha_sums == !grad_alpha_sums && !smooth_sums == !grad_smooth,
                 "vl3d_stage1_objective: every optional input comes with its gradient buffer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t hw = (int64_t)d->h * d->w, n = (int64_t)d->B * hw;
    VL3D_HIP(hipMemsetAsync(scratch, 0, 6 * sizeof(double), s));
    if (d->scale_invariant)
        hipLaunchKernelGGL(stage1_objective_gain_k, dim3(grid_for(n)), dim3(256), 0, s, n, hw, d->w, rgb, target, t_sb, t_sc, t_sr, scratch);
    S1Obj a;
    a.B = d->B; a.has_label = label != nullptr; a.has_gain = d->scale_invariant != 0; a.hw = hw; a.w = d->w;
    a.t_sb = t_sb; a.t_sc = t_sc; a.t_sr = t_sr; a.m_sb = m_sb; a.m_sr = m_sr;
    a.rgb = rgb; a.label = label; a.alpha = alpha; a.smooth = smooth_sums; a.target = target; a.tmask = target_mask;
    a.asum = reinterpret_cast<const float2 *>(alpha_sums);
    a.w_img = d->w_img; a.w_loop = d->w_loop; a.w_sp = d->w_sparsity; a.w_den = d->w_density; a.w_rs = d->w_rgb_smooth; a.w_as = d->w_a_smooth;
    a.sp_scale = d->sparsity_scale; a.eps = d->eps;
    for (int i = 0; i < 4; ++i) a.coef[i] = d->smooth_coef[i];
    a.scratch = scratch; a.out = out;
    a.g_rgb = grad_rgb; a.g_label = grad_label; a.g_alpha = grad_alpha; a.g_smooth = grad_smooth;
    a.g_asum = reinterpret_cast<float2 *>(grad_alpha_sums);
    hipLaunchKernelGGL(stage1_objective_k, dim3(grid_for(n)), dim3(256), 0, s, a);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_linear_head_fwd(int32_t n, uint64_t groups, int32_t ngroups, const float *main_term, const float *rest, const float *coef, float *out,
                                    vl3d_stream_t stream) {
    VL3D_REQUIRE(n >= 1 && n <= 16 && ngroups >= 1 && ngroups <= 16 && main_term && coef && out && (n == 1 || rest), "vl3d_linear_head_fwd: bad arguments");
    hipLaunchKernelGGL(linear_head_fwd_k, dim3(1), dim3(64), 0, (hipStream_t)stream, n, (unsigned long long)groups, ngroups, main_term, rest, coef, out);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_linear_head_bwd(int32_t n, const float *coef, const float *grad_total, float *grad_terms, vl3d_stream_t stream) {
    VL3D_REQUIRE(n >= 1 && n <= 16 && coef && grad_total && grad_terms, "vl3d_linear_head_bwd: bad arguments");
    hipLaunchKernelGGL(linear_head_bwd_k, dim3(1), dim3(64), 0, (hipStream_t)stream, n, coef, grad_total, grad_terms);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}
