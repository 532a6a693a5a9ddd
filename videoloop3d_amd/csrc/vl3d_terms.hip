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
