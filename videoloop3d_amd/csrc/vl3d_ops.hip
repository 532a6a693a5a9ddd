// Unfused drop-in operators of the reference's utils_mpi.py for gfx950:
//   warp_homography (utils_mpi.py:159-176), overcompose (:92-107), overcomposeNto0 (:110-132)
// plus library-wide error state.  These exist for API parity with the reference's L3 functions; the
// training/render hot path uses the fused kernel in vl3d_render.hip.
#include <string.h>

#include "vl3d_adam.h"

static thread_local char g_err[256] = "";

extern "C" void vl3d_set_error(const char *msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char *vl3d_last_error(void) { return g_err; }
extern "C" int vl3d_version(void) { return 100; }

namespace {

// ---------------------------------------------------------------------------------------------------
// warp_homography: images [N,C,Hs,Ws] channel-planar (the reference's layout), out [N,C,h,w].
struct WarpTap {
    int idx[4];
    float w[4];
};

__device__ __forceinline__ WarpTap warp_taps(const float *__restrict__ hm, int x, int y, int Hs, int Ws) {
    WarpTap t;
    const float px = (float)x, py = (float)y;
    float X = hm[0] * px + hm[1] * py + hm[2];
    float Y = hm[3] * px + hm[4] * py + hm[5];
    float Z = hm[6] * px + hm[7] * py + hm[8];
    float tx = texel_coord<VL3D_COORD_UTILS_MPI>(X / Z, (float)Ws / 2.0f, (float)(Ws - 1), 0.f, 0.f);
    float ty = texel_coord<VL3D_COORD_UTILS_MPI>(Y / Z, (float)Hs / 2.0f, (float)(Hs - 1), 0.f, 0.f);
    bool in = (tx > -1.0f) && (tx < (float)Ws) && (ty > -1.0f) && (ty < (float)Hs);
    if (!in) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { t.idx[i] = -1; t.w[i] = 0.f; }
        return t;
    }
    float fx0 = floorf(tx), fy0 = floorf(ty);
    float fx = tx - fx0, fy = ty - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    bool xl = x0 >= 0, xr = x0 + 1 < Ws, yt = y0 >= 0, yb = y0 + 1 < Hs;
    int base = y0 * Ws + x0;
    t.idx[0] = (xl && yt) ? base : -1;
    t.idx[1] = (xr && yt) ? base + 1 : -1;
    t.idx[2] = (xl && yb) ? base + Ws : -1;
    t.idx[3] = (xr && yb) ? base + Ws + 1 : -1;
    t.w[0] = (1.f - fx) * (1.f - fy); t.w[1] = fx * (1.f - fy);
    t.w[2] = (1.f - fx) * fy;         t.w[3] = fx * fy;
    return t;
}

__global__ __launch_bounds__(256) void warp_fwd_k(int C, int Hs, int Ws, int h, int w, const float *__restrict__ homos,
                                                  const float *__restrict__ images, float *__restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (x >= w || y >= h) return;
    WarpTap t = warp_taps(homos + 9 * n, x, y, Hs, Ws);
    const size_t splane = (size_t)Hs * Ws, oplane = (size_t)h * w;
    const float *img = images + (size_t)n * C * splane;
    float *o = out + (size_t)n * C * oplane + (size_t)y * w + x;
    for (int c = 0; c < C; ++c, img += splane, o += oplane) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (t.idx[i] >= 0) v += t.w[i] * img[t.idx[i]];
        *o = v;
    }
}

__global__ __launch_bounds__(256) void warp_bwd_k(int C, int Hs, int Ws, int h, int w, const float *__restrict__ homos,
                                                  const float *__restrict__ gout, float *__restrict__ gimg) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (x >= w || y >= h) return;
    WarpTap t = warp_taps(homos + 9 * n, x, y, Hs, Ws);
    const size_t splane = (size_t)Hs * Ws, oplane = (size_t)h * w;
    float *gi = gimg + (size_t)n * C * splane;
    const float *go = gout + (size_t)n * C * oplane + (size_t)y * w + x;
    for (int c = 0; c < C; ++c, gi += splane, go += oplane) {
        const float g = *go;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (t.idx[i] >= 0) atomicAdd(gi + t.idx[i], t.w[i] * g);
    }
}

// ---------------------------------------------------------------------------------------------------
// overcompose: alpha [P,D], content [P,D,C]; front = index 0 (utils_mpi.py:92-107)
__global__ __launch_bounds__(256) void overcompose_fwd_k(int64_t P, int D, int C, const float *__restrict__ alpha,
                                                         const float *__restrict__ content, float *__restrict__ rgb,
                                                         float *__restrict__ bw) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float *a = alpha + p * D;
    const float *ct = content + p * D * C;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    float Tr = 1.f;
    for (int d = 0; d < D; ++d) {
        const float ad = a[d];
        const float w = ad * Tr;
        bw[p * D + d] = w;
        for (int c = 0; c < C; ++c) acc[c] += ct[d * C + c] * w;
        Tr *= (1.f - ad);
    }
    for (int c = 0; c < C; ++c) rgb[p * C + c] = acc[c];
}

// Backward without any division: back-to-front recurrence R_{k-1} = a_k q'_k + (1-a_k) R_k is not possible
// with a per-plane upstream gradient on the blend weights, so use the generic two-sweep form:
//   w_k = a_k T_k ; given gw_k := dL/dw_k (= g_rgb . c_k + g_bw_k):
//   dL/da_k = T_k gw_k - sum_{j>k} gw_j w_j / (1 - a_k)        (exact 0/0 -> handled by direct product below)
// The suffix sum is accumulated back-to-front; (1-a_k)==0 falls back to an explicit product over planes.
__global__ __launch_bounds__(256) void overcompose_bwd_k(int64_t P, int D, int C, const float *__restrict__ alpha,
                                                         const float *__restrict__ content,
                                                         const float *__restrict__ g_rgb, const float *__restrict__ g_bw,
                                                         float *__restrict__ g_alpha, float *__restrict__ g_content) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float *a = alpha + p * D;
    const float *ct = content + p * D * C;
    float G[8];
    for (int c = 0; c < C; ++c) G[c] = g_rgb[p * C + c];
    // forward sweep: T_D (total transmittance) and per-plane content grads
    float Tr = 1.f;
    for (int d = 0; d < D; ++d) {
        const float w = a[d] * Tr;
        for (int c = 0; c < C; ++c) g_content[(p * D + d) * C + c] = w * G[c];
        Tr *= (1.f - a[d]);
    }
    // backward sweep: maintain T_{k} by dividing out (1-a_{k}) when safe, else recompute the prefix product
    float suffix = 0.f;        // sum_{j>k} gw_j w_j
    float Tk1 = Tr;            // T_{k+1}
    for (int d = D - 1; d >= 0; --d) {
        const float om = 1.f - a[d];
        float Tk;
        if (fabsf(om) > 1e-6f) {
            Tk = Tk1 / om;
        } else {
            Tk = 1.f;
            for (int j = 0; j < d; ++j) Tk *= (1.f - a[j]);
        }
        float gw = g_bw ? g_bw[p * D + d] : 0.f;
        for (int c = 0; c < C; ++c) gw += G[c] * ct[d * C + c];
        float behind;
        if (fabsf(om) > 1e-6f) {
            behind = suffix / om;
        } else {   // d(sum_{j>k} gw_j w_j)/da_k = -T_k * sum_{j>k} gw_j a_j prod_{k<m<j}(1-a_m)
            float r = 0.f, tt = 1.f;
            for (int j = d + 1; j < D; ++j) {
                float gwj = g_bw ? g_bw[p * D + j] : 0.f;
                for (int c = 0; c < C; ++c) gwj += G[c] * ct[j * C + c];
                r += gwj * a[j] * tt;
                tt *= (1.f - a[j]);
            }
            behind = Tk * r;
        }
        g_alpha[p * D + d] = Tk * gw - behind;
        suffix += gw * a[d] * Tk;
        Tk1 = Tk;
    }
}

// ---------------------------------------------------------------------------------------------------
// overcomposeNto0: front = LAST plane (utils_mpi.py:110-132). trans_k = prod_{j>k}(1-a_j).
__global__ __launch_bounds__(256) void nto0_fwd_k(int D, int C, int64_t HW, const float *__restrict__ alpha, int64_t a_sb,
                                                  int64_t a_sd, const float *__restrict__ content, int64_t c_sb, int64_t c_sd,
                                                  int64_t c_sc, float *__restrict__ rgb, float *__restrict__ trans) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= HW) return;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    float Tr = 1.f;
    for (int d = D - 1; d >= 0; --d) {
        const float ad = alpha[b * a_sb + d * a_sd + p];
        if (trans) trans[((int64_t)b * D + d) * HW + p] = Tr;
        const float w = ad * Tr;
        for (int c = 0; c < C; ++c) acc[c] += content[b * c_sb + d * c_sd + c * c_sc + p] * w;
        Tr *= (1.f - ad);
    }
    for (int c = 0; c < C; ++c) rgb[((int64_t)b * C + c) * HW + p] = acc[c];
}

__global__ __launch_bounds__(256) void nto0_bwd_k(int D, int C, int64_t HW, const float *__restrict__ alpha, int64_t a_sb,
                                                  int64_t a_sd, const float *__restrict__ content, int64_t c_sb, int64_t c_sd,
                                                  int64_t c_sc, const float *__restrict__ g_rgb, float *__restrict__ g_alpha,
                                                  float *__restrict__ g_content) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= HW) return;
    float G[8];
    for (int c = 0; c < C; ++c) G[c] = g_rgb[((int64_t)b * C + c) * HW + p];
    // S = sum_k w_k q_k (front-to-back = d descending), then single sweep with prefix P
    float Tr = 1.f, S = 0.f;
    for (int d = D - 1; d >= 0; --d) {
        const float ad = alpha[b * a_sb + d * a_sd + p];
        float q = 0.f;
        for (int c = 0; c < C; ++c) q += G[c] * content[b * c_sb + d * c_sd + c * c_sc + p];
        S += ad * Tr * q;
        Tr *= (1.f - ad);
    }
    Tr = 1.f;
    float Pf = 0.f;
    for (int d = D - 1; d >= 0; --d) {
        const float ad = alpha[b * a_sb + d * a_sd + p];
        float q = 0.f;
        for (int c = 0; c < C; ++c) q += G[c] * content[b * c_sb + d * c_sd + c * c_sc + p];
        const float w = ad * Tr;
        for (int c = 0; c < C; ++c) g_content[(((int64_t)b * D + d) * C + c) * HW + p] = w * G[c];
        Pf += w * q;
        const float om = 1.f - ad;
        const float behind = (fabsf(om) > 1e-12f) ? (S - Pf) / om : 0.f;
        g_alpha[((int64_t)b * D + d) * HW + p] = Tr * q - behind;
        Tr *= om;
    }
}

}  // namespace

extern "C" int vl3d_warp_fwd(int32_t N, int32_t C, int32_t Hs, int32_t Ws, int32_t h, int32_t w, const float *homos,
                             const float *images, float *out, vl3d_stream_t stream) {
    VL3D_REQUIRE(N > 0 && C > 0 && Hs > 0 && Ws > 0 && h > 0 && w > 0, "vl3d_warp_fwd: non-positive dims");
    VL3D_REQUIRE(N <= 65535, "vl3d_warp_fwd: B*D > 65535");
    VL3D_REQUIRE(homos && images && out, "vl3d_warp_fwd: null pointer");
    dim3 grid((w + 63) / 64, (h + 3) / 4, N);
    hipLaunchKernelGGL(warp_fwd_k, grid, dim3(256), 0, (hipStream_t)stream, C, Hs, Ws, h, w, homos, images, out);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_warp_bwd(int32_t N, int32_t C, int32_t Hs, int32_t Ws, int32_t h, int32_t w, const float *homos,
                             const float *grad_out, float *grad_images, vl3d_stream_t stream) {
    VL3D_REQUIRE(N > 0 && C > 0 && Hs > 0 && Ws > 0 && h > 0 && w > 0, "vl3d_warp_bwd: non-positive dims");
    VL3D_REQUIRE(N <= 65535, "vl3d_warp_bwd: B*D > 65535");
    VL3D_REQUIRE(homos && grad_out && grad_images, "vl3d_warp_bwd: null pointer");
    VL3D_HIP(hipMemsetAsync(grad_images, 0, (size_t)N * C * Hs * Ws * sizeof(float), (hipStream_t)stream));
    dim3 grid((w + 63) / 64, (h + 3) / 4, N);
    hipLaunchKernelGGL(warp_bwd_k, grid, dim3(256), 0, (hipStream_t)stream, C, Hs, Ws, h, w, homos, grad_out, grad_images);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_overcompose_fwd(int64_t P, int32_t D, int32_t C, const float *alpha, const float *content, float *rgb,
                                    float *blendweight, vl3d_stream_t stream) {
    VL3D_REQUIRE(P > 0 && D > 0 && C > 0 && C <= 8, "vl3d_overcompose_fwd: bad dims (need 1 <= C <= 8)");
    VL3D_REQUIRE(alpha && content && rgb && blendweight, "vl3d_overcompose_fwd: null pointer");
    hipLaunchKernelGGL(overcompose_fwd_k, dim3((unsigned)ceil_div64(P, 256)), dim3(256), 0, (hipStream_t)stream, P, D, C,
                       alpha, content, rgb, blendweight);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_overcompose_bwd(int64_t P, int32_t D, int32_t C, const float *alpha, const float *content,
                                    const float *grad_rgb, const float *grad_bw, float *grad_alpha, float *grad_content,
                                    vl3d_stream_t stream) {
    VL3D_REQUIRE(P > 0 && D > 0 && C > 0 && C <= 8, "vl3d_overcompose_bwd: bad dims (need 1 <= C <= 8)");
    VL3D_REQUIRE(alpha && content && grad_rgb && grad_alpha && grad_content, "vl3d_overcompose_bwd: null pointer");
    hipLaunchKernelGGL(overcompose_bwd_k, dim3((unsigned)ceil_div64(P, 256)), dim3(256), 0, (hipStream_t)stream, P, D, C,
                       alpha, content, grad_rgb, grad_bw, grad_alpha, grad_content);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_overcompose_nto0_fwd(int32_t B, int32_t D, int32_t C, int64_t HW, const float *alpha, int64_t a_sb,
                                         int64_t a_sd, const float *content, int64_t c_sb, int64_t c_sd, int64_t c_sc,
                                         float *rgb, float *trans, vl3d_stream_t stream) {
    VL3D_REQUIRE(B > 0 && B <= 65535 && D > 0 && C > 0 && C <= 8 && HW > 0, "vl3d_overcompose_nto0_fwd: bad dims");
    VL3D_REQUIRE(alpha && content && rgb, "vl3d_overcompose_nto0_fwd: null pointer");
    hipLaunchKernelGGL(nto0_fwd_k, dim3((unsigned)ceil_div64(HW, 256), B), dim3(256), 0, (hipStream_t)stream, D, C, HW, alpha,
                       a_sb, a_sd, content, c_sb, c_sd, c_sc, rgb, trans);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_overcompose_nto0_bwd(int32_t B, int32_t D, int32_t C, int64_t HW, const float *alpha, int64_t a_sb,
                                         int64_t a_sd, const float *content, int64_t c_sb, int64_t c_sd, int64_t c_sc,
                                         const float *grad_rgb, float *grad_alpha, float *grad_content,
                                         vl3d_stream_t stream) {
    VL3D_REQUIRE(B > 0 && B <= 65535 && D > 0 && C > 0 && C <= 8 && HW > 0, "vl3d_overcompose_nto0_bwd: bad dims");
    VL3D_REQUIRE(alpha && content && grad_rgb && grad_alpha && grad_content, "vl3d_overcompose_nto0_bwd: null pointer");
    hipLaunchKernelGGL(nto0_bwd_k, dim3((unsigned)ceil_div64(HW, 256), B), dim3(256), 0, (hipStream_t)stream, D, C, HW, alpha,
                       a_sb, a_sd, content, c_sb, c_sd, c_sc, grad_rgb, grad_alpha, grad_content);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

// ---------------------------------------------------------------------------------------------------
// Static tiles of a tile-culled video stack (MPI.py:288-442 / MPV.py:235-288: the reference keeps ONE static atlas shared by
// all frames): the dense stack holds T copies, kept identical by giving every copy the SUM of the T per-frame gradients.
// In place on grad (D,T,Hs,Ws,4): texels only static quads can read -> sum over frames in every frame; texels no kept quad
// can read -> 0; texels a dynamic quad can read -> untouched.  "Can read" = the quad's closed rectangle grown by one texel
// (the bilinear taps of a sample inside it), as in videoloop3d_amd/tiles.py quad_to_texel_mask.
__global__ __launch_bounds__(256) void tie_static_grad_k(int D, int T, int Hs, int Ws, vl3d_adam::Quads q, float4 *__restrict__ g,
                                                         int assume_culled_zero) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), d = blockIdx.z;
    if (x >= Ws || y >= Hs) return;
    const int cls = vl3d_adam::texel_class(q, d, x, y, Hs, Ws);      // 0 culled, 1 dynamic, 2 static (shared-border or tile-exact layout)
    const bool kept = cls != 0;
    if (cls == 1) return;
    const size_t frame = (size_t)Hs * Ws;
    float4 *p = g + (size_t)d * T * frame + (size_t)y * Ws + x;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!kept) {
        if (assume_culled_zero & 1) return;  // the culled render never writes anything but 0 there
    } else {
        for (int t = 0; t < T; ++t) {
            const float4 v = p[(size_t)t * frame];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    // bit 1 of the mode: the consumer is the tile-aware Adam, which reads a static texel's gradient from frame 0 only
    const int tw = (kept && (assume_culled_zero & 2)) ? 1 : T;
    for (int t = 0; t < tw; ++t) p[(size_t)t * frame] = s;
}

extern "C" int vl3d_tie_static_grad(int32_t D, int32_t T, int32_t Hs, int32_t Ws, const uint8_t *quad_keep, const uint8_t *quad_dyn,
                                    int32_t QH, int32_t QW, float *grad, int32_t assume_culled_zero, vl3d_stream_t stream) {
    VL3D_REQUIRE(D > 0 && T > 0 && Hs > 0 && Ws > 0 && vl3d_adam::quad_grid_ok(QH, QW, Hs, Ws) && D <= 65535, "vl3d_tie_static_grad: bad dims / quad grid");
    VL3D_REQUIRE(quad_keep && quad_dyn && grad, "vl3d_tie_static_grad: null pointer");
    hipLaunchKernelGGL(tie_static_grad_k, dim3((Ws + 63) / 64, (Hs + 3) / 4, D), dim3(256), 0, (hipStream_t)stream, D, T, Hs, Ws,
                       vl3d_adam::make_quads(quad_keep, quad_dyn, QH, QW, Hs, Ws), reinterpret_cast<float4 *>(grad), assume_culled_zero);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

// ---------------------------------------------------------------------------------------------------
// Adam step (torch.optim.Adam without amsgrad / weight decay: MPV.py:199-214 uses betas (0.9, 0.999), eps 6e-8) restricted to
// the texels a kept quad can read.  Culled texels never receive a gradient, so their moments stay 0 and Adam would leave them
// unchanged anyway -- skipping them removes 7 streams over the culled part of the stack (the optimiser is 2/3 of a stage-2
// iteration on the dense stack).  m, v, p are updated in place; bc1 = 1 - beta1^step, bc2s = sqrt(1 - beta2^step).
__global__ __launch_bounds__(256) void adam_tiles_k(int T, int Hs, int Ws, vl3d_adam::Quads q,
                                                    float4 *__restrict__ p, const float4 *__restrict__ g, float4 *__restrict__ m,
                                                    float4 *__restrict__ v, float lr_bc1, float beta1, float beta2, float eps, float bc2s) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), d = blockIdx.z;
    if (x >= Ws || y >= Hs) return;
    bool is_static = false;
    if (q.keep) {
        const int cls = vl3d_adam::texel_class(q, d, x, y, Hs, Ws);      // 0 culled, 1 dynamic, 2 static
        if (cls == 0) return;
        is_static = cls == 2;
    }
    const size_t frame = (size_t)Hs * Ws;
    size_t o = (size_t)d * T * frame + (size_t)y * Ws + x;
    auto upd = [&](float &pp, float gg, float &mm, float &vv) {
        mm = beta1 * mm + (1.0f - beta1) * gg;           // exp_avg.lerp_(grad, 1 - beta1)
        vv = beta2 * vv + (1.0f - beta2) * gg * gg;      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        pp -= lr_bc1 * (mm / (sqrtf(vv) / bc2s + eps));  // param.addcdiv_(exp_avg, sqrt(exp_avg_sq)/sqrt(bc2) + eps, value=-lr/bc1)
    };
    if (is_static) {
        // a static texel is ONE parameter with T identical copies (its gradient was summed over the frames into frame 0 by
        // vl3d_tie_static_grad): update it once from frame 0's (p, g, m, v) and write the value to every copy
        float4 pp = p[o], mm = m[o], vv = v[o];
        const float4 gg = g[o];
        upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
        m[o] = mm; v[o] = vv;
        for (int t = 0; t < T; ++t, o += frame) p[o] = pp;
        return;
    }
    for (int t = 0; t < T; ++t, o += frame) {
        float4 mm = m[o], vv = v[o];
        const float4 gg = g[o];
        // A texel no view has reached yet (g = m = v = 0: the margins of planes stored larger than the frame, half of a stage-1 stack at the
        // shipped 1.6x) keeps m = v = 0 and takes a step of exactly 0: its parameter is neither read nor written, its moments not written --
        // three streams instead of seven over that part of the stack, the same bits as the full update (-0 counts as 0).
        const unsigned any = (__float_as_uint(gg.x) | __float_as_uint(gg.y) | __float_as_uint(gg.z) | __float_as_uint(gg.w) |
                              __float_as_uint(mm.x) | __float_as_uint(mm.y) | __float_as_uint(mm.z) | __float_as_uint(mm.w) |
                              __float_as_uint(vv.x) | __float_as_uint(vv.y) | __float_as_uint(vv.z) | __float_as_uint(vv.w)) & 0x7fffffffu;
        if (any == 0u) continue;
        float4 pp = p[o];
        upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
        p[o] = pp; m[o] = mm; v[o] = vv;
    }
}

static int adam_step_tiles_impl(int32_t D, int32_t T, int32_t Hs, int32_t Ws, const uint8_t *quad_keep, const uint8_t *quad_dyn,
                                int32_t QH, int32_t QW, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float lr_bc1,
                                float beta1, float beta2, float eps, float bc2s, vl3d_stream_t stream) {
    VL3D_REQUIRE(D > 0 && D <= 65535 && T > 0 && Hs > 0 && Ws > 0, "vl3d_adam_step_tiles: bad dims");
    VL3D_REQUIRE(param && grad && exp_avg && exp_avg_sq, "vl3d_adam_step_tiles: null pointer");
    VL3D_REQUIRE(!quad_keep || vl3d_adam::quad_grid_ok(QH, QW, Hs, Ws), "vl3d_adam_step_tiles: bad quad grid");
    hipLaunchKernelGGL(adam_tiles_k, dim3((Ws + 63) / 64, (Hs + 3) / 4, D), dim3(256), 0, (hipStream_t)stream, T, Hs, Ws,
                       vl3d_adam::make_quads(quad_keep, quad_dyn, QH, QW, Hs, Ws),
                       reinterpret_cast<float4 *>(param), reinterpret_cast<const float4 *>(grad), reinterpret_cast<float4 *>(exp_avg),
                       reinterpret_cast<float4 *>(exp_avg_sq), lr_bc1, beta1, beta2, eps, bc2s);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_adam_step_tiles(int32_t D, int32_t T, int32_t Hs, int32_t Ws, const uint8_t *quad_keep, const uint8_t *quad_dyn,
                                    int32_t QH, int32_t QW,
                                    float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float lr, float beta1,
                                    float beta2, float eps, int64_t step, vl3d_stream_t stream) {
    VL3D_REQUIRE(step >= 1, "vl3d_adam_step_tiles: bad step");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    return adam_step_tiles_impl(D, T, Hs, Ws, quad_keep, quad_dyn, QH, QW, param, grad, exp_avg, exp_avg_sq, (float)((double)lr / bc1), beta1, beta2,
                                eps, (float)sqrt(bc2), stream);
}
