// Looping loss for gfx950: per-location temporal patch nearest-neighbour search (K3), vote-fold of the
// matched patches (K4) and the robust loss (K5).
//
// Replaces (reference, /root/reference): utils_vid.py:60-69 (extract_3Dpatches / unfoldNd), :72-86
// (efficient_compute_distances), :109-142 (column mins + argmin), :206-229 (FindNNpatchAndMerge / FoldNd),
// :10-26 + :348 (robust_lossfun().mean()).
//
// Formulation (SURVEY §7 "Gram vs box-filter"): patches are never materialised.  For a spatial location
// b the frame-pair energy  E[i',j'] = sum_{c,kh,kw} (x[c,i',.] - y[c,j',.])^2  is accumulated once
// (3*ps^2 terms per pair), and the patch distance is its temporal diagonal sum
// dist[i,j] = sum_kt E[i*st+kt, j*st+kt] / (3*pt*ps^2): pt-fold fewer flops than the patch-level Gram and
// no |x|^2+|y|^2-2xy cancellation (never negative).
#include "vl3d_common.h"

namespace {

constexpr int NN_THREADS = 256;
constexpr int TI = 4, TJ = 4;   // per-thread register tile of frame pairs

struct NNArgs {
    const float *x, *y;
    int32_t *nn;
    int Tx, Ty, ps, pt, stride, stridet, h_o, w_o, n1, n2;
    int TxU;            // frames of x actually covered by patches
    int TxP, TyP;       // padded to multiples of TI/TJ
    int K, KC;          // K = 3*ps*ps, chunk size
    int use_alpha;
    float alpha, inv_d;   // inv_d holds the divisor d = 3*pt*ps^2
    int64_t x_sc, x_st, x_sr, y_sc, y_st, y_sr;
};

// LDS layout (floats): Xs[KC][TxP] | Ys[KC][TyP] | E[TxP][TyP] | colmin[n2]
__global__ __launch_bounds__(NN_THREADS) void patchnn_k(NNArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Xs = smem;
    float *Ys = Xs + (size_t)a.KC * a.TxP;
    float *E = Ys + (size_t)a.KC * a.TyP;
    float *colmin = E + (size_t)a.TxP * a.TyP;

    const int b = blockIdx.x;
    const int by = b / a.w_o, bx = b % a.w_o;
    const int r0 = by * a.stride, c0 = bx * a.stride;
    const int tid = threadIdx.x;
    const int ps2 = a.ps * a.ps;
    const int tiles_j = a.TyP / TJ, ntiles = (a.TxP / TI) * tiles_j;

    for (int i = tid; i < a.TxP * a.TyP; i += NN_THREADS) E[i] = 0.f;

    for (int k0 = 0; k0 < a.K; k0 += a.KC) {
        const int kc = min(a.KC, a.K - k0);
        __syncthreads();
        // stage chunk: Xs[k][f] = x[c, f, r0+kh, c0+kw]; k fastest across lanes -> row segments of ps floats
        for (int i = tid; i < kc * a.TxP; i += NN_THREADS) {
            const int f = i / kc, kk = i - f * kc;
            const int k = k0 + kk;
            const int c = k / ps2, r = (k - c * ps2) / a.ps, q = k - c * ps2 - r * a.ps;
            Xs[kk * a.TxP + f] = (f < a.TxU) ? a.x[c * a.x_sc + f * a.x_st + (int64_t)(r0 + r) * a.x_sr + c0 + q] : 0.f;
        }
        for (int i = tid; i < kc * a.TyP; i += NN_THREADS) {
            const int f = i / kc, kk = i - f * kc;
            const int k = k0 + kk;
            const int c = k / ps2, r = (k - c * ps2) / a.ps, q = k - c * ps2 - r * a.ps;
            Ys[kk * a.TyP + f] = (f < a.Ty) ? a.y[c * a.y_sc + f * a.y_st + (int64_t)(r0 + r) * a.y_sr + c0 + q] : 0.f;
        }
        __syncthreads();
        for (int tile = tid; tile < ntiles; tile += NN_THREADS) {
            const int ti = (tile / tiles_j) * TI, tj = (tile % tiles_j) * TJ;
            float acc[TI][TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = 0.f;
            for (int kk = 0; kk < kc; ++kk) {
                const float4 xv = *reinterpret_cast<const float4 *>(Xs + kk * a.TxP + ti);
                const float4 yv = *reinterpret_cast<const float4 *>(Ys + kk * a.TyP + tj);
                const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, ya[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        const float df = xa[i] - ya[j];
                        acc[i][j] += df * df;
                    }
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) E[(ti + i) * a.TyP + tj + j] += acc[i][j];
        }
    }
    __syncthreads();

    // dist[i][j] = sum_kt E[i*st+kt][j*st+kt] * (1/d)   (utils_vid.py:82-84: divide by d)
    if (a.use_alpha) {
        for (int j = tid; j < a.n2; j += NN_THREADS) {
            float m = INFINITY;
            for (int i = 0; i < a.n1; ++i) {
                float s = 0.f;
                for (int kt = 0; kt < a.pt; ++kt) s += E[(i * a.stridet + kt) * a.TyP + j * a.stridet + kt];
                m = fminf(m, s / a.inv_d);
            }
            colmin[j] = a.alpha + m;                      // utils_vid.py:133-134
        }
        __syncthreads();
    }
    for (int i = tid; i < a.n1; i += NN_THREADS) {
        float best = INFINITY;
        int bj = 0;
        bool best_nan = false;
        for (int j = 0; j < a.n2; ++j) {
            float s = 0.f;
            for (int kt = 0; kt < a.pt; ++kt) s += E[(i * a.stridet + kt) * a.TyP + j * a.stridet + kt];
            float v = s / a.inv_d;
            if (a.use_alpha) v = v / colmin[j];           // utils_vid.py:140
            // torch.argmin: first minimum, NaN counts as minimal
            const bool vn = (v != v);
            if (!best_nan && (vn || v < best)) { best = v; bj = j; best_nan = vn; }
        }
        a.nn[(size_t)b * a.n1 + i] = bj;
    }
}

// ---------------------------------------------------------------------------------------------------
struct FoldArgs {
    const float *y;
    const int32_t *nn;
    float *sum, *weight;
    int Tx, H, W, ps, pt, stride, stridet, h_o, w_o, n1;
    int64_t y_sc, y_st, y_sr;
    int normalize;
};

__global__ __launch_bounds__(256) void vote_fold_k(FoldArgs a) {
    const int xi = blockIdx.x * 64 + (threadIdx.x & 63);
    const int eta = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int tau = blockIdx.z;
    if (xi >= a.W || eta >= a.H) return;
    // covering patch rows/cols: by*s <= eta < by*s+ps
    const int by_hi = min(a.h_o - 1, eta / a.stride);
    const int by_lo = max(0, (eta - a.ps + a.stride) / a.stride);   // ceil((eta-ps+1)/s) for eta-ps+1 > 0
    const int bx_hi = min(a.w_o - 1, xi / a.stride);
    const int bx_lo = max(0, (xi - a.ps + a.stride) / a.stride);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    int cnt = 0;
    const int64_t pix = (int64_t)eta * a.y_sr + xi;
    for (int kt = 0; kt < a.pt; ++kt) {
        const int ts = tau - kt;
        if (ts < 0 || (ts % a.stridet) != 0) continue;
        const int i = ts / a.stridet;
        if (i >= a.n1) continue;
        for (int by = by_lo; by <= by_hi; ++by)
            for (int bx = bx_lo; bx <= bx_hi; ++bx) {
                const int j = a.nn[((size_t)by * a.w_o + bx) * a.n1 + i];
                const int64_t off = (int64_t)(j * a.stridet + kt) * a.y_st + pix;
                s0 += a.y[off];
                s1 += a.y[a.y_sc + off];
                s2 += a.y[2 * a.y_sc + off];
                ++cnt;
            }
    }
    const float wgt = fmaxf((float)cnt, 1e-10f);                  // utils_vid.py:228
    const size_t o = ((size_t)tau * a.H + eta) * a.W + xi;
    const size_t cs = (size_t)a.Tx * a.H * a.W;
    if (a.normalize) { s0 /= wgt; s1 /= wgt; s2 /= wgt; }
    a.sum[o] = s0; a.sum[cs + o] = s1; a.sum[2 * cs + o] = s2;
    a.weight[o] = wgt;
}

// ---------------------------------------------------------------------------------------------------
// robust loss (utils_vid.py:10-26)
struct Rho {
    int kind;      // 0 mse, 1 abs, 2 log1p (rou==0), 3 quadratic (rou==2), 4 general
    float scale, b, d, coef;
};

__device__ __forceinline__ float rho_f(const Rho &r, float e) {
    switch (r.kind) {
        case 0: return e * e;
        case 1: return fabsf(e);
        case 2: { float s = (e / r.scale); return log1pf(s * s * 0.5f); }
        case 3: { float s = (e / r.scale); return 0.5f * s * s; }
        default: { float s = (e / r.scale); s = s * s; return r.coef * (powf(s / r.b + 1.0f, 0.5f * r.d) - 1.0f) * (r.scale * 10.0f); }
    }
}

__device__ __forceinline__ float rho_g(const Rho &r, float e) {
    switch (r.kind) {
        case 0: return 2.0f * e;
        case 1: return e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
        case 2: { float s = e / r.scale; return (s / r.scale) / (1.0f + 0.5f * s * s); }
        case 3: return e / (r.scale * r.scale);
        default: {   // d/de (b/d)((s/b+1)^(d/2)-1)*10*scale, s=(e/scale)^2  ->  10*(e/scale)*(s/b+1)^(d/2-1)
            float s = e / r.scale;
            return 10.0f * s * powf(s * s / r.b + 1.0f, 0.5f * r.d - 1.0f);
        }
    }
}

__global__ __launch_bounds__(256) void robust_fwd_k(int64_t n, const float *__restrict__ x, const float *__restrict__ y2x,
                                                    Rho r, double *__restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        acc += rho_f(r, x[i] - y2x[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]);
}

__global__ __launch_bounds__(256) void robust_bwd_k(int64_t n, const float *__restrict__ x, const float *__restrict__ y2x,
                                                    Rho r, const float *__restrict__ gout, float inv_n,
                                                    float *__restrict__ gx) {
    const float g = (*gout) * inv_n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        gx[i] = rho_g(r, x[i] - y2x[i]) * g;
}

Rho make_rho(int kind, float rou, float scale) {
    Rho r{};
    r.scale = scale;
    if (kind == VL3D_RHO_MSE) r.kind = 0;
    else if (kind == VL3D_RHO_ABS) r.kind = 1;
    else if (rou == 0.0f) r.kind = 2;
    else if (rou == 2.0f) r.kind = 3;
    else {
        r.kind = 4;
        const float eps = 1e-6f;
        r.b = fabsf(rou - 2.0f) + eps;
        r.d = rou >= 0.f ? rou + eps : rou - eps;
        r.coef = r.b / r.d;
    }
    return r;
}

int check_loss(const vl3d_loss_desc *d) {
    VL3D_REQUIRE(d != nullptr, "null loss desc");
    VL3D_REQUIRE(d->Tx > 0 && d->Ty > 0 && d->H > 0 && d->W > 0, "loss: non-positive dims");
    VL3D_REQUIRE(d->ps > 0 && d->pt > 0 && d->stride > 0 && d->stridet > 0, "loss: non-positive patch config");
    VL3D_REQUIRE(d->H >= d->ps && d->W >= d->ps && d->Tx >= d->pt && d->Ty >= d->pt, "loss: input smaller than one patch");
    VL3D_REQUIRE((d->H - d->ps) % d->stride == 0 && (d->W - d->ps) % d->stride == 0 && (d->Tx - d->pt) % d->stridet == 0,
                 "loss: x is not trimmed to the patch grid (utils_vid.py:307-320)");
    return VL3D_OK;
}

constexpr int LDS_BUDGET = 48 * 1024;      // target: 3 blocks per CU
constexpr int LDS_MAX = 150 * 1024;

int plan_nn(const vl3d_loss_desc *d, NNArgs &a, size_t &lds) {
    a.Tx = d->Tx; a.Ty = d->Ty; a.ps = d->ps; a.pt = d->pt; a.stride = d->stride; a.stridet = d->stridet;
    a.h_o = (d->H - d->ps) / d->stride + 1;
    a.w_o = (d->W - d->ps) / d->stride + 1;
    a.n1 = (d->Tx - d->pt) / d->stridet + 1;
    a.n2 = (d->Ty - d->pt) / d->stridet + 1;
    a.TxU = (a.n1 - 1) * d->stridet + d->pt;
    a.TxP = (a.TxU + TI - 1) / TI * TI;
    a.TyP = (d->Ty + TJ - 1) / TJ * TJ;
    a.K = 3 * d->ps * d->ps;
    a.use_alpha = d->use_alpha; a.alpha = d->alpha;
    a.inv_d = (float)(3 * d->pt * d->ps * d->ps);   // divisor d (utils_vid.py:83-84 divides)
    a.x_sc = d->x_sc; a.x_st = d->x_st; a.x_sr = d->x_sr; a.y_sc = d->y_sc; a.y_st = d->y_st; a.y_sr = d->y_sr;
    const size_t fixed = ((size_t)a.TxP * a.TyP + a.n2) * sizeof(float);
    const size_t per_k = (size_t)(a.TxP + a.TyP) * sizeof(float);
    VL3D_REQUIRE(fixed + 8 * per_k <= (size_t)LDS_MAX,
                 "loss: Tx*Ty frame-pair matrix does not fit in LDS (Tx*Ty too large for this round's kernel)");
    size_t budget = (size_t)LDS_BUDGET;
    if (fixed + 32 * per_k > budget) budget = (fixed + 32 * per_k < (size_t)LDS_MAX) ? fixed + 32 * per_k : (size_t)LDS_MAX;
    int kc = (int)((budget - fixed) / per_k);
    a.KC = kc < a.K ? kc : a.K;
    lds = fixed + (size_t)a.KC * (a.TxP + a.TyP) * sizeof(float);
    return VL3D_OK;
}

}  // namespace

extern "C" int64_t vl3d_patchnn_scratch_bytes(const vl3d_loss_desc *) { return 0; }

extern "C" int vl3d_patchnn(const vl3d_loss_desc *desc, const float *x, const float *y, int32_t *nn, void *,
                            vl3d_stream_t stream) {
    int rc = check_loss(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(x && y && nn, "vl3d_patchnn: null pointer");
    NNArgs a{};
    size_t lds = 0;
    rc = plan_nn(desc, a, lds);
    if (rc != VL3D_OK) return rc;
    a.x = x; a.y = y; a.nn = nn;
    static bool attr_set = false;
    if (!attr_set) {
        VL3D_HIP(hipFuncSetAttribute((const void *)patchnn_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(patchnn_k, dim3((unsigned)(a.h_o * a.w_o)), dim3(NN_THREADS), lds, (hipStream_t)stream, a);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_vote_fold(const vl3d_loss_desc *desc, const float *y, const int32_t *nn, float *sum, float *weight,
                              int32_t normalize, vl3d_stream_t stream) {
    int rc = check_loss(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(y && nn && sum && weight, "vl3d_vote_fold: null pointer");
    VL3D_REQUIRE(desc->Tx <= 65535, "vl3d_vote_fold: Tx > 65535");
    FoldArgs a{};
    a.y = y; a.nn = nn; a.sum = sum; a.weight = weight;
    a.Tx = desc->Tx; a.H = desc->H; a.W = desc->W; a.ps = desc->ps; a.pt = desc->pt;
    a.stride = desc->stride; a.stridet = desc->stridet;
    a.h_o = (desc->H - desc->ps) / desc->stride + 1;
    a.w_o = (desc->W - desc->ps) / desc->stride + 1;
    a.n1 = (desc->Tx - desc->pt) / desc->stridet + 1;
    a.y_sc = desc->y_sc; a.y_st = desc->y_st; a.y_sr = desc->y_sr;
    a.normalize = normalize;
    dim3 grid((desc->W + 63) / 64, (desc->H + 3) / 4, desc->Tx);
    hipLaunchKernelGGL(vote_fold_k, grid, dim3(256), 0, (hipStream_t)stream, a);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_robust_fwd(int64_t n, const float *x, const float *y2x, int32_t kind, float rou, float scale,
                               double *loss_sum, vl3d_stream_t stream) {
    VL3D_REQUIRE(n > 0 && x && y2x && loss_sum, "vl3d_robust_fwd: bad arguments");
    VL3D_REQUIRE(kind >= 0 && kind <= 2 && scale != 0.0f, "vl3d_robust_fwd: bad rho kind / scale");
    VL3D_HIP(hipMemsetAsync(loss_sum, 0, sizeof(double), (hipStream_t)stream));
    const unsigned blocks = (unsigned)(ceil_div64(n, 256) < 4096 ? ceil_div64(n, 256) : 4096);
    hipLaunchKernelGGL(robust_fwd_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, x, y2x, make_rho(kind, rou, scale),
                       loss_sum);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_robust_bwd(int64_t n, const float *x, const float *y2x, int32_t kind, float rou, float scale,
                               const float *grad_out, float inv_n, float *grad_x, vl3d_stream_t stream) {
    VL3D_REQUIRE(n > 0 && x && y2x && grad_out && grad_x, "vl3d_robust_bwd: bad arguments");
    VL3D_REQUIRE(kind >= 0 && kind <= 2 && scale != 0.0f, "vl3d_robust_bwd: bad rho kind / scale");
    const unsigned blocks = (unsigned)(ceil_div64(n, 256) < 4096 ? ceil_div64(n, 256) : 4096);
    hipLaunchKernelGGL(robust_bwd_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, x, y2x, make_rho(kind, rou, scale),
                       grad_out, inv_n, grad_x);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}
