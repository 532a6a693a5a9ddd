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
    int ablate;          // measurement-only switches (bit0: skip LDS scatter, bit1: skip flush stores, bit2: skip tap loads)
    const float *plan;   // device scratch written by bwd_plan_k: [0] feasible flag, [16 + 9*d ..] inverse texel homographies
};

struct Taps {
    int idx[4];     // texel index (y*Ws+x) of each tap, -1 if out of range
    float w[4];     // bilinear weights
    int x0, y0;     // top-left tap
    float tx, ty;   // texel coordinates of the sample
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
    t.tx = tx; t.ty = ty;
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
    if (a.plan && reinterpret_cast<const int *>(a.plan)[0]) return;   // the tile path owns this call
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


// =====================================================================================================
// Backward, variant "tile": LDS-staged owner-computes accumulation (no global atomics, no memset).
//
// A workgroup owns an output tile of (RW-2) x (ROWS-2) pixels and additionally recomputes a 1-pixel halo
// ring (RW x ROWS pixel region, one wave per region row).  Planes are walked front to back with the
// per-pixel composite state in registers; for every plane each pixel of the region stages its texel
// coordinates and its 4-channel gradient in LDS (plain ds_write), then every texel whose OWNER pixel
//     p0(tau) = clamp_to_frame(round(H_d^-1 tau))
// lies inside this workgroup's tile GATHERS its bilinear taps from the 3x3 staged pixels around p0 and is
// written with one coalesced 16-byte store.  All contributions to a texel come from pixels within
// |J^-1|_inf + 0.5 < 2 of its owner pixel, i.e. from the tile + 1-pixel halo, so each texel is written exactly
// once with its complete sum, in a fixed order (bitwise reproducible; no atomics at all -- a first version
// that scattered with ds_add_f32 into an LDS window measured 158 ms vs 30 ms without the LDS atomics).  Texels whose owner pixel is outside the
// frame are zero-filled by bwd_zero_unowned_k (run first; it also zeroes a 1-pixel safety band that the
// tile kernel then overwrites).  bwd_plan_k checks the geometric preconditions per call ON DEVICE
// (Z>0 over the frame, magnification < 1.4x, window fits); if they fail, these kernels exit and the
// universal atomics kernel above runs instead -- no host synchronisation either way.
constexpr int RW = 64;        // region width in pixels = one wave
constexpr int PLAN_HDR = 16;  // floats before the per-plane matrices

// texel-space homography  Ht = A_tex * H  (double), and its inverse
template <int COORD>
__device__ void texel_homography(const float *h, int Hs, int Ws, float sx, float sy, float ox, float oy, double M[9]) {
    double ax, ay, bx, by;
    if constexpr (COORD == VL3D_COORD_UTILS_MPI) { ax = (double)(Ws - 1) / Ws; ay = (double)(Hs - 1) / Hs; bx = by = 0.0; }
    else { ax = sx; ay = sy; bx = ox; by = oy; }
    for (int j = 0; j < 3; ++j) {
        M[0 + j] = ax * h[0 + j] + bx * h[6 + j];
        M[3 + j] = ay * h[3 + j] + by * h[6 + j];
        M[6 + j] = h[6 + j];
    }
}

template <int COORD>
__global__ void bwd_plan_k(RenderArgs a, int rows, float *plan) {
    __shared__ int ok_all;
    if (threadIdx.x == 0) ok_all = 1;
    __syncthreads();
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
        double M[9];
        texel_homography<COORD>(a.homos + 9 * d, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, M);
        const double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
                           M[2] * (M[3] * M[7] - M[4] * M[6]);
        bool ok = (det == det) && fabs(det) > 1e-30;
        double I[9];
        I[0] = (M[4] * M[8] - M[5] * M[7]) / det; I[1] = (M[2] * M[7] - M[1] * M[8]) / det; I[2] = (M[1] * M[5] - M[2] * M[4]) / det;
        I[3] = (M[5] * M[6] - M[3] * M[8]) / det; I[4] = (M[0] * M[8] - M[2] * M[6]) / det; I[5] = (M[2] * M[3] - M[0] * M[5]) / det;
        I[6] = (M[3] * M[7] - M[4] * M[6]) / det; I[7] = (M[1] * M[6] - M[0] * M[7]) / det; I[8] = (M[0] * M[4] - M[1] * M[3]) / det;
        // normalise so the pixel-space w of the frame centre is ~1 (keeps fp32 well scaled)
        for (int i = 0; i < 9; ++i) plan[PLAN_HDR + 9 * d + i] = (float)I[i];
        // geometric preconditions at a 3x3 grid of points of the (halo-extended) frame
        for (int gy = 0; gy < 3 && ok; ++gy)
            for (int gx = 0; gx < 3 && ok; ++gx) {
                const double x = a.col0 + a.pc + (gx == 0 ? -2.0 : (gx == 1 ? 0.5 * a.W : a.W + 1.0));
                const double y = a.row0 + a.pc + (gy == 0 ? -2.0 : (gy == 1 ? 0.5 * a.H : a.H + 1.0));
                const double X = M[0] * x + M[1] * y + M[2], Y = M[3] * x + M[4] * y + M[5], Z = M[6] * x + M[7] * y + M[8];
                if (!(Z > 1e-20)) { ok = false; break; }
                const double j00 = (M[0] * Z - X * M[6]) / (Z * Z), j01 = (M[1] * Z - X * M[7]) / (Z * Z);
                const double j10 = (M[3] * Z - Y * M[6]) / (Z * Z), j11 = (M[4] * Z - Y * M[7]) / (Z * Z);
                const double dj = j00 * j11 - j01 * j10;
                if (!(fabs(dj) > 1e-12)) { ok = false; break; }
                // |J^-1|_inf < 1.4  (contributions to a texel stay within the 1-pixel halo of its owner pixel)
                const double i_r0 = (fabs(j11) + fabs(j01)) / fabs(dj), i_r1 = (fabs(j10) + fabs(j00)) / fabs(dj);
                if (!(i_r0 < 1.4 && i_r1 < 1.4)) ok = false;
                // keep the owned footprint of a tile a small multiple of the workgroup (pure efficiency guard)
                if (!(fabs(j00) + fabs(j01) < 4.0 && fabs(j10) + fabs(j11) < 4.0)) ok = false;
            }
        if (!ok) atomicAnd(&ok_all, 0);
    }
    __syncthreads();
    if (threadIdx.x == 0) reinterpret_cast<int *>(plan)[0] = ok_all;
}

// owner pixel (float, before rounding) of texel (tx,ty) on plane d, relative to this window's pixel origin
__device__ __forceinline__ void owner_pixel(const float *__restrict__ hi, float tx, float ty, float pc, int col0, int row0,
                                            float &px, float &py) {
    const float X = hi[0] * tx + hi[1] * ty + hi[2];
    const float Y = hi[3] * tx + hi[4] * ty + hi[5];
    const float Z = hi[6] * tx + hi[7] * ty + hi[8];
    px = X / Z - pc - (float)col0;
    py = Y / Z - pc - (float)row0;
}

__global__ __launch_bounds__(256) void bwd_zero_unowned_k(RenderArgs a) {
    if (!reinterpret_cast<const int *>(a.plan)[0]) return;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int d = blockIdx.z;
    if (x >= a.Ws || y >= a.Hs) return;
    float px, py;
    owner_pixel(a.plan + PLAN_HDR + 9 * d, (float)x, (float)y, a.pc, a.col0, a.row0, px, py);
    const bool safe = (px > 0.5f) && (px < (float)a.W - 1.5f) && (py > 0.5f) && (py < (float)a.H - 1.5f);
    if (safe) return;      // owned (and written) by a tile with certainty
    const size_t frame = (size_t)a.Hs * a.Ws;
    float4 *g = reinterpret_cast<float4 *>(a.g_stack) + (size_t)d * a.T * frame + (size_t)y * a.Ws + x;
    for (int t = 0; t < a.T; ++t, g += frame) *g = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void bwd_fill_zero_if_infeasible_k(float4 *g, size_t n, const float *plan) {
    if (reinterpret_cast<const int *>(plan)[0]) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int COORD, int BORDER, int ORDER, int RACT, int AACT, int ROWS>
__global__ __launch_bounds__(RW *ROWS) void render_bwd_tile_k(RenderArgs a) {
    if (!reinterpret_cast<const int *>(a.plan)[0]) return;
    constexpr int NT = RW * ROWS;
    // per-plane staging of the region's pixels, double buffered so one barrier per plane suffices
    __shared__ float4 s_g[2][NT];     // gradient w.r.t. the sampled (POST) / activated (PRE) value of this pixel on this plane
    __shared__ float2 s_t[2][NT];     // its texel coordinates (tx,ty); -1e30 when the plane does not cover the pixel
    __shared__ float s_c[2][8];       // footprint corners of the owned tile on this plane
    const int tid = threadIdx.x, lane = tid & 63, row = tid >> 6;
    const int rx0 = blockIdx.x * (RW - 2) - 1, ry0 = blockIdx.y * (ROWS - 2) - 1;
    const int x = rx0 + lane, y = ry0 + row, t = blockIdx.z;
    const bool inimg = (x >= 0) && (x < a.W) && (y >= 0) && (y < a.H);
    // owned (interior) pixel range of this workgroup, clipped to the frame: [ix0,ix1] x [iy0,iy1]
    const int ix0 = max(rx0 + 1, 0), ix1 = min(rx0 + RW - 2, a.W - 1);
    const int iy0 = max(ry0 + 1, 0), iy1 = min(ry0 + ROWS - 2, a.H - 1);
    const float px = (float)(a.col0 + x) + a.pc, py = (float)(a.row0 + y) + a.pc;
    const size_t frame = (size_t)a.Hs * a.Ws * 4;
    const size_t plane_stride = (size_t)a.T * frame;
    const float *plane = a.stack + (size_t)t * frame;
    float *gplane = a.g_stack + (size_t)t * frame;
    float Gr = 0.f, Gg = 0.f, Gb = 0.f, gA = 0.f, S = 0.f;
    if (inimg) {
        const size_t pix = ((size_t)t * a.H + y) * a.W + x;
        Gr = a.g_rgb[pix * 3 + 0]; Gg = a.g_rgb[pix * 3 + 1]; Gb = a.g_rgb[pix * 3 + 2];
        gA = a.g_alpha ? a.g_alpha[pix] : 0.0f;
        S = Gr * a.rgb[pix * 3 + 0] + Gg * a.rgb[pix * 3 + 1] + Gb * a.rgb[pix * 3 + 2] + gA * a.alpha[pix];
    }
    float Tr = 1.0f, P = 0.0f;
    for (int d = 0; d < a.D; ++d, plane += plane_stride, gplane += plane_stride) {
        const float *h = a.homos + 9 * d;
        const int buf = d & 1;
        // (1) footprint of the owned tile on plane d from its four corners (convex image of a rectangle, Z>0).
        //     Texels owned by this tile have their owner pixel inside it, i.e. H^-1(tau) within 0.5 px of the tile
        //     (any distance beyond a frame border, where only texels within the 1.4 px contribution range matter).
        if (tid < 4) {
            const float el = (ix0 == 0) ? 1.6f : 0.6f, er = (ix1 == a.W - 1) ? 1.6f : 0.6f;
            const float et = (iy0 == 0) ? 1.6f : 0.6f, eb = (iy1 == a.H - 1) ? 1.6f : 0.6f;
            const float cx = (float)a.col0 + a.pc + ((tid & 1) ? (float)ix1 + er : (float)ix0 - el);
            const float cy = (float)a.row0 + a.pc + ((tid & 2) ? (float)iy1 + eb : (float)iy0 - et);
            const float X = h[0] * cx + h[1] * cy + h[2], Y = h[3] * cx + h[4] * cy + h[5], Z = h[6] * cx + h[7] * cy + h[8];
            s_c[buf][tid] = texel_coord<COORD>(X / Z, (float)a.Ws / 2.0f, (float)(a.Ws - 1), a.sx, a.ox);
            s_c[buf][4 + tid] = texel_coord<COORD>(Y / Z, (float)a.Hs / 2.0f, (float)(a.Hs - 1), a.sy, a.oy);
        }
        // (2) sample this pixel on plane d, composite backward, stage (tx,ty,g) in LDS
        float2 tc = make_float2(-1e30f, -1e30f);
        float4 gval = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inimg) {
            Taps tp = make_taps<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy);
            if (tp.covered) {
                float4 o, pre, tv[4];
                const float *src = plane;
                if (a.ablate & 4) {   // measurement only: all taps from a 64 KiB cache-resident window
                    src = a.stack;
#pragma unroll
                    for (int i = 0; i < 4; ++i) tp.idx[i] = tp.idx[i] >= 0 ? (tp.idx[i] & 4095) : -1;
                }
                shade<ORDER, RACT, AACT>(src, tp, o, pre, tv);
                const float q = Gr * o.x + Gg * o.y + Gb * o.z + gA;
                const float w = o.w * Tr;
                P += w * q;
                const float om = 1.0f - o.w;
                const float behind = (om > 1e-12f) ? (S - P) / om : 0.0f;
                gval = make_float4(w * Gr, w * Gg, w * Gb, Tr * q - behind);     // grad wrt activated (c, a)
                Tr *= om;
                if constexpr (ORDER == VL3D_ACT_POST)
                    gval = make_float4(gval.x * act_bwd<RACT>(pre.x, o.x), gval.y * act_bwd<RACT>(pre.y, o.y),
                                       gval.z * act_bwd<RACT>(pre.z, o.z), gval.w * act_bwd<AACT>(pre.w, o.w));
                tc = make_float2(tp.tx, tp.ty);
            }
        }
        s_t[buf][tid] = tc;
        s_g[buf][tid] = gval;
        __syncthreads();   // staging of plane d visible (the other buffer may still be read by slower waves: not touched here)
        // (3) every texel owned by this tile gathers its taps from the 3x3 pixels around its owner pixel
        const float mnx = fminf(fminf(s_c[buf][0], s_c[buf][1]), fminf(s_c[buf][2], s_c[buf][3]));
        const float mxx = fmaxf(fmaxf(s_c[buf][0], s_c[buf][1]), fmaxf(s_c[buf][2], s_c[buf][3]));
        const float mny = fminf(fminf(s_c[buf][4], s_c[buf][5]), fminf(s_c[buf][6], s_c[buf][7]));
        const float mxy = fmaxf(fmaxf(s_c[buf][4], s_c[buf][5]), fmaxf(s_c[buf][6], s_c[buf][7]));
        const int X0 = max(0, (int)floorf(fmaxf(mnx - 0.01f, -2.0f))), Y0 = max(0, (int)floorf(fmaxf(mny - 0.01f, -2.0f)));
        const int X1 = min(a.Ws - 1, (int)floorf(fminf(mxx + 0.01f, (float)a.Ws)) + 1);
        const int Y1 = min(a.Hs - 1, (int)floorf(fminf(mxy + 0.01f, (float)a.Hs)) + 1);
        const int ww = max(0, X1 - X0 + 1), wh = max(0, Y1 - Y0 + 1);
        const float inv_ww = 1.0f / (float)max(ww, 1);
        const float *hi = a.plan + PLAN_HDR + 9 * d;
        if (a.ablate & 1) continue;
        for (int idx = tid; idx < ww * wh; idx += NT) {
            const int wy = (int)(((float)idx + 0.5f) * inv_ww), wx = idx - wy * ww;
            const float tauX = (float)(X0 + wx), tauY = (float)(Y0 + wy);
            float qx, qy;
            owner_pixel(hi, tauX, tauY, a.pc, a.col0, a.row0, qx, qy);
            // owner = nearest FRAME pixel: texels just outside the frame still collect taps of the border pixels
            const float rx = fminf(fmaxf(rintf(qx), 0.0f), (float)(a.W - 1));
            const float ry = fminf(fmaxf(rintf(qy), 0.0f), (float)(a.H - 1));
            if (!(rx >= (float)ix0 && rx <= (float)ix1 && ry >= (float)iy0 && ry <= (float)iy1)) continue;
            const int lc = ((int)ry - ry0) * RW + ((int)rx - rx0);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int li = lc + dy * RW + dx;
                    const float2 c = s_t[buf][li];
                    const float wxx = 1.0f - fabsf(c.x - tauX), wyy = 1.0f - fabsf(c.y - tauY);
                    if (wxx > 0.0f && wyy > 0.0f) {
                        const float4 g = s_g[buf][li];
                        const float w = wxx * wyy;
                        acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
                    }
                }
            const size_t toff = ((size_t)(Y0 + wy) * a.Ws + (X0 + wx)) * 4;
            if constexpr (ORDER == VL3D_ACT_PRE) {   // d act(s_tau)/d s_tau factors out of the tap sum
                const float4 sv = *reinterpret_cast<const float4 *>(plane + toff);
                acc.x *= act_bwd<RACT>(sv.x, act_fwd<RACT>(sv.x)); acc.y *= act_bwd<RACT>(sv.y, act_fwd<RACT>(sv.y));
                acc.z *= act_bwd<RACT>(sv.z, act_fwd<RACT>(sv.z)); acc.w *= act_bwd<AACT>(sv.w, act_fwd<AACT>(sv.w));
            }
            if (!(a.ablate & 2)) *reinterpret_cast<float4 *>(gplane + toff) = acc;
        }
    }
}

// ---- dispatch over the compile-time conventions -------------------------------------------------------
template <int COORD, int BORDER, int ORDER, int RACT, int AACT, int ROWS>
void launch_tile(const RenderArgs &a, hipStream_t s) {
    dim3 grid((a.W + RW - 3) / (RW - 2), (a.H + ROWS - 3) / (ROWS - 2), a.T);
    hipLaunchKernelGGL((render_bwd_tile_k<COORD, BORDER, ORDER, RACT, AACT, ROWS>), grid, dim3(RW * ROWS), 0, s, a);
}

// g_tile_rows: 0 = no tile path for this call, else the ROWS of the tile kernel to launch
thread_local int g_tile_rows = 0;

template <bool BWD, int COORD, int BORDER, int ORDER, int RACT, int AACT>
void launch(const RenderArgs &a, hipStream_t s) {
    dim3 grid((a.W + TILE_X - 1) / TILE_X, (a.H + TILE_Y - 1) / TILE_Y, a.T), block(TILE_X * TILE_Y);
    if constexpr (BWD) {
        if (g_tile_rows) {
            hipLaunchKernelGGL((bwd_plan_k<COORD>), dim3(1), dim3(64), 0, s, a, g_tile_rows, const_cast<float *>(a.plan));
            const size_t n4 = (size_t)a.D * a.T * a.Hs * a.Ws;
            hipLaunchKernelGGL(bwd_fill_zero_if_infeasible_k, dim3(4096), dim3(256), 0, s, reinterpret_cast<float4 *>(a.g_stack), n4, a.plan);
            hipLaunchKernelGGL(bwd_zero_unowned_k, dim3((a.Ws + 63) / 64, (a.Hs + 3) / 4, a.D), dim3(256), 0, s, a);
            if (g_tile_rows == 8) {
                if constexpr (RACT == VL3D_ACT_SIGMOID && AACT == VL3D_ACT_SIGMOID)
                    launch_tile<COORD, BORDER, ORDER, RACT, AACT, 8>(a, s);
                else
                    launch_tile<COORD, BORDER, ORDER, RACT, AACT, 16>(a, s);
            } else {
                launch_tile<COORD, BORDER, ORDER, RACT, AACT, 16>(a, s);
            }
        }
        hipLaunchKernelGGL((render_bwd_k<COORD, BORDER, ORDER, RACT, AACT>), grid, block, 0, s, a);
    } else {
        hipLaunchKernelGGL((render_fwd_k<COORD, BORDER, ORDER, RACT, AACT>), grid, block, 0, s, a);
    }
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

extern "C" int64_t vl3d_render_bwd_scratch_bytes(const vl3d_render_desc *desc) {
    if (!desc || desc->D <= 0) return 0;
    return (int64_t)(PLAN_HDR + 9 * (int64_t)desc->D) * sizeof(float);
}

extern "C" int vl3d_render_bwd(const vl3d_render_desc *desc, const void *stack, const float *homos,
                               const float *rgb, const float *alpha, const float *grad_rgb,
                               const float *grad_alpha, float *grad_stack, void *scratch, int64_t scratch_bytes,
                               vl3d_stream_t stream) {
    int rc = check_desc(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(stack && homos && rgb && alpha && grad_rgb && grad_stack, "null pointer passed to vl3d_render_bwd");
    RenderArgs a = make_args(desc);
    a.stack = (const float *)stack; a.homos = homos;
    a.rgb = const_cast<float *>(rgb); a.alpha = const_cast<float *>(alpha);
    a.g_rgb = grad_rgb; a.g_alpha = grad_alpha; a.g_stack = grad_stack;
    // variant: 0 auto (tile kernel when its on-device plan says feasible, else atomics), 1 force atomics,
    //          2 tile with 8-row regions, 3 tile with 16-row regions
    const bool want_tile = (desc->variant & 0xf) != 1 && scratch != nullptr && scratch_bytes >= vl3d_render_bwd_scratch_bytes(desc);
    a.ablate = (desc->variant >> 4) & 0xf;
    if (want_tile) {
        a.plan = (const float *)scratch;
        g_tile_rows = ((desc->variant & 0xf) == 2) ? 8 : 16;
    } else {
        a.plan = nullptr;
        g_tile_rows = 0;
        const size_t bytes = (size_t)desc->D * desc->T * desc->Hs * desc->Ws * 4 * sizeof(float);
        VL3D_HIP(hipMemsetAsync(grad_stack, 0, bytes, (hipStream_t)stream));
    }
    rc = dispatch<true>(desc, a, (hipStream_t)stream);
    g_tile_rows = 0;
    if (rc != VL3D_OK) return rc;
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}
