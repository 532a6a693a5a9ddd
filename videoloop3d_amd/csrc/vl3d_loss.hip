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
#include <algorithm>
#include <utility>

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
// K3 v2: the same frame-pair-energy formulation fed from a PIXEL-MAJOR copy of the videos.
// video_to_pixel_major_k rewrites [3][T][H][W] (strided) as [H][W][3][TP] (TP = T padded to 4, zero filled) so that
// everything a patch location needs -- all frames of its ps x ps x 3 pixels -- is ps contiguous runs of ps*3*TP floats:
// the staging becomes fully coalesced float4 traffic (v1 gathered 44-byte row segments per frame: 13.0 ms at 720p for
// the ref-view configuration, almost all of it in the staging loads).
__global__ __launch_bounds__(256) void video_to_pixel_major_k(const float *__restrict__ v, int64_t sc, int64_t st, int64_t sr,
                                                              int T, int TP, int H, int W, float *__restrict__ out) {
    __shared__ float tile[64][65];      // [channel-frame chunk][pixel] (+1 pad: conflict-free transposed reads)
    const int row = blockIdx.y, x0 = blockIdx.x * 64, tid = threadIdx.x;
    const int CF = 3 * TP;
    for (int cf0 = 0; cf0 < CF; cf0 += 64) {
        // read: lanes over pixels (coalesced along the row), 4 (c,f) pairs per pass
        for (int j = tid >> 6; j < 64; j += 4) {
            const int cf = cf0 + j, c = cf / TP, f = cf - c * TP, x = x0 + (tid & 63);
            float val = 0.f;
            if (cf < CF && f < T && x < W) val = v[c * sc + f * st + (int64_t)row * sr + x];
            tile[j][tid & 63] = val;
        }
        __syncthreads();
        // write: lanes over (c,f) (contiguous in the pixel-major layout)
        for (int p = tid >> 6; p < 64; p += 4) {
            const int cf = cf0 + (tid & 63), x = x0 + p;
            if (cf < CF && x < W) out[((size_t)row * W + x) * CF + cf] = tile[tid & 63][p];
        }
        __syncthreads();
    }
}

struct NN2Args {
    const float *xt, *yt;   // pixel-major [H][W][3][TxP] / [H][W][3][TyP]
    int32_t *nn;
    int W, ps, pt, stride, stridet, h_o, w_o, n1, n2;
    int TxP, TyP, K, KC;
    int PX, PY;             // v6: frames per pixel of the gram16 copies of x / y (exact)
    int Wy;                 // v6: pixels per row of the gram16 y (== W, or the full frame's width when y is a crop of a prepared clip)
    int Wx, FXm;            // v6: pixels per row / frames per pixel of the gram16 x IN MEMORY (== W / PX, or the untrimmed clip's when the loss prologue wrote it)
    int use_alpha;
    float alpha, dnorm;
    int ablate;   // measurement only: 1 skip epilogue, 2 skip compute, 4 skip staging loads
};

// LDS layout (floats): Xs[KC][TxP] | Ys[KC][TyP] | E[TxP][TyP] | colmin[n2]
__global__ __launch_bounds__(NN_THREADS) void patchnn2_k(NN2Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Xs = smem;
    float *Ys = Xs + (size_t)a.KC * a.TxP;
    float *E = Ys + (size_t)a.KC * a.TyP;
    float *colmin = E + (size_t)a.TxP * a.TyP;
    const int b = blockIdx.x, by = b / a.w_o, bx = b % a.w_o;
    const int r0 = by * a.stride, c0 = bx * a.stride, tid = threadIdx.x;
    const int tiles_j = a.TyP / TJ, ntiles = (a.TxP / TI) * tiles_j;
    const int rowk = a.ps * 3;                      // k = (r*ps + q)*3 + c : one patch row = rowk consecutive k, contiguous in memory
    const int x4 = a.TxP / 4, y4 = a.TyP / 4;
    for (int i = tid; i < a.TxP * a.TyP; i += NN_THREADS) E[i] = 0.f;
    for (int k0 = 0; k0 < a.K; k0 += a.KC) {
        const int kc = min(a.KC, a.K - k0);
        __syncthreads();
        for (int i = tid; i < kc * x4; i += NN_THREADS) {
            const int kk = i / x4, f4 = i - kk * x4, k = k0 + kk, r = k / rowk, rem = k - r * rowk;
            const float4 *src = reinterpret_cast<const float4 *>(a.xt + (((size_t)(r0 + r) * a.W + c0) * 3 + rem) * a.TxP);
            reinterpret_cast<float4 *>(Xs + (size_t)kk * a.TxP)[f4] = src[f4];
        }
        for (int i = tid; i < kc * y4; i += NN_THREADS) {
            const int kk = i / y4, f4 = i - kk * y4, k = k0 + kk, r = k / rowk, rem = k - r * rowk;
            const float4 *src = reinterpret_cast<const float4 *>(a.yt + (((size_t)(r0 + r) * a.W + c0) * 3 + rem) * a.TyP);
            reinterpret_cast<float4 *>(Ys + (size_t)kk * a.TyP)[f4] = src[f4];
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
    // epilogue on all 256 threads: 4 lanes share one row (column) and scan a quarter each, then combine by shuffles.
    // dist = sum_kt E[.][.] / d (utils_vid.py:82-84); normaliser alpha + min_i dist (:133-134); first minimum wins (:141).
    const int sub = tid & 3;
    if (a.use_alpha) {
        for (int j = tid >> 2; j < a.n2; j += NN_THREADS / 4) {
            float m = INFINITY;
            for (int i = sub; i < a.n1; i += 4) {
                float sacc = 0.f;
                for (int kt = 0; kt < a.pt; ++kt) sacc += E[(i * a.stridet + kt) * a.TyP + j * a.stridet + kt];
                m = fminf(m, sacc / a.dnorm);
            }
            m = fminf(m, __shfl_xor(m, 1, 64));
            m = fminf(m, __shfl_xor(m, 2, 64));
            if (sub == 0) colmin[j] = a.alpha + m;
        }
        __syncthreads();
    }
    for (int i0 = 0; i0 < a.n1; i0 += NN_THREADS / 4) {      // uniform trip count: every lane takes part in the shuffles
        const int i = i0 + (tid >> 2);
        const int q = (a.n2 + 3) / 4, j0 = sub * q, j1 = min(a.n2, j0 + q);
        float best = INFINITY;
        int bj = j0;
        bool best_nan = false;
        if (i < a.n1) {
            for (int j = j0; j < j1; ++j) {
                float sacc = 0.f;
                for (int kt = 0; kt < a.pt; ++kt) sacc += E[(i * a.stridet + kt) * a.TyP + j * a.stridet + kt];
                float v = sacc / a.dnorm;
                if (a.use_alpha) v = v / colmin[j];           // utils_vid.py:140
                const bool vn = (v != v);                      // torch.argmin: first minimum, NaN counts as minimal
                if (!best_nan && (vn || v < best)) { best = v; bj = j; best_nan = vn; }
            }
        }
        // combine the 4 quarters in ascending-j order so that ties keep the lowest index
#pragma unroll
        for (int step = 1; step <= 2; step <<= 1) {
            const float ob = __shfl_xor(best, step, 64);
            const int oj = __shfl_xor(bj, step, 64);
            const int on = __shfl_xor((int)best_nan, step, 64);
            const bool other_lower = (sub & step) != 0;      // the partner holds the lower-j range
            bool take;
            if (best_nan || on) take = on && (!best_nan || other_lower);
            else take = (ob < best) || (ob == best && other_lower);
            if (take) { best = ob; bj = oj; best_nan = on != 0; }
        }
        if (i < a.n1 && sub == 0) a.nn[(size_t)b * a.n1 + i] = bj;
    }
}



// colmin + argmin over one location's frame-pair energies E [TxP][TyP] in LDS (utils_vid.py:133-141): with s(i, j) = sum_kt E(i + kt, j + kt),
// the reference's score is (s / d) / (alpha + min_i s / d); first minimum wins, NaN counts as minimal (torch.argmin).  Scaling by a
// positive constant commutes with min and argmin, so the column minimum is taken over the raw sums and each column gets ONE weight
// w_j = (1 / d) / (alpha + min_i s / d): a score is s * w_j -- three LDS reads, two adds and a multiply per frame pair instead of two
// fp32 divisions (which were most of the epilogue's instructions).  Without alpha the raw sums are compared.  The rounding differs
// from the reference's two divisions by an ulp, i.e. only between exact near-ties -- EXCEPT at the column minimum itself: there the
// reference's score is q / (alpha + q) with q = m / d bit-identical in numerator and denominator, which for the shipped ref-view
// alpha = 0 (configs/mpv_base.txt:52) is EXACTLY 1.0 in every column, the smallest score a row can have; a row that is the minimum of
// several columns (certain when n2 > n1) is decided by torch.argmin's first-minimum rule among exact ties.  s * w_j is 1 +- an ulp, so
// an entry equal to its column's minimum takes the tie score t_j = (m / d) / (alpha + m / d) formed with true divisions instead
// (1.0 at alpha = 0; NaN = minimal for m = 0, as 0 / 0 is in the reference).  colw: [3][n2p] = weights | minima | tie scores.
// Whole workgroup, 4 lanes per row / column.
// PREINIT: the caller has set colw[0..n2) to +inf (bit pattern) behind a barrier already
template <int NTHR, bool PREINIT = false>
__device__ __forceinline__ void nn_epilogue(const NN2Args &a, const float *E, float *colw, int n2p, size_t b, int tid, int sub) {
    float *colm = colw + n2p, *colt = colw + 2 * n2p;
    if (a.pt == 3 && a.stridet == 1) {
        // every shipped configuration: three-frame patches at temporal stride 1.  s(i, j..j+3) needs E(i, j..j+3), E(i+1, j+1..j+4) and
        // E(i+2, j+2..j+5): whole 16-byte LDS reads (the rows are 16-byte aligned, TyP is a multiple of 4), the upper halves carried
        // from one group of four columns to the next -- 3/4 of an LDS read per frame pair instead of 3, same additions in the same order.
        const int TyP = a.TyP;
        const float4 *E4 = reinterpret_cast<const float4 *>(E);
        if (a.use_alpha) {
            // column minima: a thread takes four adjacent columns of every nrg-th row; the row groups meet in an LDS integer min
            // (the sums are >= +0 or NaN: their bit patterns order like the values and NaN never wins, as with fminf)
            int *cmi = reinterpret_cast<int *>(colw);
            if (!PREINIT) {
                for (int j = tid; j < a.n2; j += NTHR) cmi[j] = 0x7f800000;
                __syncthreads();
            }
            const int ngrp = TyP / 4, nrg = NTHR / ngrp, jg = tid % ngrp, ig = tid / ngrp;
            if (ig < nrg) {
                float m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
                for (int i = ig; i < a.n1; i += nrg) {
                    const float4 *p = E4 + (i * TyP) / 4 + jg;
                    const float4 c0 = p[0], l1 = p[TyP / 4], h1 = p[TyP / 4 + 1], l2 = p[TyP / 2], h2 = p[TyP / 2 + 1];
                    m0 = fminf(m0, (c0.x + l1.y) + l2.z);
                    m1 = fminf(m1, (c0.y + l1.z) + l2.w);
                    m2 = fminf(m2, (c0.z + l1.w) + h2.x);
                    m3 = fminf(m3, (c0.w + h1.x) + h2.y);
                }
                const int j = jg * 4;
                if (j < a.n2) atomicMin(cmi + j, __float_as_int(m0));
                if (j + 1 < a.n2) atomicMin(cmi + j + 1, __float_as_int(m1));
                if (j + 2 < a.n2) atomicMin(cmi + j + 2, __float_as_int(m2));
                if (j + 3 < a.n2) atomicMin(cmi + j + 3, __float_as_int(m3));
            }
            __syncthreads();
            const float inv_d = 1.0f / a.dnorm;
            for (int j = tid; j < a.n2; j += NTHR) {
                const float m = colw[j], q = m / a.dnorm;
                colm[j] = m;
                colt[j] = q / (a.alpha + q);
                colw[j] = inv_d / (a.alpha + q);
            }
            __syncthreads();
        }
        const int q4 = ((a.n2 + 3) / 4 + 3) & ~3, j0 = sub * q4, j1 = min(a.n2, j0 + q4);     // quarters of whole column groups
        for (int i0 = 0; i0 < a.n1; i0 += NTHR / 4) {      // uniform trip count: every lane takes part in the shuffles
            const int i = i0 + (tid >> 2);
            float best = INFINITY;
            int bj = j0;
            bool best_nan = false;
            if (i < a.n1 && j0 < j1) {
                const float4 *p = E4 + (i * TyP + j0) / 4;
                float4 l1 = p[TyP / 4], l2 = p[TyP / 2];
                for (int jb = j0; jb < j1; jb += 4, ++p) {
                    const float4 c0 = p[0], h1 = p[TyP / 4 + 1], h2 = p[TyP / 2 + 1];
                    float sa[4] = {(c0.x + l1.y) + l2.z, (c0.y + l1.z) + l2.w, (c0.z + l1.w) + h2.x, (c0.w + h1.x) + h2.y};
                    if (a.use_alpha) {
                        const float4 w = *reinterpret_cast<const float4 *>(colw + jb), m = *reinterpret_cast<const float4 *>(colm + jb);
                        const float4 t = *reinterpret_cast<const float4 *>(colt + jb);
                        sa[0] = sa[0] == m.x ? t.x : sa[0] * w.x; sa[1] = sa[1] == m.y ? t.y : sa[1] * w.y;
                        sa[2] = sa[2] == m.z ? t.z : sa[2] * w.z; sa[3] = sa[3] == m.w ? t.w : sa[3] * w.w;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float v = sa[u];
                        const bool vn = (v != v);
                        if (jb + u < j1 && !best_nan && (vn || v < best)) { best = v; bj = jb + u; best_nan = vn; }
                    }
                    l1 = h1; l2 = h2;
                }
            }
            // combine the 4 quarters in ascending-j order so that ties keep the lowest index
#pragma unroll
            for (int step = 1; step <= 2; step <<= 1) {
                const float ob = __shfl_xor(best, step, 64);
                const int oj = __shfl_xor(bj, step, 64);
                const int on = __shfl_xor((int)best_nan, step, 64);
                const bool other_lower = (sub & step) != 0;      // the partner holds the lower-j range
                bool take;
                if (best_nan || on) take = on && (!best_nan || other_lower);
                else take = (ob < best) || (ob == best && other_lower);
                if (take) { best = ob; bj = oj; best_nan = on != 0; }
            }
            if (i < a.n1 && sub == 0) a.nn[b * a.n1 + i] = bj;
        }
        return;
    }
    if (a.use_alpha) {
        const float inv_d = 1.0f / a.dnorm;
        for (int j = tid >> 2; j < a.n2; j += NTHR / 4) {
            float m = INFINITY;
            const float *e0 = E + j * a.stridet;
            // four rows per batch: their LDS reads are in flight together (the workgroup is four waves: nothing else hides the latency)
            int i = sub;
            for (; i + 12 < a.n1; i += 16) {
                float sa[4] = {0.f, 0.f, 0.f, 0.f};
                for (int kt = 0; kt < a.pt; ++kt)
#pragma unroll
                    for (int u = 0; u < 4; ++u) sa[u] += e0[((i + 4 * u) * a.stridet + kt) * a.TyP + kt];
                m = fminf(fminf(m, fminf(sa[0], sa[1])), fminf(sa[2], sa[3]));
            }
            for (; i < a.n1; i += 4) {
                float sacc = 0.f;
                for (int kt = 0; kt < a.pt; ++kt) sacc += e0[(i * a.stridet + kt) * a.TyP + kt];
                m = fminf(m, sacc);
            }
            m = fminf(m, __shfl_xor(m, 1, 64));
            m = fminf(m, __shfl_xor(m, 2, 64));
            if (sub == 0) {
                const float q = m / a.dnorm;
                colm[j] = m;
                colt[j] = q / (a.alpha + q);
                colw[j] = inv_d / (a.alpha + q);
            }
        }
        __syncthreads();
    }
    for (int i0 = 0; i0 < a.n1; i0 += NTHR / 4) {      // uniform trip count: every lane takes part in the shuffles
        const int i = i0 + (tid >> 2);
        const int qn = (a.n2 + 3) / 4, j0 = sub * qn, j1 = min(a.n2, j0 + qn);
        float best = INFINITY;
        int bj = j0;
        bool best_nan = false;
        if (i < a.n1) {
            const float *e0 = E + (i * a.stridet) * a.TyP;
            int j = j0;
            for (; j + 3 < j1; j += 4) {                       // four columns per batch, compared in ascending order
                float sa[4] = {0.f, 0.f, 0.f, 0.f};
                for (int kt = 0; kt < a.pt; ++kt)
#pragma unroll
                    for (int u = 0; u < 4; ++u) sa[u] += e0[kt * a.TyP + (j + u) * a.stridet + kt];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float v = a.use_alpha ? (sa[u] == colm[j + u] ? colt[j + u] : sa[u] * colw[j + u]) : sa[u];
                    const bool vn = (v != v);
                    if (!best_nan && (vn || v < best)) { best = v; bj = j + u; best_nan = vn; }
                }
            }
            for (; j < j1; ++j) {
                float sacc = 0.f;
                for (int kt = 0; kt < a.pt; ++kt) sacc += e0[kt * a.TyP + j * a.stridet + kt];
                const float v = a.use_alpha ? (sacc == colm[j] ? colt[j] : sacc * colw[j]) : sacc;
                const bool vn = (v != v);
                if (!best_nan && (vn || v < best)) { best = v; bj = j; best_nan = vn; }
            }
        }
        // combine the 4 quarters in ascending-j order so that ties keep the lowest index
#pragma unroll
        for (int step = 1; step <= 2; step <<= 1) {
            const float ob = __shfl_xor(best, step, 64);
            const int oj = __shfl_xor(bj, step, 64);
            const int on = __shfl_xor((int)best_nan, step, 64);
            const bool other_lower = (sub & step) != 0;      // the partner holds the lower-j range
            bool take;
            if (best_nan || on) take = on && (!best_nan || other_lower);
            else take = (ob < best) || (ob == best && other_lower);
            if (take) { best = ob; bj = oj; best_nan = on != 0; }
        }
        if (i < a.n1 && sub == 0) a.nn[b * a.n1 + i] = bj;
    }
}

// ---------------------------------------------------------------------------------------------------
// K3 v4: NL = 4 neighbouring patch locations per workgroup, column-sum formulation.
// The 4 locations (by, bx0..bx0+3) share a ps x (ps + 3*stride) pixel region.  Per region row, the row (all columns,
// channels and frames: ONE contiguous run in the pixel-major layout) is staged once; for every region column q the
// frame-pair energy of that column  C = sum_c (x - y)^2  is formed once per thread tile and added to the accumulators of
// the (up to ceil(ps/stride)) locations whose window contains q.  Versus one location per workgroup (v2) this halves the
// VALU work (ps=11, stride=4: 23 columns instead of 44 per row), halves the staging traffic and amortises the per-workgroup
// fixed costs over 4 locations; E lives in registers for the whole K loop (no LDS read-modify-write per chunk).
constexpr int NL4 = 4;

// NTHR: 256 threads for the shipped clips (one 4x4 frame-pair tile per thread), 512 / 1024 for longer ones (cfg4 / cfg5)
template <bool RUNSUM, int NTHR = NN_THREADS>
__global__ __launch_bounds__(NTHR) void patchnn4_k(NN2Args a, int H_unused, int groups_x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int RWc = a.ps + (NL4 - 1) * a.stride;               // region width in pixels
    float *Xs = smem;                                           // [RWc*3][TxP]
    float *Ys = Xs + (size_t)RWc * 3 * a.TxP;                   // [RWc*3][TyP]
    float *E = smem;                                            // [TxP][TyP], one location at a time in the epilogue: ALIASES the
                                                                // staging buffers (dead by then) -> 35 KiB instead of 51 KiB of LDS
                                                                // for the ref-view cfg = 4 workgroups per CU instead of 3
    float *colmin = E + (size_t)a.TxP * a.TyP;
    const int g = blockIdx.x, by = g / groups_x, bx0 = (g % groups_x) * NL4;
    const int r0 = by * a.stride, c0 = bx0 * a.stride, tid = threadIdx.x;
    const int cols = min(RWc, a.W - c0);                        // the last group of a row may be narrower
    const int nloc = min(NL4, a.w_o - bx0);
    const int tiles_j = a.TyP / TJ, ntiles = (a.TxP / TI) * tiles_j;
    const bool has_tile = tid < ntiles;
    // thread -> frame-pair tile: y tiles fastest.  (19 y tiles for the shipped 75-frame clips put slots 16..18 of a ds_read_b128 lane
    // group on the banks of slots 0..2 -- SQ_LDS_BANK_CONFLICT is 0.75-0.85 of the LDS-active cycles, profiles/r02_pmc_summary.txt --
    // but the conflict-free alternative, x tiles fastest (13 <= 16), measured 1-3 % SLOWER in process (profiles/ab_loss.py, variant
    // 0x80: ref 3.99 vs 3.96 ms, other 3.16 vs 3.07 ms): the LDS is not what this kernel waits for.  Kept selectable for A/B.)
    const int tiles_i = a.TxP / TI;
    const bool i_fast = VL3D_ABLATE(a.ablate, 8) && tiles_i <= 16 && tiles_j > 16;
    const int ti = has_tile ? (i_fast ? tid % tiles_i : tid / tiles_j) * TI : 0, tj = has_tile ? (i_fast ? tid / tiles_i : tid % tiles_j) * TJ : 0;
    float acc[NL4][TI * TJ];
#pragma unroll
    for (int l = 0; l < NL4; ++l)
#pragma unroll
        for (int e = 0; e < TI * TJ; ++e) acc[l][e] = 0.f;
    const int x4 = cols * 3 * a.TxP / 4, y4 = cols * 3 * a.TyP / 4;
    for (int r = 0; r < a.ps; ++r) {
        __syncthreads();
        const float4 *xsrc = reinterpret_cast<const float4 *>(a.xt + ((size_t)(r0 + r) * a.W + c0) * 3 * a.TxP);
        const float4 *ysrc = reinterpret_cast<const float4 *>(a.yt + ((size_t)(r0 + r) * a.W + c0) * 3 * a.TyP);
        for (int i = tid; i < x4; i += NTHR) reinterpret_cast<float4 *>(Xs)[i] = xsrc[i];
        for (int i = tid; i < y4; i += NTHR) reinterpret_cast<float4 *>(Ys)[i] = ysrc[i];
        __syncthreads();
        if (has_tile) {
            // running sum R of the column energies along the row; location l's share of this row is R(window end) - R(before
            // window start): 16 adds per column + two snapshots per location and row instead of 16 adds per (column, covering
            // location).  R restarts every row, so the difference loses at most ~log2(23 columns) bits of a 24-bit sum.
            float R[TI * TJ];
#pragma unroll
            for (int e = 0; e < TI * TJ; ++e) R[e] = 0.f;
            for (int q = 0; q < cols; ++q) {
                if constexpr (RUNSUM) {
#pragma unroll
                    for (int l = 0; l < NL4; ++l)
                        if (q == l * a.stride) {               // window of location l starts at this column (uniform)
#pragma unroll
                            for (int e = 0; e < TI * TJ; ++e) acc[l][e] -= R[e];
                        }
                } else {                                       // few covering locations per column: plain per-column energy
#pragma unroll
                    for (int e = 0; e < TI * TJ; ++e) R[e] = 0.f;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float4 xv = *reinterpret_cast<const float4 *>(Xs + (q * 3 + c) * a.TxP + ti);
                    const float4 yv = *reinterpret_cast<const float4 *>(Ys + (q * 3 + c) * a.TyP + tj);
                    const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, ya[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < TJ; ++j) {
                            const float df = xa[i] - ya[j];
                            R[i * TJ + j] = fmaf(df, df, R[i * TJ + j]);
                        }
                }
                if constexpr (RUNSUM) {
#pragma unroll
                    for (int l = 0; l < NL4; ++l)
                        if (q == l * a.stride + a.ps - 1) {    // ... and ends at this one
#pragma unroll
                            for (int e = 0; e < TI * TJ; ++e) acc[l][e] += R[e];
                        }
                } else {
#pragma unroll
                    for (int l = 0; l < NL4; ++l) {
                        const int ql = q - l * a.stride;       // column inside location l's window?
                        if (ql >= 0 && ql < a.ps) {
#pragma unroll
                            for (int e = 0; e < TI * TJ; ++e) acc[l][e] += R[e];
                        }
                    }
                }
            }
        }
    }
    // epilogue, one location at a time through the shared E buffer
    const int sub = tid & 3;
#pragma unroll
    for (int l = 0; l < NL4; ++l) {
        if (l >= nloc) break;                                    // uniform
        __syncthreads();
        if (has_tile) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) E[(ti + i) * a.TyP + tj + j] = fmaxf(acc[l][i * TJ + j], 0.0f);   // (a difference of running sums: >= -rounding)
        }
        __syncthreads();
        const size_t b = (size_t)by * a.w_o + bx0 + l;
        nn_epilogue<NTHR>(a, E, colmin, (a.n2 + 3) & ~3, b, tid, sub);
    }
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// floor(n / d) for 0 <= n < 2^16, 0 < d < 2^12 through the hardware reciprocal: (n + 1/2) / d is at least 1 / (2 d) away from any
// integer, far more than the reciprocal's relative error times the quotient
__device__ __forceinline__ int fdiv_small(int n, int d) { return (int)(((float)n + 0.5f) * __builtin_amdgcn_rcpf((float)d)); }

__device__ __forceinline__ void lds_dma16(const float4 *g, float *lds_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_uniform, 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------------
// K3 v6: the v4 workgroup (NL neighbouring locations, a shared ps x (ps + (NL - 1) stride) region, running sums along the region's columns,
// E through the shared epilogue) with the frame-pair energies on the HALF-PRECISION matrix cores at fp32-class accuracy and the region staged
// by LDS-DMA, one stage of CHC columns x all ps rows in flight behind the one being contracted.  (Rounds 2-4 ran this workgroup on
// v_mfma_f32_16x16x4_f32 -- `patchnn5_k`, the vector rate: 32 cycles per 16x16x4 issue, 2.0-2.1 ms at 720p; removed in round 5, its numbers are
// in DESIGN.)  v_mfma_f32_16x16x32_f16 contracts 32 k in 16-17 cycles (16x the rate, MI355X_MICROARCH.md).  The cross term of  e(ti, tj) = |u(ti)|^2 + |v(tj)|^2 - 2 u(ti).v(tj)  is formed from a two-term
// split of every value, u = hi + lo with hi = f16(u), lo = f16(u - hi) (11 + 11 mantissa bits; f16 subnormals are kept by the conversion
// and by the matrix cores, measured in profiles/microbench/dma_f16.hip), as  hi.hi' + hi.lo' + lo.hi'  accumulated in fp32 (the dropped
// lo.lo' is < 2^-24 of the product: an fp32 rounding).  The two norms are fp32 sums on the side.
// gram16 form (video_to_gram16_k): per pixel and frame ONE 16-byte piece  [h0 h1 | l0 l1 | h2 l2 | nh nl]  -- the split of the three
// channels (x: u = x - 1/2; y: -2 v, v = y - 1/2) and of the norm n = sum_c u_c^2 (y: sum_c v_c^2; 22 bits, below the rounding of the fp32 window
// sum it enters) -- pixel-major [H][W][T], so a region row is one contiguous run for the LDS-DMA (strided lanes cost 5x:
// dma_f16.hip) and any crop origin of a prepared clip is valid.
// One MFMA covers TWO cells (pixels) of a region column for a 16 x 16 frame-pair tile: its four 8-element k blocks (lane >> 4) are
//   kb 0: cell a, A = [h0 h1 h0 h1 h2 h2 0 0]   kb 1: cell a, A = [l0 l1 0 0 l2 0 0 0]   kb 2 / 3: the same for cell b
// against B = the RAW y piece [h0' h1' l0' l1' h2' l2' nh' nl'] in all four: kb 0 gives h.h' + h.l', kb 1 gives l.h', and the norm
// halves meet zeros.  The A blocks come out of the raw x piece in three instructions (a select, a mask, a byte permute with a per-lane
// selector): 9 of 16 k slots carry products, 5.5 issues of 17 cycles per 11-pixel column and tile instead of 11 of 32.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint4 gram16_piece(float a0, float a1, float a2, float s) {
    const float n = fmaf(a2, a2, fmaf(a1, a1, a0 * a0));
    const float u0 = a0 * s, u1 = a1 * s, u2 = a2 * s;
    const _Float16 h0 = (_Float16)u0, h1 = (_Float16)u1, h2 = (_Float16)u2, nh = (_Float16)n;
    const _Float16 l0 = (_Float16)(u0 - (float)h0), l1 = (_Float16)(u1 - (float)h1), l2 = (_Float16)(u2 - (float)h2), nl = (_Float16)(n - (float)nh);
    auto bits = [](_Float16 h) -> unsigned { return (unsigned)__builtin_bit_cast(unsigned short, h); };
    uint4 o;
    o.x = bits(h0) | (bits(h1) << 16);
    o.y = bits(l0) | (bits(l1) << 16);
    o.z = bits(h2) | (bits(l2) << 16);
    o.w = bits(nh) | (bits(nl) << 16);
    return o;
}

template <bool IS_Y>
__global__ __launch_bounds__(256) void video_to_gram16_k(const float *__restrict__ v, int64_t sc, int64_t st, int64_t sr,
                                                         int T, int H, int W, uint4 *__restrict__ out) {
    __shared__ float tile[3][16][65];      // [channel][frame of the group][pixel] (+1 pad: conflict-free transposed reads)
    const int row = blockIdx.y, x0 = blockIdx.x * 64, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = (T + 15) >> 4;
    float pre[12];
    const float *src = v + (int64_t)row * sr + min(x0 + lane, W - 1);
    auto load = [&](int f0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int j = wave + 4 * k, c = j >> 4, f = f0 + (j & 15);
            pre[k] = f < T ? src[c * sc + f * st] : 0.5f;
        }
    };
    load(0);
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int j = wave + 4 * k;
            tile[j >> 4][j & 15][lane] = pre[k] - 0.5f;
        }
        __syncthreads();
        if (g + 1 < G) load((g + 1) * 16);
        // write: 16 lanes = the 16 frames of one pixel = 256 contiguous bytes; 4 pixels per wave and pass
        const int m = lane & 15, f = g * 16 + m;
        for (int p = wave * 4 + (lane >> 4); p < 64; p += 16)
            if (x0 + p < W && f < T)
                out[((size_t)row * W + x0 + p) * T + f] = gram16_piece(tile[0][m][p], tile[1][m][p], tile[2][m][p], IS_Y ? -2.0f : 1.0f);
        __syncthreads();
    }
}

template <int J>
__device__ __forceinline__ void nn6_issue_tile(u32x4_t &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(J * 256) : "memory");
}
template <int N, int... Js>
__device__ __forceinline__ void nn6_issue_tiles(u32x4_t (&b)[N], unsigned addr, std::integer_sequence<int, Js...>) {
    (nn6_issue_tile<Js>(b[Js], addr), ...);
}

// One wave's rows of a location, no alpha, pt = 3, stridet = 1: s(i, j) = E(i, j) + E(i + 1, j + 1) + E(i + 2, j + 2) over the wave's slab
// E [16][EP]; first minimum over j; 4 lanes per row, each a quarter of whole column groups.  The slab holds fmaxf(., 0) of finite sums:
// no NaN can reach the comparison (a NaN accumulator is stored as 0), so it is a plain `<` -- a compare, an index select and a min per
// column instead of the NaN-aware form's seven instructions.  Returns the index (valid in the lanes with sub == 0).
__device__ __forceinline__ int nn_argmin_rows3(const float *E, int EP, int n2, int i, bool active, int sub) {
    const float4 *E4 = reinterpret_cast<const float4 *>(E);
    const int q4 = ((n2 + 3) / 4 + 3) & ~3, j0 = sub * q4, j1 = min(n2, j0 + q4);
    float best = INFINITY;
    int bj = j0;
    if (active && j0 < j1) {
        const float4 *p = E4 + (i * EP + j0) / 4;
        float4 l1 = p[EP / 4], l2 = p[EP / 2];
        int jb = j0;
        for (; jb + 3 < j1; jb += 4, ++p) {
            const float4 c0 = p[0], h1 = p[EP / 4 + 1], h2 = p[EP / 2 + 1];
            const float sa[4] = {(c0.x + l1.y) + l2.z, (c0.y + l1.z) + l2.w, (c0.z + l1.w) + h2.x, (c0.w + h1.x) + h2.y};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bj = sa[u] < best ? jb + u : bj;
                best = fminf(best, sa[u]);
            }
            l1 = h1; l2 = h2;
        }
        if (jb < j1) {                                            // the quarter's last, partial group
            const float4 c0 = p[0], h1 = p[EP / 4 + 1], h2 = p[EP / 2 + 1];
            const float sa[4] = {(c0.x + l1.y) + l2.z, (c0.y + l1.z) + l2.w, (c0.z + l1.w) + h2.x, (c0.w + h1.x) + h2.y};
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (jb + u < j1) {
                    bj = sa[u] < best ? jb + u : bj;
                    best = fminf(best, sa[u]);
                }
        }
    }
#pragma unroll
    for (int step = 1; step <= 2; step <<= 1) {      // combine the 4 quarters in ascending-j order so that ties keep the lowest index
        const float ob = __shfl_xor(best, step, 64);
        const int oj = __shfl_xor(bj, step, 64);
        const bool other_lower = (sub & step) != 0;
        const bool take = (ob < best) || (ob == best && other_lower);
        if (take) { best = ob; bj = oj; }
    }
    return bj;
}

constexpr int NN6_KX = 3;
constexpr int nn6_ky(int tyt) { return tyt == 5 ? 4 : (tyt == 8 ? 6 : 7); }

// NN2Args: PX / PY = frames per pixel of the gram16 x / y (exact, no padding); xt / yt = the gram16 buffers (16-byte pieces).
// One operand register set: a second one (the reads of chunk c + 1 in flight during the MFMAs of chunk c) takes 202 registers = two waves
// per SIMD and measured 1.76 / 1.93 ms against 1.54 / 1.47 ms with three (720p, ref / other cfg): the other waves hide the reads better.
// XS: first x frame of wave w is XS * w.  16: the waves' tiles abut, the epilogue runs workgroup-wide through the shared E.  14: the
// tiles overlap by pt - 1 = 2 frames, so wave w holds every frame pair of its 14 patches: each wave finishes its own rows through a
// private 16-row slab, no workgroup barrier between the locations (no alpha, pt = 3, stridet = 1 only: the column minima of the alpha
// path need all rows -- a two-phase per-wave form of it, minima through an LDS integer min behind one barrier and the slabs written twice,
// measured 1.46 against 1.41 ms at 720p and was not kept).
#ifdef VL3D_NN6_FLAT_NORMS
#define VL3D_NN6_NSTEP 0u
#else
#define VL3D_NN6_NSTEP nstep2
#endif
template <int TYT, int NL, int NW, int XS>
__global__ __launch_bounds__(64 * NW, TYT == 5 ? 3 : 2) void patchnn6_k(NN2Args a, int groups_x, int CHC) {
    constexpr int NTHR = 64 * NW, NXT = 16 * NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KX = NN6_KX, KY = nn6_ky(TYT);                // DMA pieces per wave and stage at most (their source offsets live in registers)
    const int RWc = a.ps + (NL - 1) * a.stride;                 // region width in pixels
    const int FX = a.PX, FY = a.PY;
    const int xs4 = CHC * a.ps * FX, ys4 = CHC * a.ps * FY;       // 16-byte pieces per stage and part
    const int ybase = xs4 * 4, bufF = (xs4 + ys4) * 4;            // (floats)
    const int n2p = (a.n2 + 3) & ~3;
    const int EP = XS == 16 ? a.TyP : (((a.TyP + 3) & ~7) + 4);  // slab row pitch: 4 EP = 16 mod 32 banks (the two lane groups of a store never meet)
    float *E = smem;                                            // XS 16: [TxP][TyP], one location at a time; XS 14: NW slabs [17][EP]: aliases the staging
    float *colw = E + (XS == 16 ? (size_t)a.TxP * a.TyP : (size_t)NW * 17 * EP);
    float *sy = colw + NL * 3 * n2p;                            // [NL][TyP] y norms of the locations   (colw: [NL][3][n2p])
    float *sx = sy + NL * a.TyP;                                // [NL][NXT] x norms
    const int g = blockIdx.x, by = g / groups_x, bx0 = (g % groups_x) * NL;
    const int r0 = by * a.stride, c0 = bx0 * a.stride, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cols = min(RWc, a.W - c0);
    const int nloc = min(NL, a.w_o - bx0);
    f32x4_t acc[NL][TYT], R[TYT];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int j = 0; j < TYT; ++j) acc[l][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TYT; ++j) R[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // norms on the side: thread tid < FX sums frame tid of x, thread NXT + f frame f of y (one LDS word + one dot2 per cell, in the MFMAs' shadow)
    const bool xside = tid < FX, yside = tid >= NXT && tid - NXT < FY;
    float an[NL], Rn = 0.f;
#pragma unroll
    for (int l = 0; l < NL; ++l) an[l] = 0.f;
    int offx[KX], offy[KY];
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        const int idx = (wave + NW * k) * 64 + lane, cc = fdiv_small(idx, a.ps * FX), rem = idx - cc * a.ps * FX, r = fdiv_small(rem, FX);
        offx[k] = idx < xs4 ? ((r * a.Wx + cc) * a.FXm + (rem - r * FX)) | (cc << 24) : -1;
    }
#pragma unroll
    for (int k = 0; k < KY; ++k) {
        const int idx = (wave + NW * k) * 64 + lane, cc = fdiv_small(idx, a.ps * FY), rem = idx - cc * a.ps * FY, r = fdiv_small(rem, FY);
        offy[k] = idx < ys4 ? ((r * a.Wy + cc) * FY + (rem - r * FY)) | (cc << 24) : -1;
    }
    const int S = (cols + CHC - 1) / CHC;                       // stages
    auto issue = [&](int st) {
        const int q0 = st * CHC, nc = min(CHC, cols - q0);
        float *dst = smem + (st & 1) * bufF;
        const float4 *xsrc = reinterpret_cast<const float4 *>(a.xt) + ((size_t)r0 * a.Wx + c0 + q0) * a.FXm;
        const float4 *ysrc = reinterpret_cast<const float4 *>(a.yt) + ((size_t)r0 * a.Wy + c0 + q0) * FY;
#pragma unroll
        for (int k = 0; k < KX; ++k)
            if (offx[k] >= 0 && (offx[k] >> 24) < nc) lds_dma16(xsrc + (offx[k] & 0xffffff), dst + (wave + NW * k) * 256);
#pragma unroll
        for (int k = 0; k < KY; ++k)
            if (offy[k] >= 0 && (offy[k] >> 24) < nc) lds_dma16(ysrc + (offy[k] & 0xffffff), dst + ybase + (wave + NW * k) * 256);
    };
    // operand fragments: lane = frame (lane & 15) + 16 * k block; k blocks 0 / 1 read cell a of the chunk, 2 / 3 cell b -- 16 lanes x 16 bytes
    // of one pixel's frames are 256 contiguous bytes: conflict-free ds_read_b128
    const int kb = lane >> 4;
    const bool lo_blk = (kb & 1) != 0;
    const unsigned ymask = lo_blk ? 0u : 0xffffffffu;           // A.y = (h0 h1) in the h block, 0 in the l block
    const unsigned zsel = lo_blk ? 0x0c0c0302u : 0x01000100u;   // A.z = (h2 h2) / (l2 0) out of the piece's (h2 l2)
    const unsigned cellx = (unsigned)FX * 16u, celly = (unsigned)FY * 16u;
    const int nch = (a.ps + 1) >> 1;
    const bool odd = (a.ps & 1) != 0;
    // per-lane byte steps from chunk to chunk; into the LAST chunk of an odd column the lanes of cell b step one cell only (they re-read the
    // column's last cell, their A block is zeroed)
    const unsigned xstep2 = 2u * cellx, ystep2 = 2u * celly;
    const unsigned xstepL = (odd && kb >= 2) ? cellx : xstep2, ystepL = (odd && kb >= 2) ? celly : ystep2;
    const unsigned nstep2 = xside ? xstep2 : (yside ? ystep2 : 0u);
    const f16x2_t ones = {(_Float16)1.0f, (_Float16)1.0f};
    if (!VL3D_ABLATE(a.ablate, 4)) issue(0);
    for (int st = 0; st < S; ++st) {
        __syncthreads();                                         // stage st has landed; everyone is done with the other buffer
        if (st + 1 < S && !VL3D_ABLATE(a.ablate, 4)) issue(st + 1);
        if VL3D_ABLATE(a.ablate, 2) continue;
        const int q0 = st * CHC, nc = min(CHC, cols - q0);
        const float *buf = smem + (st & 1) * bufF;
        for (int cc = 0; cc < nc; ++cc) {
            const int q = q0 + cc;
#pragma unroll
            for (int l = 0; l < NL; ++l)
                if (q == l * a.stride) {      // window of location l starts at this column (uniform): acc = R(end) - R(before start)
#pragma unroll
                    for (int j = 0; j < TYT; ++j) acc[l][j] -= R[j];
                    an[l] -= Rn;
                }
            const bool one_cell = a.ps == 1;
            unsigned xad = (unsigned)reinterpret_cast<uintptr_t>(buf) + (unsigned)(cc * a.ps * FX + XS * wave + (lane & 15)) * 16u + ((kb >= 2 && !one_cell) ? cellx : 0u);
            unsigned yad = (unsigned)reinterpret_cast<uintptr_t>(buf + ybase) + (unsigned)(cc * a.ps * FY + (lane & 15)) * 16u + ((kb >= 2 && !one_cell) ? celly : 0u);
            unsigned nad0 = xside ? (unsigned)reinterpret_cast<uintptr_t>(buf) + (unsigned)(cc * a.ps * FX + tid) * 16u + 12u
                                  : (yside ? (unsigned)reinterpret_cast<uintptr_t>(buf + ybase) + (unsigned)(cc * a.ps * FY + tid - NXT) * 16u + 12u
                                           : (unsigned)reinterpret_cast<uintptr_t>(buf));
#ifdef VL3D_NN6_FLAT_NORMS      // measurement build only (WRONG norms): every lane reads one and the same word -- prices the bank conflicts of the side threads' reads
            nad0 = (unsigned)reinterpret_cast<uintptr_t>(buf);
#endif
            unsigned nad1 = nad0 + (VL3D_NN6_NSTEP >> 1);
            // The reads and their wait are asm statements fenced by scheduling barriers, or hipcc folds them back next to their use.
            {
                u32x4_t a0, b0[TYT];
                unsigned n00, n01;
                constexpr auto tiles = std::make_integer_sequence<int, TYT>{};
#define VL3D_NN6_ISSUE(A, N0, N1, B)                                                                      \
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5"                        \
                 : "=&v"(A), "=&v"(N0), "=&v"(N1) : "v"(xad), "v"(nad0), "v"(nad1) : "memory");           \
    nn6_issue_tiles(B, yad, tiles);                                                                       \
    __builtin_amdgcn_sched_barrier(0)
#define VL3D_NN6_WAIT(A, N0, N1, B)                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A), "+v"(N0), "+v"(N1)::"memory");                         \
    _Pragma("unroll") for (int j = 0; j < TYT; ++j) asm volatile("" : "+v"(B[j])::"memory")
#define VL3D_NN6_NEXT(C)      /* addresses of chunk C + 1 */                                             \
    {                                                                                                     \
        const bool last_ = (C) + 2 >= nch;                                                                \
        xad += last_ ? xstepL : xstep2; yad += last_ ? ystepL : ystep2;                                   \
        nad0 += VL3D_NN6_NSTEP; nad1 += VL3D_NN6_NSTEP;                                                   \
    }
#define VL3D_NN6_MMA(C, A, N0, N1, B)                                                                     \
    {                                                                                                     \
        u32x4_t s_;                                                                                       \
        s_.x = lo_blk ? A.y : A.x;                                                                        \
        s_.y = A.x & ymask;                                                                               \
        s_.z = __builtin_amdgcn_perm(A.z, A.z, zsel);                                                     \
        s_.w = 0u;                                                                                        \
        const bool odd_ = 2 * (C) + 1 >= a.ps;            /* (uniform) the chunk's second cell does not exist */ \
        if (odd_ && kb >= 2) { s_.x = 0u; s_.y = 0u; s_.z = 0u; }                                         \
        const f16x8_t af_ = __builtin_bit_cast(f16x8_t, s_);                                              \
        _Pragma("unroll") for (int j = 0; j < TYT; ++j)                                                   \
            R[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_, __builtin_bit_cast(f16x8_t, B[j]), R[j], 0, 0, 0); \
        Rn = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, N0), ones, Rn, false);                    \
        if (!odd_) Rn = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, N1), ones, Rn, false);         \
    }
                for (int c = 0; c < nch; ++c) {
                    VL3D_NN6_ISSUE(a0, n00, n01, b0);
                    VL3D_NN6_WAIT(a0, n00, n01, b0);
                    VL3D_NN6_MMA(c, a0, n00, n01, b0);
                    VL3D_NN6_NEXT(c);
                }
#undef VL3D_NN6_ISSUE
#undef VL3D_NN6_WAIT
#undef VL3D_NN6_NEXT
#undef VL3D_NN6_MMA
            }
#pragma unroll
            for (int l = 0; l < NL; ++l)
                if (q == l * a.stride + a.ps - 1) {   // ... and ends at this one
#pragma unroll
                    for (int j = 0; j < TYT; ++j) acc[l][j] += R[j];
                    an[l] += Rn;
                }
        }
    }
    // epilogue.  C/D layout of a 16x16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg
    const int sub = tid & 3;
    __syncthreads();                                             // the staging buffers are dead from here on
    if (xside) {
#pragma unroll
        for (int l = 0; l < NL; ++l) sx[l * NXT + tid] = an[l];
    } else if (tid < NXT) {
#pragma unroll
        for (int l = 0; l < NL; ++l) sx[l * NXT + tid] = 0.f;
    }
    if (tid >= NXT && tid - NXT < a.TyP) {
#pragma unroll
        for (int l = 0; l < NL; ++l) sy[l * a.TyP + tid - NXT] = yside ? an[l] : 0.f;
    }
    if constexpr (XS == 16) {
        // one location at a time through the shared E buffer, the whole workgroup on it
        for (int j = tid; j < NL * n2p; j += NTHR) reinterpret_cast<int *>(colw)[(j / n2p) * 3 * n2p + j % n2p] = 0x7f800000;     // column minima start at +inf
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (l >= nloc) break;                                    // uniform
            __syncthreads();                                         // sx / sy written / the previous location's E read
            float syv[TYT];
#pragma unroll
            for (int j = 0; j < TYT; ++j) syv[j] = sy[l * a.TyP + min(j * 16 + (lane & 15), a.TyP - 1)];
            const float4 sxv = *reinterpret_cast<const float4 *>(sx + l * NXT + wave * 16 + (lane >> 4) * 4);
            const float sxa[4] = {sxv.x, sxv.y, sxv.z, sxv.w};
#pragma unroll
            for (int j = 0; j < TYT; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int row = wave * 16 + (lane >> 4) * 4 + rr, col = j * 16 + (lane & 15);
                    if (row < a.TxP && col < a.TyP) E[row * a.TyP + col] = fmaxf(acc[l][j][rr] + (sxa[rr] + syv[j]), 0.0f);
                }
            __syncthreads();
            const size_t b = (size_t)by * a.w_o + bx0 + l;
            if VL3D_ABLATE(a.ablate, 1) { if (tid < a.n1) a.nn[b * a.n1 + tid] = 0; continue; }
            nn_epilogue<NTHR, true>(a, E, colw + l * 3 * n2p, n2p, b, tid, sub);
        }
    } else {
        // every wave finishes the 14 patches whose frames it holds, through its own slab: no workgroup barrier between the locations
        __syncthreads();                                         // sx / sy written
        float *Ew = E + wave * 17 * EP;
        const int i0 = XS * wave, nrow = min(XS, a.n1 - i0);      // (may be <= 0: a wave past the last patch)
        const int irow = lane >> 2;
        float sxa[NL][4];
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) sxa[l][rr] = sx[l * NXT + min(i0 + (lane >> 4) * 4 + rr, NXT - 1)];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (l >= nloc) break;                                    // uniform
            float syv[TYT];
#pragma unroll
            for (int j = 0; j < TYT; ++j) syv[j] = sy[l * a.TyP + min(j * 16 + (lane & 15), a.TyP - 1)];
#pragma unroll
            for (int j = 0; j < TYT; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int col = j * 16 + (lane & 15);
                    if (col < a.TyP) Ew[((lane >> 4) * 4 + rr) * EP + col] = fmaxf(acc[l][j][rr] + (sxa[l][rr] + syv[j]), 0.0f);
                }
            const size_t b = (size_t)by * a.w_o + bx0 + l;
            if VL3D_ABLATE(a.ablate, 1) { if (lane < nrow) a.nn[b * a.n1 + i0 + lane] = 0; continue; }
            const int bj = nn_argmin_rows3(Ew, EP, a.n2, irow, irow < nrow, sub);
            if (irow < nrow && sub == 0) a.nn[b * a.n1 + i0 + irow] = bj;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// get_NN_indices_low_memory on MATERIALISED patches (utils_vid.py:122-142; used by evaluations/NNMSE.py:45-56):
// X [B,n1,d], Y [B,n2,d] dense.  One workgroup per batch entry b; dist[i][j] = sum_k (X[b,i,k]-Y[b,j,k])^2 / d kept in LDS.
// API-parity kernel (the training loss never materialises patches); plain VALU, K staged in chunks.
__global__ __launch_bounds__(256) void nn_vectors_k(const float *__restrict__ X, const float *__restrict__ Y, int n1, int n2, int d,
                                                    int use_alpha, float alpha, int64_t *__restrict__ nn, int KC) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Xs = smem;                       // [n1][KC+1]
    float *Ys = Xs + (size_t)n1 * (KC + 1); // [n2][KC+1]
    float *E = Ys + (size_t)n2 * (KC + 1);  // [n1][n2]
    float *colmin = E + (size_t)n1 * n2;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *xb = X + (size_t)b * n1 * d, *yb = Y + (size_t)b * n2 * d;
    for (int i = tid; i < n1 * n2; i += 256) E[i] = 0.f;
    for (int k0 = 0; k0 < d; k0 += KC) {
        const int kc = min(KC, d - k0);
        __syncthreads();
        for (int i = tid; i < n1 * kc; i += 256) { const int f = i / kc, k = i - f * kc; Xs[f * (KC + 1) + k] = xb[(size_t)f * d + k0 + k]; }
        for (int i = tid; i < n2 * kc; i += 256) { const int f = i / kc, k = i - f * kc; Ys[f * (KC + 1) + k] = yb[(size_t)f * d + k0 + k]; }
        __syncthreads();
        for (int p = tid; p < n1 * n2; p += 256) {
            const int i = p / n2, j = p - i * n2;
            float acc = 0.f;
            for (int k = 0; k < kc; ++k) { const float df = Xs[i * (KC + 1) + k] - Ys[j * (KC + 1) + k]; acc += df * df; }
            E[p] += acc;
        }
    }
    __syncthreads();
    const float dn = (float)d;
    if (use_alpha) {
        for (int j = tid; j < n2; j += 256) {
            float m = INFINITY;
            for (int i = 0; i < n1; ++i) m = fminf(m, E[i * n2 + j] / dn);
            colmin[j] = alpha + m;
        }
        __syncthreads();
    }
    for (int i = tid; i < n1; i += 256) {
        float best = INFINITY;
        int bj = 0;
        bool best_nan = false;
        for (int j = 0; j < n2; ++j) {
            float v = E[i * n2 + j] / dn;
            if (use_alpha) v = v / colmin[j];
            const bool vn = (v != v);
            if (!best_nan && (vn || v < best)) { best = v; bj = j; best_nan = vn; }
        }
        nn[(size_t)b * n1 + i] = bj;
    }
}

// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// robust loss (utils_vid.py:10-26)
struct Rho {
    int kind;      // 0 mse, 1 abs, 2 log1p (rou==0), 3 quadratic (rou==2), 4 general
    float scale, b, d, coef;
    float inv_scale, inv_b;      // host reciprocals for the fused kernel (the two IEEE quotients by constants were ~20 instructions per voxel)
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

// value and derivative together (the fused fold kernel): the general Barron case shares one log2/exp2 pair,
// (s^2/b+1)^(d/2) = exp2(d/2 * log2(u)) and the derivative's power is that over u -- instead of two powf calls
__device__ __forceinline__ void rho_fg(const Rho &r, float e, float &f, float &g) {
    if (r.kind == 4) {
        const float s = e * r.inv_scale;
        const float u = fmaf(s * s, r.inv_b, 1.0f);
        const float p = __builtin_amdgcn_exp2f(0.5f * r.d * __builtin_amdgcn_logf(u));
        f = r.coef * (p - 1.0f) * (r.scale * 10.0f);
        g = 10.0f * s * p * __builtin_amdgcn_rcpf(u);
    } else {
        f = rho_f(r, e);
        g = rho_g(r, e);
    }
}

struct FoldArgs {
    const float *y;
    const int32_t *nn;
    float *sum, *weight;
    int Tx, H, W, ps, pt, stride, stridet, h_o, w_o, n1;
    int64_t y_sc, y_st, y_sr;
    int normalize;
    // fused robust loss (vl3d_vote_fold_robust): x (strided like the loss desc), per-element gradient rho'(x - y2x) * gscale and
    // the loss sum, all optional (x == nullptr: plain fold)
    const float *x;
    int64_t x_sc, x_st, x_sr;
    Rho rho;
    float gscale;
    float *gx;
    int64_t gx_sc, gx_st, gx_sr;      // strides of gx (channel, frame, row), unit column stride
    double *loss_sum;
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


// vote-fold, LDS-staged: one workgroup = a FT_W x FT_H pixel tile of ONE channel.  The y columns of the tile (all Ty
// frames) and the NN indices of every patch location covering the tile are staged in LDS once; each thread then produces
// its pixel's whole temporal column (all Tx frames) from LDS: the gather of utils_vid.py:217 + FoldNd (:218-227) without
// any scattered global access.  (v1 -- one thread per voxel reading y and nn through L2 -- measured 12.3 ms at 720p.)
// Tile shapes (FT_W x FT_H pixels, FT_NT threads = pixels x FT_G temporal groups): see fold_shapes[] below.

// NB > 0 (patch-major form only): a pixel is covered by at most NB x NB patch locations (NB = ceil(ps / stride): 3 for the shipped ref view,
// 2 for the other views).  The covering loops then have a FIXED trip count -- a location that does not exist reads its index from a
// row of the table that points at three all-zero frames behind the staged column, i.e. votes 0 -- so a row of NB index reads is issued
// together and its 3 NB vote reads behind them together, instead of one dependent index -> vote chain per location under dynamic
// loop bounds (the PMC pass had the kernel waiting in 66 % of its wave-cycles and VALU bound on the address arithmetic,
// profiles/r03_pmc_summary.txt).  y + 0 == y: same sums, same order.
template <int FT_W, int FT_H, int FT_NT, int NB = 0>
__global__ __launch_bounds__(FT_NT) void vote_fold_lds_k(FoldArgs a, int Ty) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NP = FT_W * FT_H, FT_G = FT_NT / NP, NT = FT_NT;
    float *ys = smem;                                            // [Ty + 3][NP]: the tile's y columns, then three zero frames
    int *nns = reinterpret_cast<int *>(smem + (size_t)(Ty + 3) * NP);   // [nby][nbx][n1], then n1 indices of the zero frames
    const int tid = threadIdx.x, pix = tid % NP, grp = tid / NP, lx = pix % FT_W, ly = pix / FT_W;
    const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H, c = blockIdx.z;
    const int xi = x0 + lx, eta = y0 + ly;
    // patch locations covering any pixel of the tile
    const int y1 = min(y0 + FT_H, a.H) - 1, x1 = min(x0 + FT_W, a.W) - 1;
    const int tby0 = max(0, (y0 - a.ps + a.stride) / a.stride), tby1 = min(a.h_o - 1, y1 / a.stride);
    const int tbx0 = max(0, (x0 - a.ps + a.stride) / a.stride), tbx1 = min(a.w_o - 1, x1 / a.stride);
    const int nby = tby1 - tby0 + 1, nbx = tbx1 - tbx0 + 1;
    // stage y[c, :, tile] (row segments of FT_W floats) and the nn indices
    const bool inb = (xi < a.W) && (eta < a.H);
    const float *ysrc = a.y + (int64_t)c * a.y_sc + (int64_t)min(eta, a.H - 1) * a.y_sr + min(xi, a.W - 1);
    // Each temporal group owns a contiguous run of frames (see below).  The x values the fused loss needs do not depend on the
    // staging: the first four of this thread's column are requested BEFORE it and the rest stay four frames ahead of their use, so
    // the column's global reads are never exposed one by one between the LDS chains.
    const bool slide = (a.pt == 3 && a.stridet == 1);
    const int chunk = (a.Tx + FT_G - 1) / FT_G, t0 = slide ? grp * chunk : grp, t1 = slide ? min(a.Tx, t0 + chunk) : a.Tx;
    const int tstep = slide ? 1 : FT_G;
    const float *xsrc = a.x ? a.x + (int64_t)c * a.x_sc + (int64_t)min(eta, a.H - 1) * a.x_sr + min(xi, a.W - 1) : nullptr;
    // (the prefetch pointer walks by a frame step: no 64-bit multiply per load)
    const int64_t xstep = (int64_t)tstep * a.x_st;
    const float *xnext = xsrc ? xsrc + (int64_t)t0 * a.x_st : nullptr;
    int tnext = t0;
    auto xload = [&]() -> float {
#if defined(VL3D_FOLD_ABLATE) && (VL3D_FOLD_ABLATE & 4)      // measurement build only (WRONG values): every fourth lane moves its four pixels' x as ONE 16-byte load
        float v = 0.f;
        if (xnext && tnext < t1 && (tid & 3) == 0 && xi + 3 < a.W) {
            const float4 q = *reinterpret_cast<const float4 *>(reinterpret_cast<uintptr_t>(xnext) & ~(uintptr_t)15);
            v = q.x + q.y + q.z + q.w;
        }
#else
        const float v = (xnext && tnext < t1) ? *xnext : 0.f;
#endif
        xnext += xstep; tnext += tstep;
        return v;
    };
    // queue depth: measured (profiles/fold_time.py, round 6): with votes and staging ablated the kernel still took 0.77 of its 1.0 ms -- its x / gx /
    // y2x streams run one 256-byte row segment per wave and request, so the bytes in flight (requests ahead x 24 waves per CU) set the rate
    // (same box, 720p, ms: other views 0.884 / 0.830 / 0.792 / 0.944 at 4 / 8 / 13 / 16 requests ahead; the ref view's NB = 3 instantiation
    // sits at its register budget for three workgroups per CU and loses with every deeper queue: 0.898 / 0.913 / 1.038 / 1.056)
#ifdef VL3D_FOLD_XQ
    constexpr int XQ = VL3D_FOLD_XQ;
#else
    constexpr int XQ = NB >= 3 ? 4 : 12;
#endif
    float xq[XQ];
#pragma unroll
    for (int k = 0; k < XQ; ++k) xq[k] = xload();
    {   // (source pointer and LDS slot walk by a group step: no 64-bit multiply per frame)
        const float *yf = ysrc + (int64_t)grp * a.y_st;
        const int64_t ystep = (int64_t)FT_G * a.y_st;
        float *yd = ys + grp * NP + pix;
#pragma unroll 8
#if defined(VL3D_FOLD_ABLATE) && (VL3D_FOLD_ABLATE & 2)      // measurement build only: no staging loads
        for (int f = grp; f < Ty; f += FT_G, yd += FT_G * NP) *yd = 0.25f;
#else
        for (int f = grp; f < Ty; f += FT_G, yf += ystep, yd += FT_G * NP) *yd = *yf;
#endif
    }
    // the covering locations' index rows: a run of n1 consecutive ints per location -- rows of the tile's locations are contiguous in nn
    // along bx, so a (by) row of the table is ONE contiguous run of nbx * n1 ints: no division per element
    {
        const int run = nbx * a.n1, mul = slide ? NP : 1;
        for (int by = 0; by < nby; ++by) {
            const int32_t *src = a.nn + ((size_t)(tby0 + by) * a.w_o + tbx0) * a.n1;
            int *dst = nns + by * run;
            for (int i = tid; i < run; i += NT) dst[i] = src[i] * mul;
        }
        for (int i = tid; i < 3 * NP; i += NT) ys[Ty * NP + i] = 0.f;
        for (int i = tid; i < a.n1; i += NT) nns[nby * run + i] = Ty * mul;
    }
    __syncthreads();
    float lacc = 0.f;
    if (inb) {
    const int by_hi = min(a.h_o - 1, eta / a.stride), by_lo = max(0, (eta - a.ps + a.stride) / a.stride);
    const int bx_hi = min(a.w_o - 1, xi / a.stride), bx_lo = max(0, (xi - a.ps + a.stride) / a.stride);
    const int npatch = (by_hi - by_lo + 1) * (bx_hi - bx_lo + 1);
    const size_t cs = (size_t)a.Tx * a.H * a.W, fs = (size_t)a.H * a.W;
    float *out = a.sum + (size_t)c * cs + (size_t)eta * a.W + xi;      // (a.sum == nullptr, fused loss only: y2x / weight stay in registers)
    float *wout = a.weight + (size_t)eta * a.W + xi;
    const int *nn0 = nns + ((by_lo - tby0) * nbx + (bx_lo - tbx0)) * a.n1;
    const float *ysp = ys + pix;
    float *gxp = a.x ? a.gx + (int64_t)c * a.gx_sc + (int64_t)t0 * a.gx_st + (int64_t)eta * a.gx_sr + xi : nullptr;      // walks by a frame step
    const int64_t gxstep = (int64_t)tstep * a.gx_st;
    // Each temporal group owns a contiguous run of frames.  With pt == 3 and stridet == 1 (every shipped configuration) the
    // votes are accumulated patch-major: patch i votes for frames i, i+1, i+2, so its NN index is read from LDS ONCE and the
    // three frame sums slide through registers (w0 = frame i, complete after patch i) -- a third of the index reads and of the
    // address arithmetic of the frame-major form below, and the same summation order per frame (kt = 2, 1, 0 -> patches i-2, i-1, i).
    float w0 = 0.f, w1 = 0.f, w2 = 0.f;
    constexpr int NBB = NB > 0 ? NB * NB : 1;
    int loc_off[NBB];            // NB > 0: index rows of the covering locations relative to nn0 (the zero-frame row for one that does not exist)
    if constexpr (NB > 0) {
        const int zero_row = (int)(nns + nby * nbx * a.n1 - nn0);
#pragma unroll
        for (int k = 0; k < NBB; ++k) {
            const int by = k / NB, bx = k % NB;
            const bool ok = by <= by_hi - by_lo && bx <= bx_hi - bx_lo;
            loc_off[k] = ok ? (by * nbx + bx) * a.n1 : zero_row;
        }
    }
    for (int tau = slide ? t0 - 2 : t0; tau < t1; tau += tstep) {
        float s = 0.f;
        int cnt = 0;
        if (slide) {
            const int i = tau;                                     // patch index == first frame it votes for
#if defined(VL3D_FOLD_ABLATE) && (VL3D_FOLD_ABLATE & 1)      // measurement build only: no votes
            if (false) {
#else
            if (i >= 0 && i < a.n1) {
#endif
                if constexpr (NB > 0) {
                    // one row of locations at a time: NB index reads together, then their 3 NB vote reads together (all NB x NB at once
                    // needed 102 registers for NB = 3 -- two workgroups per CU instead of three)
#pragma unroll
                    for (int r = 0; r < NB; ++r) {
                        int idx[NB];
#pragma unroll
                        for (int q = 0; q < NB; ++q) idx[q] = nn0[loc_off[r * NB + q] + i];
#pragma unroll
                        for (int q = 0; q < NB; ++q) {
                            const float *yp = ysp + idx[q];
                            w0 += yp[0]; w1 += yp[NP]; w2 += yp[2 * NP];
                        }
                        if (NB > 2 && r + 1 < NB) __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                const int *nrow = nn0 + i;
                for (int by = by_lo; by <= by_hi; ++by, nrow += nbx * a.n1) {
                    const int *np = nrow;
                    for (int bx = bx_lo; bx <= bx_hi; ++bx, np += a.n1) {
                        const float *yp = ysp + *np;                    // staged as index * NP
                        w0 += yp[0]; w1 += yp[NP]; w2 += yp[2 * NP];
                    }
                }
                }
            }
            s = w0; w0 = w1; w1 = w2; w2 = 0.f;
            if (tau < t0) continue;
            cnt = npatch * (min(tau, a.n1 - 1) - max(tau - 2, 0) + 1);
        } else {
        for (int kt = 0; kt < a.pt; ++kt) {
            const int ts = tau - kt;
            if (ts < 0 || (ts % a.stridet) != 0) continue;
            const int i = ts / a.stridet;
            if (i >= a.n1) continue;
            const int *nrow = nn0 + i;
            for (int by = by_lo; by <= by_hi; ++by, nrow += nbx * a.n1) {
                const int *np = nrow;
                for (int bx = bx_lo; bx <= bx_hi; ++bx, np += a.n1) s += ysp[(*np * a.stridet + kt) * NP];
            }
            cnt += npatch;
        }
        }
        const float wgt = fmaxf((float)cnt, 1e-10f);                  // utils_vid.py:228
        // (the vote count is a small integer: s * rcp(count) is within an ulp of the reference's quotient, a fifth of its instructions)
        const float v = a.normalize ? s * __builtin_amdgcn_rcpf(wgt) : s;
        if (a.sum) {      // uniform
            out[(size_t)tau * fs] = v;
            if (c == 0) wout[(size_t)tau * fs] = wgt;
        }
        if (a.x) {      // robust_lossfun(x - y2x) and its derivative while y2x is in a register (utils_vid.py:348)
            const float e = xq[0] - v;
#pragma unroll
            for (int k = 0; k + 1 < XQ; ++k) xq[k] = xq[k + 1];
            xq[XQ - 1] = xload();
            float f, g;
            rho_fg(a.rho, e, f, g);
            lacc += f;
#if defined(VL3D_FOLD_ABLATE) && (VL3D_FOLD_ABLATE & 4)      // measurement build only: ... and their gx as ONE 16-byte store
            if ((tid & 3) == 0 && xi + 3 < a.W) *reinterpret_cast<float4 *>(reinterpret_cast<uintptr_t>(gxp) & ~(uintptr_t)15) = make_float4(g, g, g, g);
#else
            *gxp = g * a.gscale;
#endif
            gxp += gxstep;
        }
    }
    }
    if (a.x) {          // block sum of the loss -> one double atomic
        __shared__ float red[FT_NT / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lacc += __shfl_down(lacc, off, 64);
        if ((tid & 63) == 0) red[tid >> 6] = lacc;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int i = 0; i < NT / 64; ++i) t += (double)red[i];
            atomicAdd(a.loss_sum, t);
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
        r.inv_b = 1.0f / r.b;
    }
    r.inv_scale = 1.0f / scale;
    return r;
}

int check_loss(const vl3d_loss_desc *d) {
    VL3D_REQUIRE(d != nullptr, "null loss desc");
    if (vl3d_check_variant(d->variant) != VL3D_OK) return VL3D_EINVAL;
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

static inline int pad4(int t) { return (t + 3) / 4 * 4; }
static inline int pad16(int t) { return (t + 15) / 16 * 16; }

extern "C" int64_t vl3d_patchnn_scratch_bytes(const vl3d_loss_desc *d) {
    if (!d || d->H <= 0 || d->W <= 0) return 0;
    const int TxU = ((d->Tx - d->pt) / d->stridet) * d->stridet + d->pt;
    // the larger of the two layouts: v6 gram16 (16 bytes per pixel and frame) and v4's pixel-major copies (3 floats, frames padded to 4)
    const int64_t v6 = 16 * ((int64_t)TxU + d->Ty), v4 = 12 * ((int64_t)pad4(TxU) + pad4(d->Ty));
    return (int64_t)d->H * d->W * (v6 > v4 ? v6 : v4);
}

// The split-f16 matrix-core kernel (v6) and its plan: instantiation, stage size, LDS bytes.  x in at most 16 NW frames (NW = 4 / 8 waves),
// y in at most 16 TYT (5 tiles x 4 locations, 8 x 2, 12 x 1 per wave).
struct NN6Plan { int nw, tyt, nl, ch, xs; size_t lds; bool ok; };
static NN6Plan plan_nn6(const NNArgs &a, int W, int Wy, int Ty, bool wave_epilogue, int FXm) {
    NN6Plan p{};
    const int FX = a.TxU, FY = Ty, TyT = (FY + 15) / 16;
    p.nw = FX <= 64 ? 4 : 8;
    // every wave finishes its own 14 patches (no alpha, three-frame patches at temporal stride 1, and the clip fits the overlapping tiles)
    p.xs = (wave_epilogue && !a.use_alpha && a.pt == 3 && a.stridet == 1 && a.n1 <= 14 * (FX <= 58 ? 4 : 8) && FX <= 14 * (FX <= 58 ? 4 : 8) + 2) ? 14 : 16;
    if (p.xs == 14) p.nw = FX <= 58 ? 4 : 8;
    p.tyt = TyT <= 5 ? 5 : (TyT <= 8 ? 8 : 12);
    p.nl = p.tyt == 5 ? 4 : (p.tyt == 8 ? 2 : 1);
    const int RWc = a.ps + (p.nl - 1) * a.stride;
    const size_t cell = (size_t)16 * (FX + FY), over = (size_t)(16 * p.tyt - FY + 16 * p.nw) * 16;      // (the tiles' over-read past the last cell)
    int ch = (int)((53 * 1024 / 2) / (cell * a.ps));
    ch = ch < 1 ? 1 : (ch > RWc ? RWc : ch);
    while (ch > 1 && ((size_t)ch * a.ps * FX > (size_t)NN6_KX * p.nw * 64 || (size_t)ch * a.ps * FY > (size_t)nn6_ky(p.tyt) * p.nw * 64)) --ch;
    p.ch = ch;
    const size_t stage = 2 * (size_t)p.ch * a.ps * cell + over;
    const int EP = ((a.TyP + 3) & ~7) + 4;
    const size_t ebuf = p.xs == 16 ? (size_t)a.TxP * a.TyP : (size_t)p.nw * 17 * EP;
    const size_t epi = (ebuf + (size_t)p.nl * (3 * ((a.n2 + 3) & ~3) + a.TyP + 16 * p.nw)) * sizeof(float);
    p.lds = stage > epi ? stage : epi;
    p.ok = FX <= 16 * p.nw && TyT <= 12 && p.lds <= 150 * 1024 &&
           (size_t)p.ch * a.ps * FX <= (size_t)NN6_KX * p.nw * 64 && (size_t)p.ch * a.ps * FY <= (size_t)nn6_ky(p.tyt) * p.nw * 64 &&   // KX / KY pieces per wave
           ((size_t)a.ps * W + p.ch) * (size_t)FXm < (1u << 24) && ((size_t)a.ps * Wy + p.ch) * (size_t)FY < (1u << 24) &&   // 24-bit DMA offsets
           16 * p.nw + a.TyP <= 64 * p.nw;                                                                             // side threads
    return p;
}

// variant bit 11 (0x800): workgroup-wide epilogue also without alpha (cross-checks)
static int launch_nn6(const NN6Plan &p, const NN2Args &b, int w_o, int h_o, hipStream_t s) {
    const int groups_x = (w_o + p.nl - 1) / p.nl;
    const dim3 grid6((unsigned)(groups_x * h_o));
#define VL3D_LAUNCH6(TYT_, NL_, NW_, XS_)                                                                                          \
    {                                                                                                                              \
        static bool attr = false;                                                                                                  \
        if (!attr) {                                                                                                               \
            VL3D_HIP(hipFuncSetAttribute((const void *)patchnn6_k<TYT_, NL_, NW_, XS_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr = true;                                                                                                           \
        }                                                                                                                          \
        hipLaunchKernelGGL((patchnn6_k<TYT_, NL_, NW_, XS_>), grid6, dim3(64 * NW_), p.lds, s, b, groups_x, p.ch);                  \
    }
#define VL3D_LAUNCH6_T(NW_, XS_)                                                                                                   \
    if (p.tyt == 5) VL3D_LAUNCH6(5, 4, NW_, XS_) else if (p.tyt == 8) VL3D_LAUNCH6(8, 2, NW_, XS_) else VL3D_LAUNCH6(12, 1, NW_, XS_)
    if (p.nw == 4) {
        if (p.xs == 14) { VL3D_LAUNCH6_T(4, 14) } else { VL3D_LAUNCH6_T(4, 16) }
    } else {
        if (p.xs == 14) { VL3D_LAUNCH6_T(8, 14) } else { VL3D_LAUNCH6_T(8, 16) }
    }
#undef VL3D_LAUNCH6_T
#undef VL3D_LAUNCH6
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

// x_gram / y_gram != nullptr: that video arrives in the NN kernel's own gram16 form (vl3d_video_to_gram_major / vl3d_loop_pad_fwd), y as the
// crop at (y_row0, y_col0) of a clip whose rows are y_pitch pixels long -- the captured video is constant training data, so it is
// rewritten once per pyramid level and not once per iteration; the render's x is written in that form by the loss prologue
static int patchnn_impl(const vl3d_loss_desc *desc, const float *x, const float *x_gram, int32_t x_pitch, int32_t x_frames, const float *y,
                        const float *y_gram, int32_t y_pitch, int32_t y_row0, int32_t y_col0, int32_t *nn, void *scratch, vl3d_stream_t stream) {
    int rc = check_loss(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE((x || x_gram) && (y || y_gram) && nn, "vl3d_patchnn: null pointer");
    NNArgs a{};
    size_t lds = 0;
    rc = plan_nn(desc, a, lds);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(!(y_gram && !x_gram) || scratch, "vl3d_patchnn_prepared: needs the scratch buffer (x's gram16 copy)");
    if (scratch != nullptr || (x_gram && y_gram)) {
        hipStream_t s = (hipStream_t)stream;
        const int pv = desc->variant & 0xf;
        VL3D_REQUIRE(!(x_gram || y_gram) || pv == 0 || pv == 6, "vl3d_patchnn: prepared (gram16) inputs run on the default kernel only");
        const int Wy = y_gram ? y_pitch : desc->W;
        NN2Args b{};
        b.nn = nn; b.W = desc->W; b.ps = a.ps; b.pt = a.pt; b.stride = a.stride; b.stridet = a.stridet;
        b.h_o = a.h_o; b.w_o = a.w_o; b.n1 = a.n1; b.n2 = a.n2; b.TxP = a.TxP; b.TyP = a.TyP; b.K = a.K; b.KC = a.KC;
        b.use_alpha = a.use_alpha; b.alpha = a.alpha; b.dnorm = a.inv_d;
        b.Wy = Wy;
        b.ablate = (desc->variant >> 4) & 15;
        dim3 tg((desc->W + 63) / 64, desc->H);
        // v6 (split-f16 matrix cores): the default wherever the clip lengths fit its instantiations
        const int Wx = x_gram ? x_pitch : desc->W, FXm = x_gram ? x_frames : a.TxU;
        const NN6Plan p6 = plan_nn6(a, Wx, Wy, desc->Ty, !(desc->variant & 0x800), FXm);
        const bool v4_has_tiles = (size_t)(a.TxP / TI) * (a.TyP / TJ) <= 1024;
        const bool use_v6 = p6.ok && (x_gram || y_gram || pv == 6 || (pv == 0 && !(p6.nl == 1 && v4_has_tiles)));
        if ((x_gram || y_gram) && !use_v6) {
            vl3d_set_error("vl3d_patchnn_prepared: these clip lengths / this frame width are outside the matrix-core kernel's range; use vl3d_patchnn");
            return VL3D_EUNSUPPORTED;
        }
        if (use_v6) {
            uint4 *xs = (uint4 *)scratch;
            if (!x_gram)
                hipLaunchKernelGGL(video_to_gram16_k<false>, tg, dim3(256), 0, s, x, desc->x_sc, desc->x_st, desc->x_sr, a.TxU, desc->H, desc->W, xs);
            const uint4 *ys = nullptr;
            if (y_gram) ys = reinterpret_cast<const uint4 *>(y_gram) + ((size_t)y_row0 * Wy + y_col0) * desc->Ty;
            else {
                // (the y copy sits behind the x copy of the largest x this scratch can hold: its place does not depend on x_gram)
                uint4 *yd = xs + (size_t)desc->H * desc->W * a.TxU;
                // variant bit 8: the y half of the scratch still holds this y from the previous call
                if (!(desc->variant & 0x100))
                    hipLaunchKernelGGL(video_to_gram16_k<true>, tg, dim3(256), 0, s, y, desc->y_sc, desc->y_st, desc->y_sr, desc->Ty, desc->H, desc->W, yd);
                ys = yd;
            }
            b.xt = x_gram ? x_gram : reinterpret_cast<const float *>(xs);
            b.yt = reinterpret_cast<const float *>(ys);
            b.PX = a.TxU; b.PY = desc->Ty;
            b.Wx = Wx; b.FXm = FXm;
            return launch_nn6(p6, b, a.w_o, a.h_o, s);
        }
        VL3D_REQUIRE(pv != 3 && pv != 6, "vl3d_patchnn: variant 3 (the fp32 matrix-core kernel) was removed in round 5; variant 6 needs clip lengths within the matrix-core kernel's range");
        float *xt = (float *)scratch;
        float *yt = xt + (size_t)desc->H * desc->W * 3 * a.TxP;
        hipLaunchKernelGGL(video_to_pixel_major_k, tg, dim3(256), 0, s, x, desc->x_sc, desc->x_st, desc->x_sr, a.TxU, a.TxP,
                           desc->H, desc->W, xt);
        if (!(desc->variant & 0x100))
            hipLaunchKernelGGL(video_to_pixel_major_k, tg, dim3(256), 0, s, y, desc->y_sc, desc->y_st, desc->y_sr, desc->Ty, a.TyP,
                               desc->H, desc->W, yt);
        b.xt = xt; b.yt = yt;
        static bool attr2 = false;
        if (!attr2) {
            VL3D_HIP(hipFuncSetAttribute((const void *)patchnn2_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr2 = true;
        }
        // v4 (4 locations per workgroup, column sums): default whenever one thread tile per frame-pair tile suffices
        const int ntiles4 = (a.TxP / TI) * (a.TyP / TJ);
        const int RWc4 = a.ps + (NL4 - 1) * a.stride;
        const size_t stage4 = (size_t)RWc4 * 3 * (a.TxP + a.TyP), epi4 = (size_t)a.TxP * a.TyP + 3 * (a.n2 + 3);   // the epilogue aliases the staging
        const size_t lds4 = (stage4 > epi4 ? stage4 : epi4) * sizeof(float);
        const bool use_v4 = (pv == 0 || pv == 4) && ntiles4 <= 1024 && lds4 <= 150 * 1024;
        if (use_v4) {
            static bool attr4 = false;
            if (!attr4) {
#define VL3D_ATTR4(R, N) VL3D_HIP(hipFuncSetAttribute((const void *)patchnn4_k<R, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
                VL3D_ATTR4(true, 256); VL3D_ATTR4(false, 256); VL3D_ATTR4(true, 512); VL3D_ATTR4(false, 512);
                VL3D_ATTR4(false, 1024);
#undef VL3D_ATTR4
                attr4 = true;
            }
            const int groups_x = (a.w_o + NL4 - 1) / NL4;
            const dim3 grid4((unsigned)(groups_x * a.h_o));
            // running sums pay when a column is shared by >= 2 locations on average (ps 11 / stride 4: 2.75; ps 3 / stride 2: 1.5);
            // one frame-pair tile per thread: 256 threads for the shipped clips, 512 / 1024 for longer ones
            const bool runsum = a.ps >= 2 * a.stride;
            const int nthr = ntiles4 <= 256 ? 256 : (ntiles4 <= 512 ? 512 : 1024);
#define VL3D_LAUNCH4(R, N) hipLaunchKernelGGL((patchnn4_k<R, N>), grid4, dim3(N), lds4, s, b, desc->H, groups_x)
            if (nthr == 256) { if (runsum) VL3D_LAUNCH4(true, 256); else VL3D_LAUNCH4(false, 256); }
            else if (nthr == 512) { if (runsum) VL3D_LAUNCH4(true, 512); else VL3D_LAUNCH4(false, 512); }
            else VL3D_LAUNCH4(false, 1024);       // the running-sum form needs 6 registers more than the 128 a 1024-thread workgroup has: per-column adds here
#undef VL3D_LAUNCH4
        } else {
            hipLaunchKernelGGL(patchnn2_k, dim3((unsigned)(a.h_o * a.w_o)), dim3(NN_THREADS), lds, s, b);
        }
        VL3D_CHECK_LAUNCH();
        return VL3D_OK;
    }
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

extern "C" int vl3d_patchnn(const vl3d_loss_desc *desc, const float *x, const float *y, int32_t *nn, void *scratch,
                            vl3d_stream_t stream) {
    return patchnn_impl(desc, x, nullptr, 0, 0, y, nullptr, 0, 0, 0, nn, scratch, stream);
}

extern "C" int64_t vl3d_gram_major_bytes(int32_t T, int32_t H, int32_t W) {
    if (T <= 0 || H <= 0 || W <= 0) return 0;
    return (int64_t)H * W * T * 16;
}

extern "C" int vl3d_video_to_gram_major(const float *y, int64_t sc, int64_t st, int64_t sr, int32_t T, int32_t H, int32_t W, float *out,
                                        vl3d_stream_t stream) {
    VL3D_REQUIRE(y && out && T > 0 && H > 0 && W > 0, "vl3d_video_to_gram_major: null pointer / non-positive dims");
    hipLaunchKernelGGL(video_to_gram16_k<true>, dim3((W + 63) / 64, H), dim3(256), 0, (hipStream_t)stream, y, sc, st, sr, T, H, W, (uint4 *)out);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_patchnn_prepared(const vl3d_loss_desc *desc, const float *x, const float *y_gram, int32_t y_pitch, int32_t y_rows,
                                     int32_t y_row0, int32_t y_col0, int32_t *nn, void *scratch, vl3d_stream_t stream) {
    VL3D_REQUIRE(desc && y_gram, "vl3d_patchnn_prepared: null pointer");
    VL3D_REQUIRE(y_row0 >= 0 && y_col0 >= 0 && y_row0 + desc->H <= y_rows && y_col0 + desc->W <= y_pitch,
                 "vl3d_patchnn_prepared: the crop (y_row0, y_col0) + (H, W) leaves the prepared clip (y_rows, y_pitch)");
    return patchnn_impl(desc, x, nullptr, 0, 0, nullptr, y_gram, y_pitch, y_row0, y_col0, nn, scratch, stream);
}

extern "C" int vl3d_patchnn_grams(const vl3d_loss_desc *desc, const float *x_gram, int32_t x_pitch, int32_t x_rows, int32_t x_frames, const float *y_gram,
                                  int32_t y_pitch, int32_t y_rows, int32_t y_row0, int32_t y_col0, int32_t *nn, vl3d_stream_t stream) {
    VL3D_REQUIRE(desc && x_gram && y_gram, "vl3d_patchnn_grams: null pointer");
    VL3D_REQUIRE(desc->H <= x_rows && desc->W <= x_pitch && desc->Tx <= x_frames, "vl3d_patchnn_grams: desc (Tx, H, W) leaves the x clip (x_frames, x_rows, x_pitch)");
    VL3D_REQUIRE(y_row0 >= 0 && y_col0 >= 0 && y_row0 + desc->H <= y_rows && y_col0 + desc->W <= y_pitch,
                 "vl3d_patchnn_grams: the crop (y_row0, y_col0) + (H, W) leaves the prepared clip (y_rows, y_pitch)");
    return patchnn_impl(desc, nullptr, x_gram, x_pitch, x_frames, nullptr, y_gram, y_pitch, y_row0, y_col0, nn, nullptr, stream);
}

// LDS-staged fold: tile shapes {FT_W, FT_H, threads}.  The whole Ty column of a tile has to sit in LDS (an NN index may point at
// any frame), so the tile area sets the LDS footprint and with it how many workgroups share a CU.  A 32x8 tile of a 75-frame clip
// (77 KiB + indices) runs ONE workgroup per CU, whose staging and voting phases never overlap with another's: 1.55 ms at 720p.
// 64x2 (three workgroups per CU, 256-byte row segments) measured 1.20 ms, 32x4 1.27, 32x2 1.28, 64x1 1.60 in one process
// (profiles/ab_fold.py; outputs bit-equal across shapes).  The narrower shapes are for clips whose columns do not fit otherwise.
struct FoldShape { int fw, fh, nt; };
static const FoldShape fold_shapes[] = {{64, 2, 512}, {32, 4, 512}, {32, 2, 256}, {64, 1, 256}, {32, 1, 256}};
constexpr int N_FOLD_SHAPES = sizeof(fold_shapes) / sizeof(fold_shapes[0]);
// most patch locations b (b*stride <= last, b*stride + ps - 1 >= first) covering a span of f pixels that starts at a multiple of f
static int fold_cover(int f, int ps, int stride) {
    int most = 0;
    for (int k = 0; k < stride; ++k) {
        const int first = (k + stride * ps) * f, last = first + f - 1;      // every residue of the tile origin, away from the border
        most = std::max(most, last / stride - (first - ps + stride) / stride + 1);
    }
    return most;
}
static size_t fold_lds_bytes(const vl3d_loss_desc *desc, int n1, const FoldShape &sh) {
    const int nby_max = fold_cover(sh.fh, desc->ps, desc->stride), nbx_max = fold_cover(sh.fw, desc->ps, desc->stride);
    // y columns + three zero frames | index rows of the covering locations + one row pointing at the zero frames
    return ((size_t)(desc->Ty + 3) * sh.fw * sh.fh + ((size_t)nby_max * nbx_max + 1) * n1) * sizeof(float);
}
// -1 = no shape fits.  First choice: the first shape that leaves room for three workgroups per CU, then two, then one.
// variant bits 12-15 (measurement hook): shape index + 1
static int fold_shape(const vl3d_loss_desc *desc, int n1) {
    const int forced = ((desc->variant >> 12) & 15) - 1;
    if (forced >= 0 && forced < N_FOLD_SHAPES && fold_lds_bytes(desc, n1, fold_shapes[forced]) <= 150 * 1024) return forced;
    for (size_t budget : {53 * 1024, 79 * 1024, 150 * 1024})
        for (int i = 0; i < N_FOLD_SHAPES; ++i)
            if (fold_lds_bytes(desc, n1, fold_shapes[i]) <= budget) return i;
    return -1;
}
template <int FW, int FH, int NT, int NB>
static int launch_fold_lds_nb(const vl3d_loss_desc *desc, const FoldArgs &a, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        VL3D_HIP(hipFuncSetAttribute((const void *)vote_fold_lds_k<FW, FH, NT, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        attr_set = true;
    }
    dim3 grid((desc->W + FW - 1) / FW, (desc->H + FH - 1) / FH, 3);
    hipLaunchKernelGGL((vote_fold_lds_k<FW, FH, NT, NB>), grid, dim3(NT), lds, s, a, desc->Ty);
    return VL3D_OK;
}
template <int FW, int FH, int NT>
static int launch_fold_lds(const vl3d_loss_desc *desc, const FoldArgs &a, size_t lds, hipStream_t s) {
    // fixed-trip covering loops for the patch-major form when a pixel has at most 2 x 2 / 3 x 3 covering locations (the shipped
    // configurations); variant bit 9 keeps the dynamic loops (A/B, cross-checks: same bits)
    const int cover = (desc->ps + desc->stride - 1) / desc->stride;
    const bool slide = desc->pt == 3 && desc->stridet == 1 && !(desc->variant & 0x200);
    if (slide && cover == 3) return launch_fold_lds_nb<FW, FH, NT, 3>(desc, a, lds, s);
    if (slide && cover <= 2) return launch_fold_lds_nb<FW, FH, NT, 2>(desc, a, lds, s);
    return launch_fold_lds_nb<FW, FH, NT, 0>(desc, a, lds, s);
}
static int launch_fold_lds_w(int shape, const vl3d_loss_desc *desc, const FoldArgs &a, hipStream_t s) {
    const size_t lds = fold_lds_bytes(desc, a.n1, fold_shapes[shape]);
    switch (shape) {
    case 0: return launch_fold_lds<64, 2, 512>(desc, a, lds, s);
    case 1: return launch_fold_lds<32, 4, 512>(desc, a, lds, s);
    case 2: return launch_fold_lds<32, 2, 256>(desc, a, lds, s);
    case 3: return launch_fold_lds<64, 1, 256>(desc, a, lds, s);
    default: return launch_fold_lds<32, 1, 256>(desc, a, lds, s);
    }
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
    // LDS-staged kernel when a tile's y columns + nn indices fit (32 wide for every shipped configuration, narrower for long clips)
    const int shape = fold_shape(desc, a.n1);
    if ((desc->variant & 0xf) != 1 && shape >= 0) {
        rc = launch_fold_lds_w(shape, desc, a, (hipStream_t)stream);
        if (rc != VL3D_OK) return rc;
    } else {
        dim3 grid((desc->W + 63) / 64, (desc->H + 3) / 4, desc->Tx);
        hipLaunchKernelGGL(vote_fold_k, grid, dim3(256), 0, (hipStream_t)stream, a);
    }
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_vote_fold_robust_strided(const vl3d_loss_desc *desc, const float *y, const int32_t *nn, const float *x, int32_t kind,
                                             float rou, float scale, float *y2x, float *weight, float *grad_x, int64_t gx_sc,
                                             int64_t gx_st, int64_t gx_sr, double *loss_sum, vl3d_stream_t stream) {
    int rc = check_loss(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(y && nn && x && grad_x && loss_sum, "vl3d_vote_fold_robust: null pointer");
    VL3D_REQUIRE((y2x == nullptr) == (weight == nullptr), "vl3d_vote_fold_robust: y2x and weight are written together or not at all");
    VL3D_REQUIRE(kind >= 0 && kind <= 2 && scale != 0.0f, "vl3d_vote_fold_robust: bad rho kind / scale");
    VL3D_REQUIRE(gx_sr >= desc->W && gx_st >= 0 && gx_sc >= 0, "vl3d_vote_fold_robust: bad grad_x strides");
    VL3D_REQUIRE(desc->Tx <= 65535, "vl3d_vote_fold_robust: Tx > 65535");
    FoldArgs a{};
    a.y = y; a.nn = nn; a.sum = y2x; a.weight = weight;
    a.Tx = desc->Tx; a.H = desc->H; a.W = desc->W; a.ps = desc->ps; a.pt = desc->pt;
    a.stride = desc->stride; a.stridet = desc->stridet;
    a.h_o = (desc->H - desc->ps) / desc->stride + 1;
    a.w_o = (desc->W - desc->ps) / desc->stride + 1;
    a.n1 = (desc->Tx - desc->pt) / desc->stridet + 1;
    a.y_sc = desc->y_sc; a.y_st = desc->y_st; a.y_sr = desc->y_sr;
    a.normalize = 1;
    a.x = x; a.x_sc = desc->x_sc; a.x_st = desc->x_st; a.x_sr = desc->x_sr;
    a.rho = make_rho(kind, rou, scale);
    a.gscale = 1.0f / (3.0f * (float)desc->Tx * (float)desc->H * (float)desc->W);     // d(mean)/d(element)
    a.gx = grad_x; a.gx_sc = gx_sc; a.gx_st = gx_st; a.gx_sr = gx_sr; a.loss_sum = loss_sum;
    const int shape = fold_shape(desc, a.n1);
    if (shape < 0) {
        vl3d_set_error("vl3d_vote_fold_robust: tile does not fit LDS; use vl3d_vote_fold + vl3d_robust_fwd/bwd");
        return VL3D_EUNSUPPORTED;
    }
    VL3D_HIP(hipMemsetAsync(loss_sum, 0, sizeof(double), (hipStream_t)stream));
    rc = launch_fold_lds_w(shape, desc, a, (hipStream_t)stream);
    if (rc != VL3D_OK) return rc;
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_vote_fold_robust(const vl3d_loss_desc *desc, const float *y, const int32_t *nn, const float *x, int32_t kind,
                                     float rou, float scale, float *y2x, float *weight, float *grad_x, double *loss_sum,
                                     vl3d_stream_t stream) {
    if (!desc) { vl3d_set_error("vl3d_vote_fold_robust: null descriptor"); return VL3D_EINVAL; }
    const int64_t fs = (int64_t)desc->H * desc->W;
    return vl3d_vote_fold_robust_strided(desc, y, nn, x, kind, rou, scale, y2x, weight, grad_x, fs * desc->Tx, fs, desc->W, loss_sum, stream);
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

// x *= *scale, skipped entirely (one scalar load per wave, no traffic) when the scale is exactly 1: the upstream gradient of a loss that is
// differentiated directly is 1, and the fused looping loss has its gradient buffer ready since the forward
__global__ __launch_bounds__(256) void scale_inplace_k(int64_t n4, float4 *__restrict__ x, const float *__restrict__ scale) {
    const float s = *scale;
    if (s == 1.0f) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = x[i];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        x[i] = v;
    }
}

extern "C" int vl3d_scale_inplace(int64_t n, float *x, const float *scale, vl3d_stream_t stream) {
    VL3D_REQUIRE(n > 0 && (n & 3) == 0 && x && scale, "vl3d_scale_inplace: n must be a positive multiple of 4 (16-byte aligned buffer)");
    const int64_t n4 = n / 4;
    const unsigned blocks = (unsigned)(ceil_div64(n4, 256) < 8192 ? ceil_div64(n4, 256) : 8192);
    hipLaunchKernelGGL(scale_inplace_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n4, reinterpret_cast<float4 *>(x), scale);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

// ---------------------------------------------------------------------------------------------------
// The loss prologue of MPMeshVid.forward (MPV.py:484-507): NHWC render -> loop-padded, scale-invariant-gained video for the loss.
//   rgb_pad = cat(rgb, rgb[:pad]);  scale = (exp(mean(log((mean_f res + .01) / (mean_t rgb.detach() + .01)))) + 3) / 4;  x = rgb_pad * scale
// was cat + mul (two copies of the padded clip), two frame means, five scalar kernels, and in the backward mul, two slices, an add and the
// NCHW -> NHWC copy the render backward needs -- ~20 launches around kernels that take 0.1-0.2 ms at the training crop.  Here: one kernel
// for the log-ratio sum, one that writes x [3,T+pad,h,w] from rgb [T,h,w,3], one that folds the pad frames' gradient back into
// g_rgb [T,h,w,3] (the gain has no gradient: the reference detaches rgb in it).
// res (F,3,h,w) through strides in floats (r_sf frame, r_sc channel, r_sr row; unit column stride): a crop of the captured clip needs no copy
__global__ __launch_bounds__(256) void loop_gain_k(int T, int F, int64_t hw, int w, const float *__restrict__ rgb, const float *__restrict__ res,
                                                   int64_t r_sf, int64_t r_sc, int64_t r_sr, double *__restrict__ log_sum) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    if (p < hw) {
        float r0 = 0.f, r1 = 0.f, r2 = 0.f, y0 = 0.f, y1 = 0.f, y2 = 0.f;
        const float *rp = rgb + p * 3;
#pragma unroll 4
        for (int t = 0; t < T; ++t, rp += hw * 3) { r0 += rp[0]; r1 += rp[1]; r2 += rp[2]; }
        const int64_t py = p / w, px = p - py * w;
        const float *yp = res + py * r_sr + px;
#pragma unroll 4
        for (int f = 0; f < F; ++f, yp += r_sf) { y0 += yp[0]; y1 += yp[r_sc]; y2 += yp[2 * r_sc]; }
        const float it = 1.0f / (float)T, jf = 1.0f / (float)F;
        v = __logf((y0 * jf + 0.01f) / (r0 * it + 0.01f)) + __logf((y1 * jf + 0.01f) / (r1 * it + 0.01f)) + __logf((y2 * jf + 0.01f) / (r2 * it + 0.01f));
    }
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(log_sum, (double)((red[0] + red[1]) + (red[2] + red[3])));
}

__device__ __forceinline__ float loop_gain(const double *log_sum, int64_t hw) {
    if (!log_sum) return 1.0f;
    return (__expf((float)(*log_sum / (double)(3 * hw))) + 3.0f) * 0.25f;
}

__global__ __launch_bounds__(256) void loop_pad_fwd_k(int T, int pad, int64_t hw, const float *__restrict__ rgb, const double *__restrict__ log_sum,
                                                      float *__restrict__ x) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y, ts = t < T ? t : t - T;
    if (p >= hw) return;
    const float g = loop_gain(log_sum, hw);
    const float *rp = rgb + ((int64_t)ts * hw + p) * 3;
    const int64_t frames = T + pad;
    x[((int64_t)0 * frames + t) * hw + p] = rp[0] * g;
    x[((int64_t)1 * frames + t) * hw + p] = rp[1] * g;
    x[((int64_t)2 * frames + t) * hw + p] = rp[2] * g;
}

// ... and the same with the NN kernel's gram16 form of x written beside the video: the render's NHWC output is read ONCE for the loss's
// two layouts (the separate video -> gram16 pass re-read x: 0.30 of the 1.66 ms of a 720p search).  A workgroup takes 64 pixels of a row
// through all T + pad frames in groups of 16: lanes over pixels read 768 contiguous bytes per frame, write x coalesced per channel, the
// tile goes through LDS, 16 lanes = the 16 frames of a pixel write 256 contiguous bytes of pieces.
__global__ __launch_bounds__(256) void loop_pad_fwd_gram_k(int T, int pad, int H, int W, const float *__restrict__ rgb, const double *__restrict__ log_sum,
                                                           float *__restrict__ x, uint4 *__restrict__ xg) {
    __shared__ float tile[3][16][65];
    const int row = blockIdx.y, x0 = blockIdx.x * 64, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int F = T + pad, G = (F + 15) >> 4;
    const int64_t hw = (int64_t)H * W;
    const float gain = loop_gain(log_sum, hw);
    const int px = min(x0 + lane, W - 1);
    const bool inx = x0 + lane < W;
    float pre[12];
    auto load = [&](int f0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int j = wave + 4 * k, c = j >> 4, f = f0 + (j & 15), ts = f < T ? f : f - T;
            pre[k] = f < F ? rgb[(((int64_t)ts * H + row) * W + px) * 3 + c] * gain : 0.5f;
        }
    };
    load(0);
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int j = wave + 4 * k, c = j >> 4, f = g * 16 + (j & 15);
            if (f < F && inx) x[((int64_t)c * F + f) * hw + (int64_t)row * W + px] = pre[k];
            tile[c][j & 15][lane] = pre[k] - 0.5f;
        }
        __syncthreads();
        if (g + 1 < G) load((g + 1) * 16);
        const int m = lane & 15, f = g * 16 + m;
        for (int p = wave * 4 + (lane >> 4); p < 64; p += 16)
            if (x0 + p < W && f < F)
                xg[((size_t)row * W + x0 + p) * F + f] = gram16_piece(tile[0][m][p], tile[1][m][p], tile[2][m][p], 1.0f);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void loop_pad_bwd_k(int T, int pad, int64_t hw, const float *__restrict__ gx, int64_t gx_sc, int64_t gx_st,
                                                      const double *__restrict__ log_sum, float *__restrict__ g_rgb) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y;
    if (p >= hw) return;
    const float g = loop_gain(log_sum, hw);
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v[c] = gx[c * gx_sc + t * gx_st + p];
        if (t < pad) v[c] += gx[c * gx_sc + (T + t) * gx_st + p];
    }
    float *o = g_rgb + ((int64_t)t * hw + p) * 3;
    o[0] = v[0] * g; o[1] = v[1] * g; o[2] = v[2] * g;
}

extern "C" int vl3d_loop_gain_strided(int32_t T, int32_t F, int32_t h, int32_t w, const float *rgb, const float *res, int64_t r_sf, int64_t r_sc,
                                      int64_t r_sr, double *log_sum, vl3d_stream_t stream) {
    VL3D_REQUIRE(T > 0 && F > 0 && h > 0 && w > 0 && rgb && res && log_sum, "vl3d_loop_gain: bad arguments");
    const int64_t hw = (int64_t)h * w;
    VL3D_HIP(hipMemsetAsync(log_sum, 0, sizeof(double), (hipStream_t)stream));
    hipLaunchKernelGGL(loop_gain_k, dim3((unsigned)ceil_div64(hw, 256)), dim3(256), 0, (hipStream_t)stream, T, F, hw, w, rgb, res, r_sf, r_sc, r_sr,
                       log_sum);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_loop_gain(int32_t T, int32_t F, int32_t h, int32_t w, const float *rgb, const float *res, double *log_sum,
                              vl3d_stream_t stream) {
    const int64_t hw = (int64_t)(h > 0 ? h : 0) * (w > 0 ? w : 0);
    return vl3d_loop_gain_strided(T, F, h, w, rgb, res, 3 * hw, hw, w, log_sum, stream);
}

extern "C" int vl3d_loop_pad_fwd(int32_t T, int32_t pad, int32_t h, int32_t w, const float *rgb, const double *log_sum, float *x,
                                 vl3d_stream_t stream) {
    VL3D_REQUIRE(T > 0 && pad >= 0 && pad <= T && T + pad <= 65535 && h > 0 && w > 0 && rgb && x, "vl3d_loop_pad_fwd: bad arguments");
    const int64_t hw = (int64_t)h * w;
    hipLaunchKernelGGL(loop_pad_fwd_k, dim3((unsigned)ceil_div64(hw, 256), (unsigned)(T + pad)), dim3(256), 0, (hipStream_t)stream, T, pad, hw, rgb,
                       log_sum, x);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_loop_pad_fwd_gram(int32_t T, int32_t pad, int32_t h, int32_t w, const float *rgb, const double *log_sum, float *x, float *x_gram,
                                      vl3d_stream_t stream) {
    VL3D_REQUIRE(T > 0 && pad >= 0 && pad <= T && T + pad <= 65535 && h > 0 && w > 0 && rgb && x && x_gram, "vl3d_loop_pad_fwd_gram: bad arguments");
    hipLaunchKernelGGL(loop_pad_fwd_gram_k, dim3((unsigned)((w + 63) / 64), (unsigned)h), dim3(256), 0, (hipStream_t)stream, T, pad, h, w, rgb, log_sum, x,
                       reinterpret_cast<uint4 *>(x_gram));
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_loop_pad_bwd(int32_t T, int32_t pad, int32_t h, int32_t w, const float *grad_x, int64_t gx_sc, int64_t gx_st,
                                 const double *log_sum, float *grad_rgb, vl3d_stream_t stream) {
    VL3D_REQUIRE(T > 0 && pad >= 0 && pad <= T && T <= 65535 && h > 0 && w > 0 && grad_x && grad_rgb, "vl3d_loop_pad_bwd: bad arguments");
    const int64_t hw = (int64_t)h * w;
    hipLaunchKernelGGL(loop_pad_bwd_k, dim3((unsigned)ceil_div64(hw, 256), (unsigned)T), dim3(256), 0, (hipStream_t)stream, T, pad, hw, grad_x, gx_sc,
                       gx_st, log_sum, grad_rgb);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_nn_vectors(int64_t B, int32_t n1, int32_t n2, int32_t d, const float *X, const float *Y, int32_t use_alpha,
                               float alpha, int64_t *nn, vl3d_stream_t stream) {
    VL3D_REQUIRE(B > 0 && B < (1ll << 31) && n1 > 0 && n2 > 0 && d > 0 && X && Y && nn, "vl3d_nn_vectors: bad arguments");
    const size_t fixed = ((size_t)n1 * n2 + n2) * sizeof(float);
    VL3D_REQUIRE(fixed + (size_t)(n1 + n2) * 5 * sizeof(float) <= 150 * 1024, "vl3d_nn_vectors: n1*n2 distance matrix does not fit in LDS");
    int kc = (int)((150 * 1024 - fixed) / ((size_t)(n1 + n2) * sizeof(float))) - 1;
    if (kc > 64) kc = 64;
    if (kc > d) kc = d;
    const size_t lds = fixed + (size_t)(n1 + n2) * (kc + 1) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        VL3D_HIP(hipFuncSetAttribute((const void *)nn_vectors_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(nn_vectors_k, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, X, Y, n1, n2, d, use_alpha, alpha, nn, kc);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

// ---------------------------------------------------------------------------------------------------
// NN-error metric support (evaluations/NNMSE.py:45-56): per patch location b
//   err[b] = sum_i sum_{c,kt,kh,kw} | y[c, nn_b[i]*st+kt, .] - x[c, i*st+kt, .] |   (the caller normalises / groups by macro block)
__global__ __launch_bounds__(256) void patch_l1_k(const float *__restrict__ x, const float *__restrict__ y, const int32_t *__restrict__ nn,
                                                  int ps, int pt, int stride, int stridet, int w_o, int n1, int64_t x_sc, int64_t x_st,
                                                  int64_t x_sr, int64_t y_sc, int64_t y_st, int64_t y_sr, float *__restrict__ err) {
    __shared__ float red[4];
    const int b = blockIdx.x, by = b / w_o, bx = b % w_o, r0 = by * stride, c0 = bx * stride;
    const int per = 3 * pt * ps * ps, total = n1 * per;
    float acc = 0.f;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int i = e / per, k = e - i * per;
        const int c = k / (pt * ps * ps), r1 = k - c * pt * ps * ps, kt = r1 / (ps * ps), r2 = r1 - kt * ps * ps, kh = r2 / ps, kw = r2 - kh * ps;
        const int j = nn[(size_t)b * n1 + i];
        const int64_t sp = (int64_t)(r0 + kh);
        const float xv = x[c * x_sc + (int64_t)(i * stridet + kt) * x_st + sp * x_sr + c0 + kw];
        const float yv = y[c * y_sc + (int64_t)(j * stridet + kt) * y_st + sp * y_sr + c0 + kw];
        acc += fabsf(yv - xv);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) err[b] = red[0] + red[1] + red[2] + red[3];
}

extern "C" int vl3d_patch_l1(const vl3d_loss_desc *desc, const float *x, const float *y, const int32_t *nn, float *err,
                             vl3d_stream_t stream) {
    int rc = check_loss(desc);
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(x && y && nn && err, "vl3d_patch_l1: null pointer");
    const int h_o = (desc->H - desc->ps) / desc->stride + 1, w_o = (desc->W - desc->ps) / desc->stride + 1;
    const int n1 = (desc->Tx - desc->pt) / desc->stridet + 1;
    hipLaunchKernelGGL(patch_l1_k, dim3((unsigned)(h_o * w_o)), dim3(256), 0, (hipStream_t)stream, x, y, nn, desc->ps, desc->pt,
                       desc->stride, desc->stridet, w_o, n1, desc->x_sc, desc->x_st, desc->x_sr, desc->y_sc, desc->y_st, desc->y_sr, err);
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}
