// Mixed read+write HBM streaming ceiling on MI355X: what a hand-written kernel reaches when it WRITES about as much as it reads --
// the backward render's pattern (24 GB of stack in, 24 GB of gradient out per cfg3 launch; read_bw.hip is the read-only ceiling
// the forward is priced against).  Sweeps the read:write ratio, temporal vs non-temporal stores, workgroup shape and the
// plane-strided layout of the render (D streams 1.2 GB apart, 1-KiB row segments).
//   hipcc --offload-arch=gfx950 -O3 -o rw_bw rw_bw.hip && ./rw_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

// R reads per W writes, grid-strided, one-shot per thread group of UNROLL elements
template <int R, int W, bool NT, int UNROLL>
__global__ void rw_k(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            v[u] = src[i + u * stride];
#pragma unroll
            for (int r = 1; r < R; ++r) v[u] += src[(size_t)r * n + i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int w = 0; w < W; ++w) {
                if constexpr (NT) __builtin_nontemporal_store(v[u], &dst[(size_t)w * n + i + u * stride]);
                else dst[(size_t)w * n + i + u * stride] = v[u];
            }
    }
}

// render-shaped: a workgroup of 512 threads owns a 32 x 16-texel tile of frames t, t+1 and walks D planes front to back; per
// plane it reads its tile (2 x 512 x 16 B) and writes the same tile of the gradient.  Layout (D, T, Hs, Ws) of 16-byte texels.
template <bool NT, int PF>
__global__ __launch_bounds__(512) void render_like_k(const f4 *__restrict__ src, f4 *__restrict__ dst, int D, int T, int Hs, int Ws, int tiles_x,
                                                     int tiles_y) {
    const int b = blockIdx.x;
    const int q = gridDim.x >> 3, r = gridDim.x & 7, xcd = b & 7, k = b >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    const int tile_x = bid % tiles_x, rest = bid / tiles_x, tile_y = rest % tiles_y, t0 = (rest / tiles_y) * 2;
    const int x = tile_x * 32 + (threadIdx.x & 31), y = tile_y * 16 + (threadIdx.x >> 5);
    if (x >= Ws || y >= Hs) return;
    const size_t frame = (size_t)Hs * Ws, plane = (size_t)T * frame;
    size_t o = (size_t)t0 * frame + (size_t)y * Ws + x;
    f4 a[PF], bq[PF];
    for (int d = 0; d < D; d += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) { a[p] = src[o + p * plane]; bq[p] = src[o + p * plane + frame]; }
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if constexpr (NT) { __builtin_nontemporal_store(a[p] * 2.f, &dst[o + p * plane]); __builtin_nontemporal_store(bq[p] * 2.f, &dst[o + p * plane + frame]); }
            else { dst[o + p * plane] = a[p] * 2.f; dst[o + p * plane + frame] = bq[p] * 2.f; }
        }
        o += PF * plane;
    }
}

// the backward's memory pattern without its arithmetic: a workgroup owns an (RX-2) x (RY-2) pixel tile of FR frames, its RX x RY
// threads each read TAPS taps per plane and frame (offsets 0, +1 texel, +1 row, +1 row +1 texel: the L1 absorbs the overlap),
// the interior threads write one gradient texel per plane and frame (non-temporal), plus a 2-byte owner-table entry per plane.
// STORE: 0 non-temporal stores of the interior (ragged row segments, as the owner-computes backward writes them), 1 the same with
// plain write-back stores (the L2 may merge the partial lines of horizontally adjacent tiles), 2 non-temporal stores of row segments
// ALIGNED to SNAP texels (ownership snapped to SNAP-texel columns: what aligned ownership would give), 3 = 2 with plain stores
// HX: halo columns on either side (1 = the shipped kernels).  Ownership snapped to SNAP-texel columns moves a segment's ends by up to
// SNAP / 2 texels, so a REAL aligned-ownership kernel has to stage 1 + SNAP / 2 halo columns: HX = 3 for SNAP = 4 (26 of 32 columns owned).
// HY: halo rows above and below (1 = the shipped kernels; 0 = the memory pattern of a backward WITHOUT the vertical halo -- a persistent
// workgroup walking a column strip top to bottom that carries the boundary rows' staged values from block to block: its upper bound, the
// carry itself costs nothing here).
template <int RX, int RY, int FR, int TAPS, bool OWNER, int STORE = 0, int SNAP = 8, int HX = 1, int HY = 1>
__global__ __launch_bounds__(RX *RY) void bwd_like_k(const f4 *__restrict__ src, f4 *__restrict__ dst, const unsigned short *__restrict__ owner, int D,
                                                     int T, int Hs, int Ws, int tiles_x, int tiles_y) {
    const int b = blockIdx.x;
    const int q = gridDim.x >> 3, r = gridDim.x & 7, xcd = b & 7, k = b >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    const int tile_x = bid % tiles_x, rest = bid / tiles_x, tile_y = rest % tiles_y, t0 = (rest / tiles_y) * FR;
    const int lx = threadIdx.x % RX, ly = threadIdx.x / RX;
    constexpr int IWX = RX - 2 * HX;
    const int gx = tile_x * IWX - HX + lx;
    const int gy = tile_y * (RY - 2 * HY) - HY + ly;
    const int x = min(max(gx, 0), Ws - 2), y = min(max(gy, 0), Hs - 2);
    bool interior = lx >= HX && lx < RX - HX && ly >= HY && ly < RY - HY && gx < Ws && gy < Hs;
    if constexpr (STORE >= 2) {      // owned columns [snap(tile_x * IW), snap((tile_x + 1) * IW)): every column owned exactly once, SNAP-aligned ends
        const int l = (tile_x * IWX + SNAP / 2) / SNAP * SNAP, rr = ((tile_x + 1) * IWX + SNAP / 2) / SNAP * SNAP;
        interior = gx >= l && gx < rr && gx < Ws && ly >= HY && ly < RY - HY && gy < Hs && gx >= 0;
    }
    const size_t frame = (size_t)Hs * Ws, plane = (size_t)T * frame;
    size_t o = (size_t)t0 * frame + (size_t)y * Ws + x;
    size_t oo = (size_t)y * Ws + x;
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < D; ++d, o += plane, oo += frame) {
        f4 v[FR];
        unsigned e = 0;
        if constexpr (OWNER) e = owner[oo];
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            v[f] = src[o + f * frame];
            if constexpr (TAPS == 4) v[f] += src[o + f * frame + 1] + src[o + f * frame + Ws] + src[o + f * frame + Ws + 1];
        }
        if (interior) {
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                if constexpr (STORE == 0 || STORE == 2) __builtin_nontemporal_store(v[f] + acc, &dst[o + f * frame]);
                else dst[o + f * frame] = v[f] + acc;
            }
        }
        acc.x += (float)e;
    }
}

// The fp16-STACK backward's pattern (round 6; verdict round 5, item 4: "what do 1.47x read traffic and the mixed stream allow?"): bwd_like_k's halo
// pattern with 8-BYTE texels (four halves) -- a wave's row segment is 256 B (region 32 wide) or 512 B (64 wide) instead of 512 B / 1 KiB, the
// gradient store 8 bytes per lane.  E = float2 stands in for the four halves (same bytes, same addresses).  Same geometry and frame pairing
// as the shipped fp16 instantiation of render_bwd_pair_k (32 x 16 x 2).
template <typename E, int RX, int RY, int FR>
__global__ __launch_bounds__(RX *RY) void bwd_like_e_k(const E *__restrict__ src, E *__restrict__ dst, const unsigned short *__restrict__ owner, int D,
                                                       int T, int Hs, int Ws, int tiles_x, int tiles_y) {
    const int b = blockIdx.x;
    const int q = gridDim.x >> 3, r = gridDim.x & 7, xcd = b & 7, k = b >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    const int tile_x = bid % tiles_x, rest = bid / tiles_x, tile_y = rest % tiles_y, t0 = (rest / tiles_y) * FR;
    const int lx = threadIdx.x % RX, ly = threadIdx.x / RX;
    const int gx = tile_x * (RX - 2) - 1 + lx, gy = tile_y * (RY - 2) - 1 + ly;
    const int x = min(max(gx, 0), Ws - 2), y = min(max(gy, 0), Hs - 2);
    const bool interior = lx >= 1 && lx < RX - 1 && ly >= 1 && ly < RY - 1 && gx < Ws && gy < Hs;
    const size_t frame = (size_t)Hs * Ws, plane = (size_t)T * frame;
    size_t o = (size_t)t0 * frame + (size_t)y * Ws + x;
    size_t oo = (size_t)y * Ws + x;
    float acc = 0.f;
    for (int d = 0; d < D; ++d, o += plane, oo += frame) {
        E v[FR];
        const unsigned e = owner[oo];
#pragma unroll
        for (int f = 0; f < FR; ++f) v[f] = src[o + f * frame] + src[o + f * frame + 1] + src[o + f * frame + Ws] + src[o + f * frame + Ws + 1];
        if (interior) {
#pragma unroll
            for (int f = 0; f < FR; ++f) __builtin_nontemporal_store(v[f] + acc, &dst[o + f * frame]);
        }
        acc += (float)e;
    }
}

// The optimiser fused into the owner store (verdict round 3, next #3), as a memory pattern: the shipped halo pattern, but an interior thread
// does not store its gradient texel -- it READS the two moments of its texel (m, v: two more streams at the tile's ragged segments; p comes
// with the taps) and WRITES p, m, v (three streams instead of one).  Compared below with what it replaces: the shipped pattern's one
// gradient store plus a separate streaming step kernel over whole rows (4 reads p, g, m, v; 3 writes p, m, v).
template <int RX, int RY, int FR, bool NT = true>
__global__ __launch_bounds__(RX *RY) void bwd_like_adam_k(const f4 *__restrict__ src, f4 *__restrict__ dst, const unsigned short *__restrict__ owner, int D,
                                                          int T, int Hs, int Ws, int tiles_x, int tiles_y, size_t unit) {
    const int b = blockIdx.x;
    const int q = gridDim.x >> 3, r = gridDim.x & 7, xcd = b & 7, k = b >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    const int tile_x = bid % tiles_x, rest = bid / tiles_x, tile_y = rest % tiles_y, t0 = (rest / tiles_y) * FR;
    const int lx = threadIdx.x % RX, ly = threadIdx.x / RX;
    const int gx = tile_x * (RX - 2) - 1 + lx;
    const int x = min(max(gx, 0), Ws - 2), y = min(max(tile_y * (RY - 2) - 1 + ly, 0), Hs - 2);
    const bool interior = lx >= 1 && lx < RX - 1 && ly >= 1 && ly < RY - 1 && gx < Ws && tile_y * (RY - 2) - 1 + ly < Hs;
    const size_t frame = (size_t)Hs * Ws, plane = (size_t)T * frame;
    size_t o = (size_t)t0 * frame + (size_t)y * Ws + x;
    size_t oo = (size_t)y * Ws + x;
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < D; ++d, o += plane, oo += frame) {
        f4 v[FR];
        const unsigned e = owner[oo];
#pragma unroll
        for (int f = 0; f < FR; ++f) v[f] = src[o + f * frame] + src[o + f * frame + 1] + src[o + f * frame + Ws] + src[o + f * frame + Ws + 1];
        if (interior) {
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                const size_t i = o + f * frame;
                const f4 g = v[f] + acc, m = src[unit + i] * 0.9f + g * 0.1f, vv = src[2 * unit + i] * 0.999f + g * g * 0.001f;
                if constexpr (NT) {
                    __builtin_nontemporal_store(v[f] - m * 0.01f, &dst[i]);
                    __builtin_nontemporal_store(m, &dst[unit + i]);
                    __builtin_nontemporal_store(vv, &dst[2 * unit + i]);
                } else {      // plain write-back stores: the L2 may merge the partial lines of horizontally adjacent tiles
                    dst[i] = v[f] - m * 0.01f;
                    dst[unit + i] = m;
                    dst[2 * unit + i] = vv;
                }
            }
        }
        acc.x += (float)e;
    }
}

// HALO-ATOMICS backward (SURVEY §7's design, priced in round 4): NO halo -- a workgroup sweeps exactly the RX x RY pixels it owns (x1.00
// instead of x1.22 pixels swept per pixel owned) -- so a texel whose 3 x 3 gathering pixels straddle a tile border gets a PARTIAL sum from
// each side: the tile's perimeter threads add theirs with float atomics (4 x global_atomic_add_f32 per 16-byte texel) to their own
// texel AND to the texel across the border (owned by the neighbour, who needs this tile's share), interior threads keep the plain
// non-temporal store.  The ring's texels must start from zero: ZERO = 1 charges that fill to the kernel's own time (a pre-pass writes
// the perimeter texels of every tile, as the owner-table pre-pass would).  Sums arrive in any order: not bitwise reproducible.
template <int RX, int RY, int FR, int TAPS>
__global__ __launch_bounds__(RX *RY) void bwd_like_ha_k(const f4 *__restrict__ src, f4 *__restrict__ dst, const unsigned short *__restrict__ owner, int D,
                                                        int T, int Hs, int Ws, int tiles_x, int tiles_y) {
    const int b = blockIdx.x;
    const int q = gridDim.x >> 3, r = gridDim.x & 7, xcd = b & 7, k = b >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    const int tile_x = bid % tiles_x, rest = bid / tiles_x, tile_y = rest % tiles_y, t0 = (rest / tiles_y) * FR;
    const int lx = threadIdx.x % RX, ly = threadIdx.x / RX;
    const int gx = tile_x * RX + lx, gy = tile_y * RY + ly;
    const int x = min(max(gx, 1), Ws - 3), y = min(max(gy, 1), Hs - 3);
    const bool live = gx < Ws && gy < Hs;
    const bool ring = lx == 0 || lx == RX - 1 || ly == 0 || ly == RY - 1;
    // the texel across the border this perimeter thread also contributes to (corners: the row neighbour; the diagonal's share is 1/16 of the
    // ring and would only add atomics)
    const int nb = ly == 0 ? -Ws : (ly == RY - 1 ? Ws : (lx == 0 ? -1 : 1));
    const size_t frame = (size_t)Hs * Ws, plane = (size_t)T * frame;
    size_t o = (size_t)t0 * frame + (size_t)y * Ws + x;
    size_t oo = (size_t)y * Ws + x;
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < D; ++d, o += plane, oo += frame) {
        f4 v[FR];
        const unsigned e = owner[oo];
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            v[f] = src[o + f * frame];
            if constexpr (TAPS == 4) v[f] += src[o + f * frame + 1] + src[o + f * frame + Ws] + src[o + f * frame + Ws + 1];
        }
        if (live) {
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                const f4 g = v[f] + acc;
                if (!ring) {
                    __builtin_nontemporal_store(g, &dst[o + f * frame]);
                } else {
                    float *p0 = reinterpret_cast<float *>(&dst[o + f * frame]), *p1 = reinterpret_cast<float *>(&dst[o + f * frame + nb]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { unsafeAtomicAdd(p0 + c, g[c]); unsafeAtomicAdd(p1 + c, g[c] * 0.5f); }
                }
            }
        }
        acc.x += (float)e;
    }
}
// the zero fill the ring needs before the atomics: the perimeter texels (and the row / column across the border) of every tile, all planes and frames
template <int RX, int RY>
__global__ __launch_bounds__(256) void ha_zero_ring_k(f4 *__restrict__ dst, int D, int T, int Hs, int Ws) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Ws || y >= Hs) return;
    const int mx = x % RX, my = y % RY;
    if (!(mx == 0 || mx == RX - 1 || my == 0 || my == RY - 1)) return;
    const size_t frame = (size_t)Hs * Ws;
    f4 *g = dst + (size_t)blockIdx.z * T * frame + (size_t)y * Ws + x;
    for (int t = 0; t < T; ++t, g += frame) __builtin_nontemporal_store(f4{0.f, 0.f, 0.f, 0.f}, g);
}

// the same pattern on a FRAME-PAIR INTERLEAVED layout (D, T/2, Hs, Ws, 2 frames, 4): a thread's two frames are 32 contiguous bytes,
// a wave's row segment is 2 KiB (region 64 wide) or 1 KiB (32 wide) -- what a pair kernel would stream if the stack were stored so.
template <int RX, int RY, bool OWNER, int STORE, int SNAP>
__global__ __launch_bounds__(RX *RY) void bwd_like_il_k(const f4 *__restrict__ src, f4 *__restrict__ dst, const unsigned short *__restrict__ owner, int D,
                                                        int T, int Hs, int Ws, int tiles_x, int tiles_y) {
    const int b = blockIdx.x;
    const int q = gridDim.x >> 3, r = gridDim.x & 7, xcd = b & 7, k = b >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    const int tile_x = bid % tiles_x, rest = bid / tiles_x, tile_y = rest % tiles_y, tp = rest / tiles_y;
    const int lx = threadIdx.x % RX, ly = threadIdx.x / RX;
    const int x = min(max(tile_x * (RX - 2) - 1 + lx, 0), Ws - 2), y = min(max(tile_y * (RY - 2) - 1 + ly, 0), Hs - 2);
    bool interior = lx >= 1 && lx < RX - 1 && ly >= 1 && ly < RY - 1 && tile_x * (RX - 2) - 1 + lx < Ws && tile_y * (RY - 2) - 1 + ly < Hs;
    if constexpr (STORE >= 2) {
        const int gx = tile_x * (RX - 2) - 1 + lx;
        const int l = (tile_x * (RX - 2) + SNAP / 2) / SNAP * SNAP, rr = ((tile_x + 1) * (RX - 2) + SNAP / 2) / SNAP * SNAP;
        interior = gx >= l && gx < rr && gx < Ws && ly >= 1 && ly < RY - 1 && tile_y * (RY - 2) - 1 + ly < Hs && gx >= 0;
    }
    const size_t frame2 = (size_t)Hs * Ws * 2, plane = (size_t)(T / 2) * frame2;      // in f4 units
    size_t o = (size_t)tp * frame2 + ((size_t)y * Ws + x) * 2;
    size_t oo = (size_t)y * Ws + x;
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < D; ++d, o += plane, oo += (size_t)Hs * Ws) {
        unsigned e = 0;
        if constexpr (OWNER) e = owner[oo];
        f4 v0 = src[o], v1 = src[o + 1];
        v0 += src[o + 2] + src[o + 2 * Ws] + src[o + 2 * Ws + 2];
        v1 += src[o + 3] + src[o + 2 * Ws + 1] + src[o + 2 * Ws + 3];
        if (interior) {
            __builtin_nontemporal_store(v0 + acc, &dst[o]);
            __builtin_nontemporal_store(v1 + acc, &dst[o + 1]);
        }
        acc.x += (float)e;
    }
}

// multi-pixel threads: a 512-thread workgroup owns a (64-2) x (RY-2) tile of 2 frames, thread (col, rowq) handles the pixels of its
// column at rows rowq, rowq + 8, ... (RY / 8 of them) one after the other per plane -- the pattern of a backward whose tiles are 4x
// larger (halo x1.10 at RY = 32 instead of x1.22) at the same workgroup size.
template <int RY, int TAPS, int STORE, int SNAP>
__global__ __launch_bounds__(512) void bwd_like_mp_k(const f4 *__restrict__ src, f4 *__restrict__ dst, const unsigned short *__restrict__ owner, int D,
                                                     int T, int Hs, int Ws, int tiles_x, int tiles_y) {
    constexpr int RX = 64, PPT = RY / 8;
    const int b = blockIdx.x;
    const int q = gridDim.x >> 3, r = gridDim.x & 7, xcd = b & 7, k = b >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    const int tile_x = bid % tiles_x, rest = bid / tiles_x, tile_y = rest % tiles_y, t0 = (rest / tiles_y) * 2;
    const int lx = threadIdx.x & 63, rq = threadIdx.x >> 6;
    const int gx = tile_x * (RX - 2) - 1 + lx, x = min(max(gx, 0), Ws - 2);
    bool xin = lx >= 1 && lx < RX - 1 && gx < Ws;
    if constexpr (STORE >= 2) {
        const int l = (tile_x * (RX - 2) + SNAP / 2) / SNAP * SNAP, rr = ((tile_x + 1) * (RX - 2) + SNAP / 2) / SNAP * SNAP;
        xin = gx >= l && gx < rr && gx < Ws && gx >= 0;
    }
    const size_t frame = (size_t)Hs * Ws, plane = (size_t)T * frame;
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            const int ly = rq + 8 * p, gy = tile_y * (RY - 2) - 1 + ly, y = min(max(gy, 0), Hs - 2);
            const size_t o = (size_t)d * plane + (size_t)t0 * frame + (size_t)y * Ws + x;
            const unsigned e = owner[(size_t)d * frame + (size_t)y * Ws + x];
            f4 v0 = src[o], v1 = src[o + frame];
            if constexpr (TAPS == 4) {
                v0 += src[o + 1] + src[o + Ws] + src[o + Ws + 1];
                v1 += src[o + frame + 1] + src[o + frame + Ws] + src[o + frame + Ws + 1];
            }
            if (xin && ly >= 1 && ly < RY - 1 && gy < Hs) {
                __builtin_nontemporal_store(v0 + acc, &dst[o]);
                __builtin_nontemporal_store(v1 + acc, &dst[o + frame]);
            }
            acc.x += (float)e;
        }
    }
}

template <typename F>
static void run(const char *name, F launch, double bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 4;
    printf("%-64s %7.3f ms  %5.2f TB/s\n", name, ms, bytes / ms / 1e9);
    fflush(stdout);
}

typedef float f2_t __attribute__((ext_vector_type(2)));
__global__ void rw8_k(const f2_t *__restrict__ src, f2_t *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        f2_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u] * 2.f, &dst[i + u * stride]);
    }
}

int main(int argc, char **argv) {
    const bool aligned_only = argc > 1 && (argv[1][0] == 'a' || argv[1][0] == 'h' || argv[1][0] == 'f' || argv[1][0] == 'v' || argv[1][0] == 'p');      // ./rw_bw aligned | halo: only that comparison at the end
    const bool halo_only = argc > 1 && argv[1][0] == 'h';
    const size_t unit = 8ull << 30, n = unit / 16;     // 8 GiB per stream
    f4 *src, *dst;
    if (hipMalloc(&src, 3 * unit) != hipSuccess || hipMalloc(&dst, 3 * unit) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 0, 3 * unit); hipMemset(dst, 0, 3 * unit);
    char name[160];
    if (!aligned_only) {
#define RUN(R, W, NT, U)                                                                                                   \
    for (int bs : {256, 512, 1024})                                                                                         \
        for (int per_cu : {4, 16, 0}) {                                                                                     \
            const size_t grid = per_cu ? 256 * per_cu : (n / ((size_t)bs * U));                                             \
            snprintf(name, sizeof name, "rw_k  %d read : %d write  %s  unroll %d  block %4d  grid %8zu", R, W, NT ? "nt-store" : "store   ", U, bs, grid); \
            run(name, [&] { hipLaunchKernelGGL((rw_k<R, W, NT, U>), dim3((unsigned)grid), dim3(bs), 0, 0, src, dst, n); }, (double)(R + W) * unit); \
        }
    RUN(1, 1, true, 4) RUN(1, 1, false, 4) RUN(1, 1, true, 1) RUN(1, 1, true, 8)
    RUN(2, 1, true, 4) RUN(2, 1, false, 4)
    RUN(3, 2, true, 4) RUN(1, 2, true, 4)
    // render-shaped: D=32, T=50, 720 x 1280 (cfg3): 23.6 GB in, 23.6 GB out
    {
        const int D = 32, T = 12, Hs = 720, Ws = 1280;      // T = 12: 5.7 GB per direction, inside the 24 GiB buffers
        const int tx = Ws / 32, ty = Hs / 16;
        const double bytes = 2.0 * D * T * Hs * Ws * 16;
        const unsigned grid = tx * ty * (T / 2);
        run("render_like  nt-store  1 plane in flight", [&] { hipLaunchKernelGGL((render_like_k<true, 1>), dim3(grid), dim3(512), 0, 0, src, dst, D, T, Hs, Ws, tx, ty); }, bytes);
        run("render_like  nt-store  2 planes in flight", [&] { hipLaunchKernelGGL((render_like_k<true, 2>), dim3(grid), dim3(512), 0, 0, src, dst, D, T, Hs, Ws, tx, ty); }, bytes);
        run("render_like  nt-store  4 planes in flight", [&] { hipLaunchKernelGGL((render_like_k<true, 4>), dim3(grid), dim3(512), 0, 0, src, dst, D, T, Hs, Ws, tx, ty); }, bytes);
        run("render_like  store     2 planes in flight", [&] { hipLaunchKernelGGL((render_like_k<false, 2>), dim3(grid), dim3(512), 0, 0, src, dst, D, T, Hs, Ws, tx, ty); }, bytes);
    }
    // the backward's pattern: region shape x frames per workgroup x taps, cfg3 geometry with T = 12 frames
    {
        const int D = 32, T = 12, Hs = 720, Ws = 1280;
        const double bytes = 2.0 * D * T * Hs * Ws * 16;      // algorithmic: every texel read once, every gradient texel written once
        unsigned short *owner;
        hipMalloc(&owner, (size_t)D * Hs * Ws * 2 + 4096);
        hipMemset(owner, 0, (size_t)D * Hs * Ws * 2 + 4096);
#define BL(RX, RY, FR, TAPS, OWN)                                                                                              \
        {                                                                                                                      \
            const int tx = (Ws + RX - 3) / (RX - 2), ty = (Hs + RY - 3) / (RY - 2);                                            \
            snprintf(name, sizeof name, "bwd_like  region %3d x %2d  frames %d  taps %d  owner %d  (halo x%.2f)", RX, RY, FR, TAPS, OWN,         \
                     (double)RX * RY / ((RX - 2) * (RY - 2)));                                                                 \
            run(name, [&] { hipLaunchKernelGGL((bwd_like_k<RX, RY, FR, TAPS, OWN>), dim3((unsigned)(tx * ty * (T / FR))), dim3(RX * RY), 0, 0, src, dst, \
                                               owner, D, T, Hs, Ws, tx, ty); }, bytes);                                        \
        }
        BL(32, 16, 2, 1, false) BL(32, 16, 2, 4, false) BL(32, 16, 2, 4, true)
        BL(64, 8, 2, 4, true) BL(64, 16, 1, 4, true) BL(64, 16, 2, 4, true) BL(32, 32, 1, 4, true) BL(32, 32, 2, 4, true)
        BL(128, 4, 2, 4, true) BL(16, 32, 2, 4, true) BL(64, 8, 1, 4, true) BL(32, 16, 1, 4, true) BL(32, 8, 2, 4, true) BL(32, 16, 4, 4, true)
#define BLS(RX, RY, FR, ST, SNAP)                                                                                              \
        {                                                                                                                      \
            const int tx = (Ws + RX - 3) / (RX - 2), ty = (Hs + RY - 3) / (RY - 2);                                            \
            snprintf(name, sizeof name, "bwd_like  region %3d x %2d  frames %d  store mode %d  snap %d", RX, RY, FR, ST, SNAP);                 \
            run(name, [&] { hipLaunchKernelGGL((bwd_like_k<RX, RY, FR, 4, true, ST, SNAP>), dim3((unsigned)(tx * ty * (T / FR))), dim3(RX * RY), 0, 0, src, dst, \
                                               owner, D, T, Hs, Ws, tx, ty); }, bytes);                                        \
        }
#define BLI(RX, RY, ST, SNAP)                                                                                                  \
        {                                                                                                                      \
            const int tx = (Ws + RX - 3) / (RX - 2), ty = (Hs + RY - 3) / (RY - 2);                                            \
            snprintf(name, sizeof name, "bwd_like INTERLEAVED pairs  region %3d x %2d  store mode %d  snap %d", RX, RY, ST, SNAP);               \
            run(name, [&] { hipLaunchKernelGGL((bwd_like_il_k<RX, RY, true, ST, SNAP>), dim3((unsigned)(tx * ty * (T / 2))), dim3(RX * RY), 0, 0, src, dst, \
                                               owner, D, T, Hs, Ws, tx, ty); }, bytes);                                        \
        }
#define BLM(RY, ST, SNAP)                                                                                                      \
        {                                                                                                                      \
            const int tx = (Ws + 61) / 62, ty = (Hs + RY - 3) / (RY - 2);                                                      \
            snprintf(name, sizeof name, "bwd_like MULTI-PIXEL threads  region  64 x %2d  frames 2  store mode %d  snap %d  (halo x%.2f)", RY, ST, SNAP, 64.0 * RY / (62.0 * (RY - 2))); \
            run(name, [&] { hipLaunchKernelGGL((bwd_like_mp_k<RY, 4, ST, SNAP>), dim3((unsigned)(tx * ty * (T / 2))), dim3(512), 0, 0, src, dst, \
                                               owner, D, T, Hs, Ws, tx, ty); }, bytes);                                        \
        }
        BLM(8, 0, 8) BLM(16, 0, 8) BLM(32, 0, 8) BLM(64, 0, 8) BLM(16, 2, 4) BLM(32, 2, 4) BLM(64, 2, 4)
        BLI(32, 16, 0, 8) BLI(32, 16, 2, 4) BLI(64, 8, 0, 8) BLI(64, 8, 2, 4) BLI(64, 16, 0, 8) BLI(32, 8, 0, 8) BLI(32, 8, 2, 4) BLI(128, 4, 0, 8)
        BLS(32, 16, 2, 0, 8) BLS(32, 16, 2, 1, 8) BLS(32, 16, 2, 2, 4) BLS(32, 16, 2, 2, 8) BLS(32, 16, 2, 3, 8) BLS(32, 16, 2, 2, 16)
        BLS(64, 8, 2, 0, 8) BLS(64, 8, 2, 1, 8) BLS(64, 8, 2, 2, 4) BLS(64, 8, 2, 2, 8) BLS(64, 8, 2, 3, 8) BLS(64, 8, 2, 2, 16)
        BLS(64, 16, 1, 0, 8) BLS(64, 16, 1, 1, 8) BLS(64, 16, 1, 2, 8) BLS(64, 16, 1, 3, 8)
        BLS(64, 8, 1, 0, 8) BLS(64, 8, 1, 1, 8) BLS(64, 8, 1, 2, 8) BLS(128, 8, 1, 0, 8) BLS(128, 8, 1, 2, 8) BLS(128, 4, 1, 0, 8) BLS(128, 4, 1, 2, 8)
    }
    }
    // The fused optimiser epilogue against what it replaces (round 4): ./rw_bw fused
    if (argc > 1 && argv[1][0] == 'f') {
        const int D = 32, T = 12, Hs = 720, Ws = 1280;
        const size_t texels = (size_t)D * T * Hs * Ws;
        unsigned short *owner;
        hipMalloc(&owner, (size_t)D * Hs * Ws * 2 + 4096);
        hipMemset(owner, 0, (size_t)D * Hs * Ws * 2 + 4096);
        for (int rep = 0; rep < 2; ++rep) {
            {
                const int tx = (Ws + 29) / 30, ty = (Hs + 13) / 14;
                run("backward pattern 32x16x2, gradient STORED (1 read + 1 write stream)            [bytes: 2 streams]",
                    [&] { hipLaunchKernelGGL((bwd_like_k<32, 16, 2, 4, true, 0, 8, 1>), dim3((unsigned)(tx * ty * (T / 2))), dim3(512), 0, 0, src, dst, owner, D, T, Hs, Ws, tx, ty); },
                    2.0 * texels * 16);
                run("backward pattern 32x16x2, Adam IN the owner store (3 read + 3 write streams)    [bytes: 6 streams]",
                    [&] { hipLaunchKernelGGL((bwd_like_adam_k<32, 16, 2>), dim3((unsigned)(tx * ty * (T / 2))), dim3(512), 0, 0, src, dst, owner, D, T, Hs, Ws, tx, ty, n); },
                    6.0 * texels * 16);
                run("backward pattern 32x16x2, Adam IN the owner store, PLAIN (write-back) stores       [bytes: 6 streams]",
                    [&] { hipLaunchKernelGGL((bwd_like_adam_k<32, 16, 2, false>), dim3((unsigned)(tx * ty * (T / 2))), dim3(512), 0, 0, src, dst, owner, D, T, Hs, Ws, tx, ty, n); },
                    6.0 * texels * 16);
                const int tx1 = (Ws + 61) / 62;
                run("backward pattern 64x16x1, Adam IN the owner store (3 read + 3 write streams)    [bytes: 6 streams]",
                    [&] { hipLaunchKernelGGL((bwd_like_adam_k<64, 16, 1>), dim3((unsigned)(tx1 * ty * T)), dim3(1024), 0, 0, src, dst, owner, D, T, Hs, Ws, tx1, ty, n); },
                    6.0 * texels * 16);
                run("backward pattern 64x16x2, gradient STORED (1 read + 1 write stream)            [bytes: 2 streams]",
                    [&] { hipLaunchKernelGGL((bwd_like_k<64, 16, 2, 4, true, 0, 8, 1>), dim3((unsigned)(tx1 * ty * (T / 2))), dim3(1024), 0, 0, src, dst, owner, D, T, Hs, Ws, tx1, ty); },
                    2.0 * texels * 16);
                run("backward pattern 64x16x2, Adam IN the owner store (3 read + 3 write streams)    [bytes: 6 streams]",
                    [&] { hipLaunchKernelGGL((bwd_like_adam_k<64, 16, 2>), dim3((unsigned)(tx1 * ty * (T / 2))), dim3(1024), 0, 0, src, dst, owner, D, T, Hs, Ws, tx1, ty, n); },
                    6.0 * texels * 16);
                const int ty8 = (Hs + 5) / 6;
                run("backward pattern 64x8x2, Adam IN the owner store (3 read + 3 write streams)     [bytes: 6 streams]",
                    [&] { hipLaunchKernelGGL((bwd_like_adam_k<64, 8, 2>), dim3((unsigned)(tx1 * ty8 * (T / 2))), dim3(512), 0, 0, src, dst, owner, D, T, Hs, Ws, tx1, ty8, n); },
                    6.0 * texels * 16);
            }
            {
                const size_t grid = texels / (256 * 4);
                run("separate step kernel over the same texels (4 read + 3 write streams, whole rows) [bytes: 7 streams]",
                    [&] { hipLaunchKernelGGL((rw_k<4, 3, true, 4>), dim3((unsigned)grid), dim3(256), 0, 0, src, dst, texels); }, 7.0 * texels * 16);
            }
        }
        printf("fused = row 2;  unfused = row 1 + row 3\n");
        return 0;
    }
    // The fp16 stack's backward pattern against the fp32 one, same geometry (round 6): ./rw_bw p
    if (argc > 1 && argv[1][0] == 'p') {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const int D = 32, T = 12, Hs = 720, Ws = 1280;
        const size_t texels = (size_t)D * T * Hs * Ws;
        unsigned short *owner;
        hipMalloc(&owner, (size_t)D * Hs * Ws * 2 + 4096);
        hipMemset(owner, 0, (size_t)D * Hs * Ws * 2 + 4096);
#define BLE(E, EB, RX, RY, FR)                                                                                                 \
        {                                                                                                                      \
            const int tx = (Ws + RX - 3) / (RX - 2), ty = (Hs + RY - 3) / (RY - 2);                                            \
            snprintf(name, sizeof name, "bwd_like  %2d-byte texels  region %3d x %2d  frames %d  (halo x%.2f)", EB, RX, RY, FR,  \
                     (double)RX * RY / ((RX - 2) * (RY - 2)));                                                                 \
            run(name, [&] { hipLaunchKernelGGL((bwd_like_e_k<E, RX, RY, FR>), dim3((unsigned)(tx * ty * (T / FR))), dim3(RX * RY), 0, 0, \
                                               reinterpret_cast<const E *>(src), reinterpret_cast<E *>(dst), owner, D, T, Hs, Ws, tx, ty); }, 2.0 * texels * EB); \
        }
        for (int rep = 0; rep < 2; ++rep) {
            BLE(f4, 16, 32, 16, 2) BLE(f2, 8, 32, 16, 2) BLE(f2, 8, 32, 16, 4) BLE(f2, 8, 64, 8, 2) BLE(f2, 8, 64, 16, 1) BLE(f2, 8, 64, 16, 2) BLE(f2, 8, 64, 8, 4) BLE(f2, 8, 128, 4, 2)
        }
        // (streaming reference at 8 bytes per lane: what the HBM gives a read:write = 1:1 kernel whose accesses are 8-byte wide)
        run("rw 1:1, 8-byte lanes, nt-store (grid-strided)", [&] { hipLaunchKernelGGL((rw8_k), dim3(256 * 16), dim3(512), 0, 0, reinterpret_cast<const f2 *>(src),
                                                                                    reinterpret_cast<f2 *>(dst), unit / 8); }, 2.0 * unit);
        return 0;
    }
    // A backward WITHOUT the vertical halo (round 5): the shipped pattern against its no-vertical-halo upper bound, same geometry.
    if (argc > 1 && argv[1][0] == 'v') {
        const int D = 32, T = 12, Hs = 720, Ws = 1280;
        const double bytes = 2.0 * D * T * Hs * Ws * 16;
        unsigned short *owner;
        hipMalloc(&owner, (size_t)D * Hs * Ws * 2 + 4096);
        hipMemset(owner, 0, (size_t)D * Hs * Ws * 2 + 4096);
#define BLV(RX, RY, FR, HY)                                                                                                    \
        {                                                                                                                      \
            const int tx = (Ws + RX - 3) / (RX - 2), ty = (Hs + RY - 2 * HY - 1) / (RY - 2 * HY);                              \
            snprintf(name, sizeof name, "bwd_like  region %3d x %2d  frames %d  vertical halo %d  (x%.2f pixels swept per pixel owned)", RX, RY, FR, HY, \
                     (double)RX * RY / ((RX - 2) * (RY - 2 * HY)));                                                            \
            run(name, [&] { hipLaunchKernelGGL((bwd_like_k<RX, RY, FR, 4, true, 0, 8, 1, HY>), dim3((unsigned)(tx * ty * (T / FR))), dim3(RX * RY), 0, 0, \
                                               src, dst, owner, D, T, Hs, Ws, tx, ty); }, bytes);                              \
        }
        for (int rep = 0; rep < 3; ++rep) {
            BLV(32, 16, 2, 1) BLV(32, 16, 2, 0) BLV(64, 16, 1, 1) BLV(64, 16, 1, 0) BLV(64, 8, 2, 1) BLV(64, 8, 2, 0)
        }
        return 0;
    }
    // Halo-atomics backward against the shipped halo pattern (round 4), same geometry, algorithmic bytes as the unit.
    if (halo_only) {
        const int D = 32, T = 12, Hs = 720, Ws = 1280;
        const double bytes = 2.0 * D * T * Hs * Ws * 16;
        unsigned short *owner;
        hipMalloc(&owner, (size_t)D * Hs * Ws * 2 + 4096);
        hipMemset(owner, 0, (size_t)D * Hs * Ws * 2 + 4096);
#define BLH(RX, RY, FR, ZERO)                                                                                                  \
        {                                                                                                                      \
            const int tx = (Ws + RX - 1) / RX, ty = (Hs + RY - 1) / RY;                                                        \
            snprintf(name, sizeof name, "bwd_like HALO-ATOMICS  region %3d x %2d  frames %d  ring = %4.1f %% of the texels  zero fill %s", RX, RY, FR, \
                     100.0 * (2.0 * RX + 2.0 * RY - 4) / (RX * RY), ZERO ? "timed" : "free ");                                  \
            run(name, [&] { if (ZERO) hipLaunchKernelGGL((ha_zero_ring_k<RX, RY>), dim3((Ws + 63) / 64, (Hs + 3) / 4, D), dim3(256), 0, 0, dst, D, T, Hs, Ws); \
                            hipLaunchKernelGGL((bwd_like_ha_k<RX, RY, FR, 4>), dim3((unsigned)(tx * ty * (T / FR))), dim3(RX * RY), 0, 0, src, dst, \
                                               owner, D, T, Hs, Ws, tx, ty); }, bytes);                                        \
        }
#define BLREF(RX, RY, FR)                                                                                                      \
        {                                                                                                                      \
            const int tx = (Ws + RX - 3) / (RX - 2), ty = (Hs + RY - 3) / (RY - 2);                                            \
            snprintf(name, sizeof name, "bwd_like  shipped halo pattern  region %3d x %2d  frames %d  (x%.2f pixels swept per pixel owned)", RX, RY, FR, \
                     (double)RX * RY / ((RX - 2) * (RY - 2)));                                                                 \
            run(name, [&] { hipLaunchKernelGGL((bwd_like_k<RX, RY, FR, 4, true, 0, 8, 1>), dim3((unsigned)(tx * ty * (T / FR))), dim3(RX * RY), 0, 0, \
                                               src, dst, owner, D, T, Hs, Ws, tx, ty); }, bytes);                              \
        }
        for (int rep = 0; rep < 2; ++rep) {
            BLREF(32, 16, 2) BLH(32, 16, 2, false) BLH(32, 16, 2, true)
            BLREF(64, 16, 1) BLH(64, 16, 1, false) BLH(64, 16, 1, true)
            BLREF(64, 8, 2) BLH(64, 8, 2, false) BLH(64, 8, 2, true)
            BLH(32, 32, 1, false) BLH(32, 32, 1, true)
        }
        return 0;
    }
    // Aligned ownership priced WITH the halo it needs (round 3): segments snapped to SNAP-texel columns move by up to SNAP / 2, so the
    // region stages 1 + SNAP / 2 halo columns per side and owns RX - 2 - SNAP columns: more regions, more tap re-reads.
    {
        const int D = 32, T = 12, Hs = 720, Ws = 1280;
        const double bytes = 2.0 * D * T * Hs * Ws * 16;
        unsigned short *owner;
        hipMalloc(&owner, (size_t)D * Hs * Ws * 2 + 4096);
        hipMemset(owner, 0, (size_t)D * Hs * Ws * 2 + 4096);
#define BLA(RX, RY, FR, ST, SNAP, HX)                                                                                          \
        {                                                                                                                      \
            const int tx = (Ws + RX - 2 * HX - 1) / (RX - 2 * HX), ty = (Hs + RY - 3) / (RY - 2);                              \
            snprintf(name, sizeof name, "bwd_like  region %3d x %2d  frames %d  store mode %d  snap %2d  x-halo %d (x%.2f pixels swept per pixel owned)", \
                     RX, RY, FR, ST, SNAP, HX, (double)RX * RY / ((RX - 2 * HX) * (RY - 2)));                                  \
            run(name, [&] { hipLaunchKernelGGL((bwd_like_k<RX, RY, FR, 4, true, ST, SNAP, HX>), dim3((unsigned)(tx * ty * (T / FR))), dim3(RX * RY), 0, 0, \
                                               src, dst, owner, D, T, Hs, Ws, tx, ty); }, bytes);                              \
        }
        for (int rep = 0; rep < 2; ++rep) {
            BLA(32, 16, 2, 0, 8, 1)      // shipped pattern: ragged 30-texel segments
            BLA(32, 16, 2, 2, 4, 1)      // round 2's row: snapped stores WITHOUT the halo they need (not buildable)
            BLA(32, 16, 2, 0, 4, 3)      // the wider halo alone (ragged 26-texel segments)
            BLA(32, 16, 2, 2, 4, 3)      // aligned ownership as a kernel could build it: 26 of 32 columns owned, 64-byte aligned ends
            BLA(32, 16, 2, 2, 8, 5)      // 128-byte aligned ends: 22 of 32
            BLA(64, 8, 2, 0, 8, 1) BLA(64, 8, 2, 2, 4, 3) BLA(64, 8, 2, 2, 8, 5)
            BLA(64, 16, 1, 0, 8, 1) BLA(64, 16, 1, 2, 4, 3) BLA(64, 16, 1, 2, 8, 5)
        }
    }
    return 0;
}
