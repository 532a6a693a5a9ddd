// Crop-aware Adam for the dense plane stack: the optimiser half of a stage-2 iteration (train_3dvid.py:263-290, MPV.py:199-214).
//
// A training iteration renders ONE crop (180 x 320 of 360 x 640, configs/mpv_base.txt:21-24): only the texels inside the crop's
// parallax footprint -- about a quarter of every plane -- receive a gradient.  torch.optim.Adam still streams (p, g, m, v) over the
// whole stack, because a texel with zero gradient keeps moving on its momentum: m <- b1 m, v <- b2 v, p <- p - lr_t m^/(sqrt(v^)+eps).
// That tail is a pure function of (p, m, v) and the per-step scalars, so it can be applied LATER, exactly: every 8 x 8-texel tile
// of every plane remembers the last step it is current for, and
//   * adam_window_catchup_k (before the render) replays, in registers, the zero-gradient steps the tiles of the coming crop's
//     window have missed -- the same fp32 operations in the same order as the dense update with g = 0 -- and hands the render a
//     compact copy of the window with the CURRENT parameters (the render then writes a compact gradient: no zero fill of the
//     other 3/4 either); nothing is written back to the stack at this point;
//   * adam_window_step_k (after the backward) replays the same missed steps again (registers) and applies the current step to
//     the window from the compact gradient: one read and one write of (p, m, v) per iteration;
//   * a catch-up over the full stack brings everything current (checkpoints, lod(), evaluation renders).
// Per-step scalars (lr / (1 - b1^t), sqrt(1 - b2^t)) come from a device table written by the host when the step is taken, so a
// changing learning rate (train_3dvid.py:263-277) is replayed as it was.
#include "vl3d_adam.h"

namespace {

using namespace vl3d_adam;      // TS (the bookkeeping tile side), adam_upd, replay: vl3d_adam.h -- shared with the render backward's fused store

struct Win { int y0, x0, wh, ww; };

// Per-plane boxes (optional, [D] x (y0, y1, x0, x1) in plane texels, tile aligned, inside the window): the window of a crop is ONE box
// for all planes, the union of their footprints; a plane's own footprint is smaller by the parallax between the planes.  Texels of the
// window outside their plane's box cannot be sampled in this iteration: their gradient is exactly zero, so their update stays deferred
// like that of every texel outside the window.  The table travels in the kernel arguments (16 bytes per plane, read through the scalar
// cache with the uniform plane index): no device buffer, no host-to-device copy, no synchronisation in the training loop.
constexpr int MAX_BOX_PLANES = 128;
struct BoxTable { int n; int4 b[MAX_BOX_PLANES]; };
__device__ __forceinline__ bool outside_box(const BoxTable &boxes, int d, int x, int y) {
    if (!boxes.n) return false;
    const int4 b = boxes.b[d];
    return y < b.x || y >= b.y || x < b.z || x >= b.w;
}
static BoxTable make_boxes(const int32_t *host_boxes, int D) {
    BoxTable t;
    t.n = (host_boxes && D <= MAX_BOX_PLANES) ? D : 0;       // more planes than the table holds: the whole window for every plane
    for (int d = 0; d < t.n; ++d) t.b[d] = make_int4(host_boxes[4 * d], host_boxes[4 * d + 1], host_boxes[4 * d + 2], host_boxes[4 * d + 3]);
    return t;
}

// PACKED storage of a tile-culled model (the reference stores static quads once, dynamic quads per frame and culled quads not at all:
// MPI.py:364-436, MPV.py:235-288).  Parameters and moments live in a pool of 8 x 8-texel blocks (the bookkeeping tiles): blocks [D][tiles_y][tiles_x]
// int32 = -1 for a block no kept quad can read (no storage), else slot << 1 | dynamic -- a static block (only static quads can read it) is ONE
// slot of 64 texels, a dynamic block T consecutive slots (frame-major).  A texel's class (texel_class) and everything computed from it
// are those of the dense layout: only the address changes, so training from the pool gives the dense model's bits.  blocks == NULL: the
// dense (D,T,Hs,Ws,4) stack.
struct Layout { const int *blocks; };
// index (in texels) of frame 0 of texel (d, y, x) and the stride between its frames (0: one shared copy)
__device__ __forceinline__ void texel_slot(const Layout &L, int d, int y, int x, int T, int Hs, int Ws, int tiles_y, int tiles_x, size_t &o, size_t &fs) {
    if (!L.blocks) {
        fs = (size_t)Hs * Ws;
        o = (size_t)d * T * fs + (size_t)y * Ws + x;
        return;
    }
    const int e = L.blocks[((size_t)d * tiles_y + y / TS) * tiles_x + x / TS];      // (>= 0 for every texel that is a parameter)
    o = (size_t)(e >> 1) * (TS * TS) + (size_t)((y % TS) * TS + (x % TS));
    fs = (e & 1) ? (size_t)(TS * TS) : 0;
}

// one thread per window texel and plane, looping over the frames.  `upto`: the step the window must be current for (the
// step about to be taken minus one).  compact (optional): (D,T,wh,ww,4) copy of the window's parameters after the catch-up.
// (115 VGPRs, four waves per SIMD.  Round 6 measured amdgpu_waves_per_eu 5 / 6 / 8 on the tile-culled schedule: 0.243 / 0.244 / 0.278 ms against 0.240 --
// occupancy is not what bounds a tile-culled window's catch-up; docs/kernels/K6_optimiser.md)
template <bool FLUSH>
__global__ __launch_bounds__(256) void adam_window_catchup_k(int T, int Hs, int Ws, Win w, float4 *__restrict__ p, float4 *__restrict__ m,
                                                             float4 *__restrict__ v, const int *__restrict__ last_step, int tiles_y, int tiles_x,
                                                             const float2 *__restrict__ hist, int upto, float beta1, float beta2, float eps,
                                                             float4 *__restrict__ compact, Quads q, float culled_alpha, int flags,
                                                             int writeback, const BoxTable boxes, Layout lay) {
    const int lx = blockIdx.x * 64 + (threadIdx.x & 63), ly = blockIdx.y * 4 + (threadIdx.x >> 6), d = blockIdx.z;
    if (lx >= w.ww || ly >= w.wh) return;
    const int x = w.x0 + lx, y = w.y0 + ly;
    const int mirror = flags & 1;
    // flags bit 1 ("lean"): the caller's compact buffer holds finite values everywhere already (a persistent buffer, zero-filled once), so
    // the slots the render cannot read with a non-zero weight -- texels outside their plane's box, culled texels -- are left alone: for a
    // tile-culled model the dense compact window was 2.3 GB of stores per iteration for the 16 % of it that are parameters
    const bool lean = (flags & 2) != 0;
    if (outside_box(boxes, d, x, y)) {           // this plane's taps cannot reach the texel: its slots of the compact copy are never read
        if (compact && !lean) {                  // (zeros all the same: nothing uninitialised for a later reader to trip over)
            size_t oc = (size_t)d * T * w.wh * w.ww + (size_t)ly * w.ww + lx;
            for (int t = 0; t < T; ++t, oc += (size_t)w.wh * w.ww) compact[oc] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const int from = last_step[((size_t)d * tiles_y + y / TS) * tiles_x + x / TS];
    const size_t cframe = (size_t)w.wh * w.ww;
    size_t oc = (size_t)d * T * cframe + (size_t)ly * w.ww + lx;
    const int cls = texel_class(q, d, x, y, Hs, Ws);
    if (cls == 0) {           // culled: not a parameter; the render must see it transparent (and finite)
        if (compact && !lean)
            for (int t = 0; t < T; ++t, oc += cframe) compact[oc] = make_float4(0.f, 0.f, 0.f, culled_alpha);
        return;
    }
    size_t o, frame;
    texel_slot(lay, d, y, x, T, Hs, Ws, tiles_y, tiles_x, o, frame);
    if constexpr (FLUSH) {      // a FLUSH (written back in place, no compact copy): per class, as every round so far
        if (cls == 2) {           // static: the one parameter lives in frame 0
            float4 pp = p[o];
            if (from < upto) {
                float4 mm = m[o], vv = v[o];
                replay(pp, mm, vv, hist, from, upto, beta1, beta2, eps);
                p[o] = pp; m[o] = mm; v[o] = vv;
            }
            if (compact)
                for (int t = 0; t < T; ++t, oc += cframe) compact[oc] = pp;
            if (mirror && frame)  // refresh the other frames' slots so that the stack reads consistently everywhere (a static block has one copy)
                for (int t = 1; t < T; ++t) p[o + (size_t)t * frame] = pp;
            return;
        }
        if (from >= upto) {       // current already: copy only
            if (compact)
                for (int t = 0; t < T; ++t, o += frame, oc += cframe) compact[oc] = p[o];
            return;
        }
        int t = 0;
        constexpr int NF = 4;
        for (; t + NF <= T; t += NF, o += NF * frame, oc += NF * cframe) {
            float4 pp[NF], mm[NF], vv[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) { pp[f] = p[o + f * frame]; mm[f] = m[o + f * frame]; vv[f] = v[o + f * frame]; }
            for (int s = from + 1; s <= upto; ++s) {
                const float2 h = hist[s];
#pragma unroll
                for (int f = 0; f < NF; ++f) adam_upd4(pp[f], make_float4(0.f, 0.f, 0.f, 0.f), mm[f], vv[f], h.x, beta1, beta2, eps, h.y);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                p[o + f * frame] = pp[f]; m[o + f * frame] = mm[f]; v[o + f * frame] = vv[f];
                if (compact) compact[oc + f * cframe] = pp[f];
            }
        }
        for (; t < T; ++t, o += frame, oc += cframe) {
            float4 pp = p[o], mm = m[o], vv = v[o];
            replay(pp, mm, vv, hist, from, upto, beta1, beta2, eps);
            p[o] = pp; m[o] = mm; v[o] = vv;
            if (compact) compact[oc] = pp;
        }
        return;
    }
    // The render's catch-up (compact copy only).  A wave of a tile-culled window mixes the classes -- static next to dynamic quads every 11 texels, tiles
    // that are current next to tiles that are behind every 8 -- and ran the three per-class loops (static: T stores; current: T copies; behind: T / 4
    // replay trips) one after the other, each for its lanes only.  ONE loop now, four frames per trip, for every lane: a static lane replays its one
    // parameter first and then only stores; a current lane skips the moment loads (and the replay loop's zero trips); the operations per value are the
    // per-class loops' (the same bits).
    const bool dyn = cls == 1;
    float4 ps = make_float4(0.f, 0.f, 0.f, 0.f);
    int from_l = from;
    if (!dyn) {
        ps = p[o];
        if (from < upto) {
            float4 mm = m[o], vv = v[o];
            replay(ps, mm, vv, hist, from, upto, beta1, beta2, eps);
        }
        from_l = upto;
    }
#ifdef VL3D_CATCHUP_ABLATE      // measurement only (profiles/r06_pmc_catchup.sh): 1 = no replay (the loads and stores alone)
    if (VL3D_CATCHUP_ABLATE & 1) from_l = upto;
#endif
    const bool behind = from_l < upto;
    int t = 0;
    constexpr int NF = 4;
    for (; t + NF <= T; t += NF, o += NF * frame, oc += NF * cframe) {
        float4 pp[NF], mm[NF], vv[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            pp[f] = dyn ? p[o + f * frame] : ps;
            if (behind) { mm[f] = m[o + f * frame]; vv[f] = v[o + f * frame]; }
            else mm[f] = vv[f] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int s = from_l + 1; s <= upto; ++s) {
            const float2 h = hist[s];
#pragma unroll
            for (int f = 0; f < NF; ++f) adam_upd4(pp[f], make_float4(0.f, 0.f, 0.f, 0.f), mm[f], vv[f], h.x, beta1, beta2, eps, h.y);
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) compact[oc + f * cframe] = pp[f];
    }
    for (; t < T; ++t, o += frame, oc += cframe) {
        float4 pp = dyn ? p[o] : ps, mm = make_float4(0.f, 0.f, 0.f, 0.f), vv = mm;
        if (behind) { mm = m[o]; vv = v[o]; }
        replay(pp, mm, vv, hist, from_l, upto, beta1, beta2, eps);
        compact[oc] = pp;
    }
}

__global__ __launch_bounds__(256) void adam_window_step_k(int T, int Hs, int Ws, Win w, float4 *__restrict__ p, const float4 *__restrict__ g,
                                                          float4 *__restrict__ m, float4 *__restrict__ v, float lr_bc1, float beta1, float beta2,
                                                          float eps, float bc2s, Quads q, int static_tied, const int *__restrict__ last_step,
                                                          int tiles_y, int tiles_x, const float2 *__restrict__ hist, int step,
                                                          const BoxTable boxes, Layout lay, const int *__restrict__ dyn_stepped) {
    // (the tail of vl3d_render_bwd_adam: the owner-computes backward stepped the DYNAMIC texels itself unless its device-side plan said infeasible)
    const bool skip_dyn = dyn_stepped && *dyn_stepped;
    if (skip_dyn && !q.keep) return;
    const int lx = blockIdx.x * 64 + (threadIdx.x & 63), ly = blockIdx.y * 4 + (threadIdx.x >> 6), d = blockIdx.z;
    if (lx >= w.ww || ly >= w.wh) return;
    if (outside_box(boxes, d, w.x0 + lx, w.y0 + ly)) return;     // zero gradient by construction: the update stays deferred
    // the zero-gradient steps this texel's tile has not seen yet are replayed HERE, in front of the real step: the catch-up before
    // the render only computed the current parameters for the compact copy and wrote nothing back (3 write streams fewer)
    const int from = last_step[((size_t)d * tiles_y + (w.y0 + ly) / TS) * tiles_x + (w.x0 + lx) / TS];
    const size_t cframe = (size_t)w.wh * w.ww;
    size_t oc = (size_t)d * T * cframe + (size_t)ly * w.ww + lx;
    const int cls = texel_class(q, d, w.x0 + lx, w.y0 + ly, Hs, Ws);
    if (cls == 0 || (cls == 1 && skip_dyn)) return;
    size_t o, frame;
    texel_slot(lay, d, w.y0 + ly, w.x0 + lx, T, Hs, Ws, tiles_y, tiles_x, o, frame);
    if (cls == 2) {           // static: gradient = the sum over the frames (frame order: deterministic), one update, one write
        float4 gg = g[oc];
        if (!static_tied) {      // static_tied: frame 0 already holds the frame sum (tie_static_grad)
            // ten frames' loads in flight, then their ten adds IN FRAME ORDER (the same sum as one load and one add per trip, which walked the T frames
            // with a full memory round trip each: this loop is the step kernel of a tile-culled model, half of whose kept texels are static)
            constexpr int NS = 10;
            int t = 1;
            for (; t + NS <= T; t += NS) {
                float4 gt[NS];
#pragma unroll
                for (int f = 0; f < NS; ++f) gt[f] = g[oc + (size_t)(t + f) * cframe];
#pragma unroll
                for (int f = 0; f < NS; ++f) { gg.x += gt[f].x; gg.y += gt[f].y; gg.z += gt[f].z; gg.w += gt[f].w; }
            }
            for (; t < T; ++t) {
                const float4 gt = g[oc + (size_t)t * cframe];
                gg.x += gt.x; gg.y += gt.y; gg.z += gt.z; gg.w += gt.w;
            }
        }
        float4 pp = p[o], mm = m[o], vv = v[o];
        replay(pp, mm, vv, hist, from, step - 1, beta1, beta2, eps);
        adam_upd(pp.x, gg.x, mm.x, vv.x, lr_bc1, beta1, beta2, eps, bc2s);
        adam_upd(pp.y, gg.y, mm.y, vv.y, lr_bc1, beta1, beta2, eps, bc2s);
        adam_upd(pp.z, gg.z, mm.z, vv.z, lr_bc1, beta1, beta2, eps, bc2s);
        adam_upd(pp.w, gg.w, mm.w, vv.w, lr_bc1, beta1, beta2, eps, bc2s);
        p[o] = pp; m[o] = mm; v[o] = vv;
        return;
    }
    for (int t = 0; t < T; ++t, o += frame, oc += cframe) {
        float4 pp = p[o], mm = m[o], vv = v[o];
        replay(pp, mm, vv, hist, from, step - 1, beta1, beta2, eps);
        const float4 gg = g[oc];
        adam_upd(pp.x, gg.x, mm.x, vv.x, lr_bc1, beta1, beta2, eps, bc2s);
        adam_upd(pp.y, gg.y, mm.y, vv.y, lr_bc1, beta1, beta2, eps, bc2s);
        adam_upd(pp.z, gg.z, mm.z, vv.z, lr_bc1, beta1, beta2, eps, bc2s);
        adam_upd(pp.w, gg.w, mm.w, vv.w, lr_bc1, beta1, beta2, eps, bc2s);
        p[o] = pp; m[o] = mm; v[o] = vv;
    }
}

// the host's box table as a device array (the render backward's fused store reads it per texel: kernel arguments of ITS launch are taken)
__global__ __launch_bounds__(MAX_BOX_PLANES) void write_boxes_k(const BoxTable boxes, int4 *out) {
    if ((int)threadIdx.x < boxes.n) out[threadIdx.x] = boxes.b[threadIdx.x];
}

// (hist_row: a step's marks also put the step's scalars into its row of the history table -- what the catch-ups of LATER steps replay it with; the
// host used to upload the row with an 8-byte fill of its own, one launch per iteration)
__global__ __launch_bounds__(256) void mark_tiles_k(int *last_step, int tiles_y, int tiles_x, int ty0, int tx0, int nty, int ntx, int D, int step,
                                                    const BoxTable boxes, float2 *hist_row = nullptr, float2 hist_val = make_float2(0.f, 0.f)) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && hist_row) *hist_row = hist_val;
    if (i >= D * nty * ntx) return;
    const int tx = i % ntx, ty = (i / ntx) % nty, d = i / (ntx * nty);
    if (outside_box(boxes, d, (tx0 + tx) * TS, (ty0 + ty) * TS)) return;
    last_step[((size_t)d * tiles_y + ty0 + ty) * tiles_x + tx0 + tx] = step;
}

int check_window(int32_t D, int32_t T, int32_t Hs, int32_t Ws, int32_t y0, int32_t x0, int32_t wh, int32_t ww) {
    VL3D_REQUIRE(D > 0 && D <= 65535 && T > 0 && Hs > 0 && Ws > 0, "adam window: bad dims");
    VL3D_REQUIRE(y0 >= 0 && x0 >= 0 && wh > 0 && ww > 0 && y0 + wh <= Hs && x0 + ww <= Ws, "adam window: window outside the plane");
    VL3D_REQUIRE(y0 % TS == 0 && x0 % TS == 0 && ((y0 + wh) % TS == 0 || y0 + wh == Hs) && ((x0 + ww) % TS == 0 || x0 + ww == Ws),
                 "adam window: the window must be aligned to the bookkeeping tiles (or end at the plane border)");
    return VL3D_OK;
}

}  // namespace

// Bound on the deferral: one wave per (plane, bookkeeping tile of TS x TS texels); a tile that has missed at least `min_depth` steps is
// brought up to `upto` (replayed, written back, marked), every other wave returns after one 4-byte read.  Run after every step, it keeps
// the replay a returning crop window pays (twice: catch-up for the render, then again in front of the real update) below min_depth
// steps per texel, for stack_bytes * 6 / min_depth of extra traffic per step.  On the reference's schedule (72 crops per epoch at the
// last pyramid level, examples/stage2_schedule.py) windows came back after 30-220 steps: catch-up + step 6.0 ms per iteration
// against 3.0 ms with six cycling crops.
__global__ __launch_bounds__(64) void adam_flush_older_k(int T, int Hs, int Ws, float4 *__restrict__ p, float4 *__restrict__ m,
                                                         float4 *__restrict__ v, int *__restrict__ last_step, int tiles_y, int tiles_x,
                                                         const float2 *__restrict__ hist, int upto, int min_depth, float beta1, float beta2,
                                                         float eps, Quads q, Layout lay) {
    const int tile = blockIdx.x, d = blockIdx.y, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    int *ls = last_step + ((size_t)d * tiles_y + ty) * tiles_x + tx;
    const int from = *ls;
    if (upto - from < min_depth) return;                     // uniform
    const int x = tx * TS + (threadIdx.x % TS), y = ty * TS + (threadIdx.x / TS);
    if (x < Ws && y < Hs) {
        const int cls = texel_class(q, d, x, y, Hs, Ws);
        const int nt = cls == 0 ? 0 : (cls == 2 ? 1 : T);    // culled: no parameter; static: the one copy in frame 0
        size_t o = 0, frame = 0;
        if (nt) texel_slot(lay, d, y, x, T, Hs, Ws, tiles_y, tiles_x, o, frame);
        // four frames per trip, like the window catch-up: twelve independent loads in flight and four independent replay chains (the same operations
        // per value); one frame per trip left every lane of the flush with a memory round trip and one dependent sqrt / rcp chain per frame
        constexpr int NF = 4;
        int t = 0;
        for (; t + NF <= nt; t += NF, o += NF * frame) {
            float4 pp[NF], mm[NF], vv[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) { pp[f] = p[o + f * frame]; mm[f] = m[o + f * frame]; vv[f] = v[o + f * frame]; }
            for (int s = from + 1; s <= upto; ++s) {
                const float2 h = hist[s];
#pragma unroll
                for (int f = 0; f < NF; ++f) adam_upd4(pp[f], make_float4(0.f, 0.f, 0.f, 0.f), mm[f], vv[f], h.x, beta1, beta2, eps, h.y);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) { p[o + f * frame] = pp[f]; m[o + f * frame] = mm[f]; v[o + f * frame] = vv[f]; }
        }
        for (; t < nt; ++t, o += frame) {
            float4 pp = p[o], mm = m[o], vv = v[o];
            replay(pp, mm, vv, hist, from, upto, beta1, beta2, eps);
            p[o] = pp; m[o] = mm; v[o] = vv;
        }
    }
    if (threadIdx.x == 0) *ls = upto;                         // this wave is the tile's only reader and writer in this launch
}

// chosen frames of a packed model as a dense (D,n,Hs,Ws,4) stack (evaluation renders, MPV.py:439 `atlas_dyn[ts]`): one thread per texel
// and plane; blocks without storage read (0, 0, 0, culled_alpha), static blocks their one copy in every frame
__global__ __launch_bounds__(256) void packed_unpack_k(int T, int Hs, int Ws, int tiles_y, int tiles_x, Layout lay, const float4 *__restrict__ pool,
                                                       int n, const int *__restrict__ frames, float culled_alpha, float4 *__restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), d = blockIdx.z;
    if (x >= Ws || y >= Hs) return;
    const int e = lay.blocks[((size_t)d * tiles_y + y / TS) * tiles_x + x / TS];
    const size_t frame = (size_t)Hs * Ws;
    float4 *o = out + (size_t)d * n * frame + (size_t)y * Ws + x;
    if (e < 0) {
        for (int i = 0; i < n; ++i, o += frame) *o = make_float4(0.f, 0.f, 0.f, culled_alpha);
        return;
    }
    const size_t base = (size_t)(e >> 1) * (TS * TS) + (size_t)((y % TS) * TS + (x % TS)), fs = (e & 1) ? (size_t)(TS * TS) : 0;
    for (int i = 0; i < n; ++i, o += frame) *o = pool[base + (size_t)frames[i] * fs];
}

extern "C" int vl3d_packed_unpack_frames(int32_t D, int32_t T, int32_t Hs, int32_t Ws, const int32_t *blocks, const float *pool, int32_t n,
                                         const int32_t *frames, float culled_alpha, float *out, vl3d_stream_t stream) {
    VL3D_REQUIRE(D > 0 && D <= 65535 && T > 0 && Hs > 0 && Ws > 0 && n > 0 && blocks && pool && frames && out, "vl3d_packed_unpack_frames: bad arguments");
    const int tiles_y = (Hs + TS - 1) / TS, tiles_x = (Ws + TS - 1) / TS;
    hipLaunchKernelGGL(packed_unpack_k, dim3((Ws + 63) / 64, (Hs + 3) / 4, D), dim3(256), 0, (hipStream_t)stream, T, Hs, Ws, tiles_y, tiles_x,
                       Layout{blocks}, reinterpret_cast<const float4 *>(pool), n, frames, culled_alpha, reinterpret_cast<float4 *>(out));
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int32_t vl3d_adam_window_tile(void) { return TS; }

extern "C" int vl3d_adam_flush_older(int32_t D, int32_t T, int32_t Hs, int32_t Ws, float *param, float *exp_avg, float *exp_avg_sq,
                                     int32_t *last_step, const float *hist, int32_t upto, int32_t min_depth, float beta1, float beta2, float eps,
                                     const uint8_t *quad_keep, const uint8_t *quad_dyn, int32_t QH, int32_t QW, const int32_t *blocks,
                                     vl3d_stream_t stream) {
    VL3D_REQUIRE(D > 0 && D <= 65535 && T > 0 && Hs > 0 && Ws > 0, "vl3d_adam_flush_older: bad dims");
    VL3D_REQUIRE(!blocks || quad_keep, "vl3d_adam_flush_older: the packed layout belongs to a tile-culled model (quad maps)");
    VL3D_REQUIRE(param && exp_avg && exp_avg_sq && last_step && hist && upto >= 0 && min_depth >= 1, "vl3d_adam_flush_older: null pointer / bad step");
    VL3D_REQUIRE(!quad_keep || quad_grid_ok(QH, QW, Hs, Ws), "vl3d_adam_flush_older: bad quad grid");
    static_assert(TS * TS == 64, "one wave per bookkeeping tile");
    const int tiles_y = (Hs + TS - 1) / TS, tiles_x = (Ws + TS - 1) / TS;
    hipLaunchKernelGGL(adam_flush_older_k, dim3(tiles_y * tiles_x, D), dim3(64), 0, (hipStream_t)stream, T, Hs, Ws, reinterpret_cast<float4 *>(param),
                       reinterpret_cast<float4 *>(exp_avg), reinterpret_cast<float4 *>(exp_avg_sq), last_step, tiles_y, tiles_x,
                       reinterpret_cast<const float2 *>(hist), upto, min_depth, beta1, beta2, eps, make_quads(quad_keep, quad_dyn, QH, QW, Hs, Ws), Layout{blocks});
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_adam_window_catchup_boxes(int32_t D, int32_t T, int32_t Hs, int32_t Ws, int32_t y0, int32_t x0, int32_t wh, int32_t ww,
                                              float *param, float *exp_avg, float *exp_avg_sq, int32_t *last_step, const float *hist,
                                              int32_t upto, float beta1, float beta2, float eps, float *compact, const uint8_t *quad_keep,
                                              const uint8_t *quad_dyn, int32_t QH, int32_t QW, float culled_alpha, int32_t mirror_static,
                                              const int32_t *plane_boxes, const int32_t *blocks, vl3d_stream_t stream) {
    int rc = check_window(D, T, Hs, Ws, y0, x0, wh, ww);
    VL3D_REQUIRE(!blocks || quad_keep, "vl3d_adam_window_catchup: the packed layout belongs to a tile-culled model (quad maps)");
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(param && exp_avg && exp_avg_sq && last_step && hist && upto >= 0, "vl3d_adam_window_catchup: null pointer / negative step");
    VL3D_REQUIRE(!quad_keep || quad_grid_ok(QH, QW, Hs, Ws), "vl3d_adam_window_catchup: bad quad grid");
    const BoxTable boxes = make_boxes(plane_boxes, D);
    const int tiles_y = (Hs + TS - 1) / TS, tiles_x = (Ws + TS - 1) / TS;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(compact ? adam_window_catchup_k<false> : adam_window_catchup_k<true>, dim3((ww + 63) / 64, (wh + 3) / 4, D), dim3(256), 0, s, T, Hs, Ws, Win{y0, x0, wh, ww},
                       reinterpret_cast<float4 *>(param), reinterpret_cast<float4 *>(exp_avg), reinterpret_cast<float4 *>(exp_avg_sq), last_step,
                       tiles_y, tiles_x, reinterpret_cast<const float2 *>(hist), upto, beta1, beta2, eps, reinterpret_cast<float4 *>(compact),
                       make_quads(quad_keep, quad_dyn, QH, QW, Hs, Ws), culled_alpha, mirror_static, compact ? 0 : 1,
                       boxes, Layout{blocks});
    if (!compact) {      // a flush writes the replayed state back and marks the tiles; a catch-up for a render only fills the compact copy
        const int nty = (y0 + wh + TS - 1) / TS - y0 / TS, ntx = (x0 + ww + TS - 1) / TS - x0 / TS;
        hipLaunchKernelGGL(mark_tiles_k, dim3((D * nty * ntx + 255) / 256), dim3(256), 0, s, last_step, tiles_y, tiles_x, y0 / TS, x0 / TS, nty, ntx, D, upto, boxes);
    }
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_adam_window_catchup(int32_t D, int32_t T, int32_t Hs, int32_t Ws, int32_t y0, int32_t x0, int32_t wh, int32_t ww,
                                        float *param, float *exp_avg, float *exp_avg_sq, int32_t *last_step, const float *hist,
                                        int32_t upto, float beta1, float beta2, float eps, float *compact, const uint8_t *quad_keep,
                                        const uint8_t *quad_dyn, int32_t QH, int32_t QW, float culled_alpha, int32_t mirror_static,
                                        vl3d_stream_t stream) {
    return vl3d_adam_window_catchup_boxes(D, T, Hs, Ws, y0, x0, wh, ww, param, exp_avg, exp_avg_sq, last_step, hist, upto, beta1, beta2, eps,
                                          compact, quad_keep, quad_dyn, QH, QW, culled_alpha, mirror_static, nullptr, nullptr, stream);
}

static int window_step_impl(int32_t D, int32_t T, int32_t Hs, int32_t Ws, int32_t y0, int32_t x0, int32_t wh, int32_t ww,
                            float *param, const float *grad_compact, float *exp_avg, float *exp_avg_sq, int32_t *last_step,
                            const float *hist, float lr, float beta1, float beta2, float eps, int64_t step,
                            const uint8_t *quad_keep, const uint8_t *quad_dyn, int32_t QH, int32_t QW, int32_t static_tied,
                            const int32_t *plane_boxes, const int32_t *blocks, const int *dyn_stepped, void *boxes_dev, hipStream_t s) {
    int rc = check_window(D, T, Hs, Ws, y0, x0, wh, ww);
    VL3D_REQUIRE(!blocks || quad_keep, "vl3d_adam_window_step: the packed layout belongs to a tile-culled model (quad maps)");
    if (rc != VL3D_OK) return rc;
    VL3D_REQUIRE(param && grad_compact && exp_avg && exp_avg_sq && last_step && hist && step >= 1, "vl3d_adam_window_step: null pointer / bad step");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const int tiles_y = (Hs + TS - 1) / TS, tiles_x = (Ws + TS - 1) / TS;
    const BoxTable boxes = make_boxes(plane_boxes, D);
    if (boxes_dev && boxes.n) hipLaunchKernelGGL(write_boxes_k, dim3(1), dim3(MAX_BOX_PLANES), 0, s, boxes, reinterpret_cast<int4 *>(boxes_dev));
    hipLaunchKernelGGL(adam_window_step_k, dim3((ww + 63) / 64, (wh + 3) / 4, D), dim3(256), 0, s, T, Hs, Ws, Win{y0, x0, wh, ww},
                       reinterpret_cast<float4 *>(param), reinterpret_cast<const float4 *>(grad_compact), reinterpret_cast<float4 *>(exp_avg),
                       reinterpret_cast<float4 *>(exp_avg_sq), (float)((double)lr / bc1), beta1, beta2, eps, (float)sqrt(bc2),
                       make_quads(quad_keep, quad_dyn, QH, QW, Hs, Ws), static_tied, last_step, tiles_y, tiles_x,
                       reinterpret_cast<const float2 *>(hist), (int)step, boxes, Layout{blocks}, dyn_stepped);
    const int nty = (y0 + wh + TS - 1) / TS - y0 / TS, ntx = (x0 + ww + TS - 1) / TS - x0 / TS;
    hipLaunchKernelGGL(mark_tiles_k, dim3((D * nty * ntx + 255) / 256), dim3(256), 0, s, last_step, tiles_y, tiles_x, y0 / TS, x0 / TS, nty, ntx, D, (int)step, boxes,
                       reinterpret_cast<float2 *>(const_cast<float *>(hist)) + step, make_float2((float)((double)lr / bc1), (float)sqrt(bc2)));
    VL3D_CHECK_LAUNCH();
    return VL3D_OK;
}

extern "C" int vl3d_adam_window_step_boxes(int32_t D, int32_t T, int32_t Hs, int32_t Ws, int32_t y0, int32_t x0, int32_t wh, int32_t ww,
                                           float *param, const float *grad_compact, float *exp_avg, float *exp_avg_sq, int32_t *last_step,
                                           const float *hist, float lr, float beta1, float beta2, float eps, int64_t step,
                                           const uint8_t *quad_keep, const uint8_t *quad_dyn, int32_t QH, int32_t QW, int32_t static_tied,
                                           const int32_t *plane_boxes, const int32_t *blocks, vl3d_stream_t stream) {
    return window_step_impl(D, T, Hs, Ws, y0, x0, wh, ww, param, grad_compact, exp_avg, exp_avg_sq, last_step, hist, lr, beta1, beta2, eps, step,
                            quad_keep, quad_dyn, QH, QW, static_tied, plane_boxes, blocks, nullptr, nullptr, (hipStream_t)stream);
}

// (vl3d_adam.h) the tail of vl3d_render_bwd_adam.  Called BEFORE the render kernels with grad_compact == NULL: only the checks and the box
// table onto the device (boxes_dev); called after them: the step of what the backward left + the tile marks.
int vl3d_adam_window_step_tail(int32_t D, int32_t T, int32_t Hs, int32_t Ws, int32_t y0, int32_t x0, int32_t wh, int32_t ww, float *param,
                               const float *grad_compact, float *exp_avg, float *exp_avg_sq, int32_t *last_step, const float *hist, float lr,
                               float beta1, float beta2, float eps, int64_t step, const uint8_t *quad_keep, const uint8_t *quad_dyn, int32_t QH,
                               int32_t QW, const int32_t *plane_boxes, const int32_t *blocks, const int *plan_ok, void *boxes_dev,
                               hipStream_t stream) {
    if (!grad_compact) {
        int rc = check_window(D, T, Hs, Ws, y0, x0, wh, ww);
        if (rc != VL3D_OK) return rc;
        VL3D_REQUIRE(!plane_boxes || (boxes_dev && D <= MAX_BOX_PLANES), "vl3d_render_bwd_adam: per-plane boxes need boxes_scratch and at most 128 planes");
        VL3D_REQUIRE(!quad_keep || quad_grid_ok(QH, QW, Hs, Ws), "vl3d_render_bwd_adam: bad quad grid");
        if (plane_boxes) {
            const BoxTable boxes = make_boxes(plane_boxes, D);
            for (int d = 0; d < D; ++d) {
                const int4 b = boxes.b[d];
                if (b.y <= b.x || b.w <= b.z) continue;      // an empty box: nothing of this plane is stepped
                VL3D_REQUIRE(b.x % TS == 0 && b.z % TS == 0 && b.x >= y0 && b.z >= x0 && b.y <= y0 + wh && b.w <= x0 + ww &&
                                 (b.y % TS == 0 || b.y == Hs) && (b.w % TS == 0 || b.w == Ws),
                             "vl3d_render_bwd_adam: plane boxes must be aligned to the bookkeeping tiles and lie inside the window");
            }
            hipLaunchKernelGGL(write_boxes_k, dim3(1), dim3(MAX_BOX_PLANES), 0, stream, boxes, reinterpret_cast<int4 *>(boxes_dev));
            VL3D_CHECK_LAUNCH();
        }
        return VL3D_OK;
    }
    return window_step_impl(D, T, Hs, Ws, y0, x0, wh, ww, param, grad_compact, exp_avg, exp_avg_sq, last_step, hist, lr, beta1, beta2, eps, step,
                            quad_keep, quad_dyn, QH, QW, 0, plane_boxes, blocks, plan_ok, nullptr, stream);
}

extern "C" int vl3d_adam_window_step(int32_t D, int32_t T, int32_t Hs, int32_t Ws, int32_t y0, int32_t x0, int32_t wh, int32_t ww,
                                     float *param, const float *grad_compact, float *exp_avg, float *exp_avg_sq, int32_t *last_step,
                                     const float *hist, float lr, float beta1, float beta2, float eps, int64_t step,
                                     const uint8_t *quad_keep, const uint8_t *quad_dyn, int32_t QH, int32_t QW, int32_t static_tied,
                                     vl3d_stream_t stream) {
    return vl3d_adam_window_step_boxes(D, T, Hs, Ws, y0, x0, wh, ww, param, grad_compact, exp_avg, exp_avg_sq, last_step, hist, lr, beta1, beta2,
                                       eps, step, quad_keep, quad_dyn, QH, QW, static_tied, nullptr, nullptr, stream);
}

// the per-step scalars of the table, computed exactly like vl3d_adam_window_step / vl3d_adam_step_tiles compute theirs
extern "C" void vl3d_adam_step_scalars(float lr, float beta1, float beta2, int64_t step, float *lr_bc1, float *bc2s) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    *lr_bc1 = (float)((double)lr / bc1);
    *bc2s = (float)sqrt(bc2);
}
