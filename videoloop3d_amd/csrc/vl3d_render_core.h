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
// Forward (render_fwd2_k): one thread per output pixel, planes walked front-to-back in registers, taps fetched
// straight through the vector L1 (texture-cache style; the 4x tap redundancy between neighbouring pixels is
// absorbed by L1, vertical reuse by the tile height + XCD-aware tile order).
// Backward: a single front-to-back sweep that uses the saved forward outputs,
//   sum_{j>k} w_j q_j = (G.C + gA.A) - sum_{j<=k} w_j q_j   (SURVEY §9.3),
// with the stack gradient produced either by the LDS-staged owner-computes gather kernel (render_bwd_tile_k, no
// atomics) or, for geometry outside its preconditions, by the universal global-atomics kernel (render_bwd_k).
//
// This header holds every render kernel and its launch template; it is compiled once per sampling / compositing convention
// (vl3d_render_conv.inc, included by the vl3d_render_c*.hip stubs) so that the conventions build in parallel.
#pragma once
#include "vl3d_common.h"

// per-plane records of the homography table: 9 floats (the 3x3 matrix) by default; the VL3D_COORD_AFFINE_PLANES convention
// (vl3d_render_c5_mpv_planes.hip) carries 16 per plane -- matrix, then the hard-cut coverage box (x0, x1, y0, y1) in texel
// coordinates, 3 pad -- so that every plane can have its own affine texel transform (folded into its matrix by the host) and its own
// quad extent: the reference's atlas-cell layout (MPV.py:75-81: plane p samples at xm*pitch - (p % grid_w)/grid_w, include/vl3d.h).
#include "vl3d_adam.h"

#ifndef VL3D_HS
#define VL3D_HS 9
#define VL3D_HN 9
#endif

namespace vl3d_render_detail {

// occupancy targets of the two frame-pair kernels (waves per SIMD hipcc must leave room for): measurement builds override them
// (profiles/build_variant.sh NAME -DVL3D_PAIR_MIN_WAVES=...; round 5 session 4 A/B'd 3 / 5 / 6 against 4: profiles/r05d_ab_min_waves.txt)
#ifndef VL3D_PAIR_MIN_WAVES
#define VL3D_PAIR_MIN_WAVES 4
#endif
#ifndef VL3D_FWD2X_MIN_WAVES
#define VL3D_FWD2X_MIN_WAVES 4
#endif

// measurement builds only (-DVL3D_REG_ABLATE=bits, WRONG results; profiles/r06_reg_isa.sh prices the regulariser terms by instruction count):
// forward 1 = no sign encoding / sign-word stores, 2 = no neighbour differences (no LDS reads, no sums), 4 = no LDS exchange (store + barrier);
// backward 8 = no sign decode (reg_grad32), 16 = no sign-word loads
#ifdef VL3D_REG_ABLATE
#define VL3D_REGAB(bit) ((VL3D_REG_ABLATE & (bit)) != 0)
#else
#define VL3D_REGAB(bit) false
#endif
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
    float *asum;          // forward out (optional): per pixel (sum_k a_k, sum_k a_k^2) for the sparsity regulariser (MPV.py:511-515)
    const float *g_asum;  // backward in (optional): per pixel dL/d(sum a), dL/d(sum a^2)
    const float *g_reg;  // device float[4]: dL/d(sum|dx rgb|), dL/d(sum|dy rgb|), dL/d(sum|dx a|), dL/d(sum|dy a|) or NULL
    double *reg_sums;    // device double[4] (forward of the layer-space smoothness regularisers)
    int tiles_x, tiles_y; // tile grid of the owner-computes backward
    int fwd_variant;     // forward kernel selector (see launch<>)
    int ablate;          // measurement-only switches (bit0: skip LDS scatter, bit1: skip flush stores, bit2: skip tap loads)
    float q_inv_cw, q_inv_ch;   // tile culling: quads per texel along x / y = QW/(Ws-1), QH/(Hs-1) (float division done once, on the host)
    float q_x0, q_y0;           // ... with the stack being the texel window [q_y0, q_y0+Hs) x [q_x0, q_x0+Ws) of a q_Hs x q_Ws plane the
    int q_Hs, q_Ws;             // quad grid is laid over (desc->cull_*; the whole plane by default)
    int gather9;         // 1: never take the 2x2 gather (variant 4; the 3x3 gather is the definition the 2x2 one must equal bit for bit)
    int owner4;          // 1: a single frame (T = 1) builds its owner table four texels per thread (bwd_owner_table4_k; variant 3 keeps the one-texel pass).  (In the padding behind gather9.)
    const float *plan;   // device scratch written by bwd_plan_k: [0] feasible flag, [16 + 12*d ..] inverse texel homographies,
                         // then (bwd_windows_k) one int4 texel window per (tile, plane)
    const unsigned short *owner;   // device scratch written by bwd_owner_table_k: per (plane, texel) a 6-bit code of the owner
                                   // tile (tile_y & 7, tile_x & 7) << 10 | the owner pixel's index in its tile's region
    // tile culling (optional): quad_keep [D][QH][QW] bytes, 1 = the quad (cell of the plane's vertex grid) may be visible.
    // cull_masks (forward): per 64x8-pixel workgroup two 64-bit words, bit d = plane d can contribute to the workgroup.
    int g_f16;               // the stack is fp16 (8-byte texels) and so is its gradient
    int Tstride;             // frames between two planes of the stack allocation (= T but for vl3d_render_fwd_frames: a run of frames of a longer clip);
                             // read by the plain forward kernels only.  (HERE, in the padding behind g_f16: as a ninth int behind col0 it shifted every
                             // later kernel argument by 4 bytes and the cfg3 frame-pair backward measured 13.4-14.2 against 12.3-12.5 ms, profiles/r05d_ab_lib.txt)
    const unsigned char *quad_keep;
    int QH, QW;
    const unsigned long long *cull_masks;
    // dispatch options (plain arguments: the ABI is re-entrant from any number of host threads / streams)
    int tile_rows;           // backward: 0 = no owner-computes path for this call (atomics kernel only); 16 = tile kernel; 17 = 16 rows,
                             // frame-pair kernels allowed
    int reg_fwd;             // forward dispatch: 1 = the regulariser-sums kernel instead of the render, 2 = render AND sums in one pass
    // hit-slot layer order of the smoothness regularisers (see "Layer regularisers in hit-slot order" below): the caller's reg_state
    // buffer, written by the forward with regularisers and read by the backward
    unsigned char *reg_flags;          // [H][W]      bit 0/1/2/3: the pair with the right / lower / left / upper neighbour is IRREGULAR
    unsigned long long *reg_masks;     // [H][W][2]   bit d: plane d covers the pixel (frame independent)
    // sign words: one uint16 per (plane, frame, pixel), stored in GROUPS OF FOUR PLANES as one uint64 -- [ceil(D/4)][T][H][W][4] -- so that
    // a forward thread stores 8 bytes every fourth plane and a backward thread loads 8 bytes every fourth plane (2-byte stores of 126-byte
    // row segments measured +1.6 ms on the 7.4 ms forward at cfg3: every cache line was written in parts)
    unsigned short *reg_signs;         // signs of (this pixel's layer value - right neighbour's) | (... - lower neighbour's) << 8
    unsigned short *reg_patch;         // the same towards the left | upper neighbour, irregular pairs only
    // stage 1's learned loop mask as a fifth composited channel (MPI.py:115-117, 568-583; vl3d_render_fwd_mask / _bwd_mask):
    // label = sum_k w_k sigmoid(sample(mask_k)) with the colour composite's weights w_k = a_k T_k, which get no gradient from it
    const float *mask;       // (D,T,Hs,Ws) logits, one float per texel of the stack
    float *label;            // (T,H,W)
    const float *g_label;    // (T,H,W)
    float *g_mask;           // (D,T,Hs,Ws), overwritten
    unsigned uv_seed;            // desc->uv_noise_seed: the jitter field of add_uv_noise (0: off); forward kernels and the atomics backward
    int grad_culled_unwritten;   // desc->grad_flags bit 0: texels owned on a plane the workgroup skips (all of them culled) are not zero-filled
    // the optimiser step fused into the owner store (vl3d_render_bwd_adam; ad.p == NULL otherwise): `stack` is the optimiser's compact window copy
    vl3d_adam_epilogue ad;
    // TILE-EXACT layout of a tile-culled model (include/vl3d.h "Tile-exact layout": a NEGATIVE quad grid at the ABI): every quad owns a tile of
    // q_th x q_tw texels with its OWN border row / column, as the reference's sparsified atlases store them (MPI.py:380-418); the plane is
    // QH * q_th x QW * q_tw texels and a sample's texel coordinate is its shared-border (lattice) coordinate plus its quad index.  0: quads share
    // their border texels.  (At the END of the struct: no earlier kernel argument moves -- see Tstride above.)
    int q_th, q_tw;
};

// one entry point per compiled convention (coord_mode, border_mode, act_order): vl3d_render_c*.hip
int conv_utils_zeros_pre(bool bwd, const vl3d_render_desc *d, const RenderArgs &a, hipStream_t s);
int conv_utils_zeros_post(bool bwd, const vl3d_render_desc *d, const RenderArgs &a, hipStream_t s);
int conv_utils_hardcut_pre(bool bwd, const vl3d_render_desc *d, const RenderArgs &a, hipStream_t s);
int conv_affine_hardcut_post_sig(bool bwd, const vl3d_render_desc *d, const RenderArgs &a, hipStream_t s);
int conv_affine_hardcut_post_other(bool bwd, const vl3d_render_desc *d, const RenderArgs &a, hipStream_t s);
int conv_affine_planes_hardcut_post(bool bwd, const vl3d_render_desc *d, const RenderArgs &a, hipStream_t s);

}  // namespace vl3d_render_detail

namespace {
using vl3d_render_detail::RenderArgs;

// does the texel-space box [tnx,txx] x [tny,txy] (grown by 2 texels) touch a kept quad of plane d?  Quads are the QH x QW cells of
// the plane's vertex grid, (Ws-1)/QW x (Hs-1)/QH texels each.  Conservative by construction: a workgroup skips a plane only
// when every tap of every one of its pixels is a texel no kept quad can read, i.e. a culled texel whose alpha is exactly 0.
__device__ __forceinline__ bool box_touches_kept_quad(const RenderArgs &a, int d, float tnx, float txx, float tny, float txy) {
    tnx += a.q_x0; txx += a.q_x0; tny += a.q_y0; txy += a.q_y0;       // window-local -> plane texel coordinates
    if (!(txx >= -2.0f && tnx <= (float)a.q_Ws + 1.0f && txy >= -2.0f && tny <= (float)a.q_Hs + 1.0f)) return !(tnx == tnx && tny == tny);   // outside the plane (NaN: keep)
    // (tile-exact layout: the box arrives in LATTICE coordinates -- texel_coord of the corners -- and a quad spans q_tw - 1 of them)
    const float cw = a.q_tw ? (float)(a.q_tw - 1) : (float)max(a.q_Ws - 1, 1) / (float)a.QW, ch = a.q_th ? (float)(a.q_th - 1) : (float)max(a.q_Hs - 1, 1) / (float)a.QH;
    const int qx0 = max(0, (int)floorf((tnx - 2.0f) / cw)), qx1 = min(a.QW - 1, (int)floorf((txx + 2.0f) / cw));
    const int qy0 = max(0, (int)floorf((tny - 2.0f) / ch)), qy1 = min(a.QH - 1, (int)floorf((txy + 2.0f) / ch));
    const unsigned char *k = a.quad_keep + (size_t)d * a.QH * a.QW;
    for (int qy = qy0; qy <= qy1; ++qy)
        for (int qx = qx0; qx <= qx1; ++qx)
            if (k[qy * a.QW + qx]) return true;
    return false;
}

// Uniform, read-only tables (homographies, plan records) are read through the constant address space: the loads become
// s_load_* into SGPRs.  Through a generic pointer hipcc cannot prove them invariant across the gradient stores of the
// loop and emits per-lane global_load + s_waitcnt vmcnt(0) -- a full memory latency in front of every plane's taps.
typedef const __attribute__((address_space(4))) float *cfloat_p;
typedef const __attribute__((address_space(4))) int *cint_p;
template <int N>
__device__ __forceinline__ void load_uniform(const float *p, float (&out)[N]) {
    const cfloat_p c = (cfloat_p)p;
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = c[i];
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Taps2 {
    unsigned off;      // byte offset of tap (x0,y0) inside the frame for 16-byte texels; the taps are off + {0, dx, dy, dx+dy}
    f4 w;              // bilinear weights, 0 for taps outside the plane (a vector, not an array: stays in registers)
    float cov;         // 1 if the plane covers this pixel else 0
    float tx, ty;      // texel coordinates of the sample
};
// uniform byte steps between the 4 taps of a sample (0 along an axis of size 1: its second tap has weight 0)
struct TapStep { unsigned dx, dy; };
template <bool F16>
__device__ __forceinline__ TapStep make_tap_step(int Hs, int Ws) {
    constexpr unsigned TEXB = F16 ? 8 : 16;
    return TapStep{Ws > 1 ? TEXB : 0u, Hs > 1 ? (unsigned)Ws * TEXB : 0u};
}

__device__ __forceinline__ float fast_rcp(float z) {
    float r = __builtin_amdgcn_rcpf(z);
    float e = fmaf(-z, r, 1.0f);
    return fmaf(r, e, r);
}

// (x, y) / z from one hardware reciprocal and one residual correction per quotient: q = x*r, q += (x - q*z)*r.  The 1-ulp
// error of v_rcp_f32 only enters through the correction term, so the result agrees with the IEEE quotients the reference /
// oracle compute to within rounding of the last bit, at a quarter of the instruction count of the IEEE expansion.
__device__ __forceinline__ f2 fast_div2(f2 xy, float z) {
    // explicit FMAs: every kernel that inlines this gets the same instruction sequence, hence bit-identical coordinates
    const float rz = __builtin_amdgcn_rcpf(z);
    const f2 rz2 = f2{rz, rz}, z2 = f2{z, z};
    const f2 q = xy * rz2;
    const f2 e = __builtin_elementwise_fma(-q, z2, xy);
    return __builtin_elementwise_fma(e, rz2, q);
}

// bilinear tent weight max(0, 1 - |d|) in ONE full-rate VALU instruction (|.| and clamp are free VOP3 modifiers; the C form
// compiles to and/sub/max, and v_max_f32 alone issues at half the rate of v_sub_f32 -- profiles/microbench/valu_rates)
__device__ __forceinline__ float tent_weight(float d) {
    float w;
    asm("v_sub_f32_e64 %0, 1.0, |%1| clamp" : "=v"(w) : "v"(d));
    return w;
}

// Tile culling at sample level: a sample that falls into a culled quad of its plane is not covered (the reference's mesh has no
// face there, MPI.py:288-442 / MPV.py:389-449) -- whatever the texels hold.  keep == nullptr: no culling.
struct QuadCull {
    const unsigned char *keep;   // [QH][QW] of THIS plane
    int QH, QW;
    float inv_cw, inv_ch;        // quads per texel along x / y: QW/(Ws-1), QH/(Hs-1)
    float x0, y0;                // texel origin of the stack window inside the plane (0 for a whole plane)
    int own;                     // tile-exact layout: the texel coordinate is the lattice coordinate + the quad index (RenderArgs::q_th)
};
__device__ __forceinline__ QuadCull plane_cull(const RenderArgs &a, int d) {
    if (!a.quad_keep) return QuadCull{nullptr, 0, 0, 0.f, 0.f, 0.f, 0.f, 0};
    return QuadCull{a.quad_keep + (size_t)d * a.QH * a.QW, a.QH, a.QW, a.q_inv_cw, a.q_inv_ch, a.q_x0, a.q_y0, a.q_th};     // the two quotients come from the host: uniform, in SGPRs
}

// integer form of the taps: base tap (x0,y0) with x0 <= Ws-2, y0 <= Hs-2 (so the 2x2 block is inside the plane) + weights
struct TapsI {
    int x0, y0;
    f4 w;              // bilinear weights, 0 for taps outside the plane (a vector, not an array: stays in registers)
    float cov;         // 1 if the plane covers this pixel else 0
    float tx, ty;      // texel coordinates of the sample
};

// Sample position -> base tap + weights.  The base tap is floor(t) clamped to [0, S-2], so all four taps are valid addresses
// at constant steps from ONE offset, and each weight is the tent max(0, 1-|t - tap|): for the sample's two neighbours that is
// (1-f, f); for a tap that is not a neighbour (sample within one texel outside the plane, or t == S-1) it is 0, which is
// exactly grid_sample's zeros padding (utils_mpi.py:159-176) -- no per-tap validity selects, no per-tap address clamps.
// add_uv_noise (MPV.py:420-423, MPI.py:519-522): the jitter of one sample, uniform in [-0.5, 0.5) texels per axis, a counter hash of
// (seed, plane, frame pixel) -- restated by oracle/mpi_oracle.uv_jitter_field.  `d` < 0 or seed == 0: no jitter.
struct UvNoise { unsigned seed; int d; };
__device__ __forceinline__ f2 uv_jitter(UvNoise nz, float px, float py) {
    const unsigned ix = (unsigned)(int)floorf(px), iy = (unsigned)(int)floorf(py);      // the frame pixel (pixel centres are 0 or 0.5)
    unsigned hh = nz.seed ^ (ix * 0x9E3779B1u) ^ (iy * 0x85EBCA77u) ^ ((unsigned)nz.d * 0xC2B2AE3Du);
    hh ^= hh >> 15; hh *= 0x2C1B3C6Du; hh ^= hh >> 12; hh *= 0x297A2D39u; hh ^= hh >> 15;
    return f2{((float)(hh & 0xffffu) + 0.5f) * (1.0f / 65536.0f) - 0.5f, ((float)(hh >> 16) + 0.5f) * (1.0f / 65536.0f) - 0.5f};
}

template <int COORD, int BORDER>
__device__ __forceinline__ TapsI make_taps_i(const float *__restrict__ h, float px, float py, int Hs, int Ws,
                                             float sx, float sy, float ox, float oy, QuadCull qc = QuadCull{nullptr, 0, 0, 0.f, 0.f, 0.f, 0.f, 0},
                                             UvNoise nz = UvNoise{0u, 0}) {
    TapsI t;
    const f2 px2 = f2{px, px}, py2 = f2{py, py};
    const f2 XY = __builtin_elementwise_fma(f2{h[0], h[3]}, px2, __builtin_elementwise_fma(f2{h[1], h[4]}, py2, f2{h[2], h[5]}));
    const float Z = fmaf(h[6], px, fmaf(h[7], py, h[8]));
    const f2 pxy = fast_div2(XY, Z);
    float tx0 = texel_coord<COORD>(pxy.x, (float)Ws / 2.0f, (float)(Ws - 1), sx, ox);
    float ty0 = texel_coord<COORD>(pxy.y, (float)Hs / 2.0f, (float)(Hs - 1), sy, oy);
    // tile culling: the quad the sample falls into (cells of the plane's vertex grid, in plane coordinates: window offset added back)
    // (the utils_mpi convention -- no tile-exact layout there -- finds its quad at the END of this function, where it always did: computed up
    // here, hipcc contracts the coordinate's last multiply differently and the culled kernels lose their bit-equality with the plain ones)
    int qx = 0, qy = 0;
    if (COORD != VL3D_COORD_UTILS_MPI && qc.keep) {      // uniform branch
        qx = min(max((int)floorf((tx0 + qc.x0) * qc.inv_cw), 0), qc.QW - 1); qy = min(max((int)floorf((ty0 + qc.y0) * qc.inv_ch), 0), qc.QH - 1);
        // TILE-EXACT layout (MPI.py:380-418: every kept quad is a tile with its own border samples; MPV.py:394-427: a face's UVs span exactly
        // its tile): what the homography + (sx, ox) gave is the LATTICE coordinate L (quads sharing their borders, qx = floor(L / (tw - 1)));
        // the sample's texel coordinate inside the plane of QW x tw texels is L + qx -- local position L - qx (tw - 1) in [0, tw - 1] of tile
        // qx, whose first texel is qx tw.  A piecewise translation: tap weights, hard cut and window offsets work on it unchanged, and the
        // second tap of a sample never leaves its tile with a non-zero weight (local position tw - 1 exactly: weight 0).
        if constexpr (COORD != VL3D_COORD_UTILS_MPI) {      // (the planar conventions only: check_cull refuses it elsewhere)
            if (qc.own) { tx0 += (float)qx; ty0 += (float)qy; }
        }
    }
    float tx = tx0, ty = ty0;
    // (the affine conventions only -- MPV.py / MPI.py's planar path is where the reference has the flag, and the entry points refuse it
    // elsewhere: in the utils_mpi instantiations a possibly-jittered tx would also stop hipcc from contracting the last multiply of the
    // coordinate into `tx - x0`, one ulp away from the kernels that do not carry the branch)
    if constexpr (COORD != VL3D_COORD_UTILS_MPI) {
        if (nz.seed) {      // uniform branch: the taps move, the coverage tests below keep the unjittered position (the rasteriser's, in the reference)
            const f2 j = uv_jitter(nz, px, py);
            tx += j.x; ty += j.y;
        }
    }
    t.tx = tx; t.ty = ty;
    const float x0f = __builtin_amdgcn_fmed3f(floorf(tx), 0.0f, (float)max(Ws - 2, 0));
    const float y0f = __builtin_amdgcn_fmed3f(floorf(ty), 0.0f, (float)max(Hs - 2, 0));
    t.x0 = (int)x0f; t.y0 = (int)y0f;
    const float dx = tx - x0f, dy = ty - y0f;
    // an axis of size 1 has no second tap: its tent is pushed to 0
    const float wx0 = tent_weight(dx), wx1 = tent_weight(dx - (Ws > 1 ? 1.0f : 3e38f));
    const float wy0 = tent_weight(dy), wy1 = tent_weight(dy - (Hs > 1 ? 1.0f : 3e38f));
    t.w = f4{wx0, wx1, wx0, wx1} * f4{wy0, wy0, wy1, wy1};
    if constexpr (BORDER == VL3D_BORDER_HARDCUT && VL3D_HN == 13) {   // per-plane quad extent (atlas cells: the box is not the texel range)
        const bool cov = (tx0 >= h[9]) && (tx0 <= h[10]) && (ty0 >= h[11]) && (ty0 <= h[12]);
        t.cov = cov ? 1.0f : 0.0f;
    } else if constexpr (BORDER == VL3D_BORDER_HARDCUT) {   // MPV.py:374-453: the plane quad ends at the outermost texel centres
        const bool cov = (tx0 >= 0.0f) && (tx0 <= (float)(Ws - 1)) && (ty0 >= 0.0f) && (ty0 <= (float)(Hs - 1));
        t.cov = cov ? 1.0f : 0.0f;
    } else {                                         // zeros padding: covered while any tap is inside, i.e. any weight > 0
        t.cov = ((t.w[0] + t.w[1]) + (t.w[2] + t.w[3]) > 0.0f) ? 1.0f : 0.0f;
    }
    if (qc.keep) {      // uniform branch
        if constexpr (COORD == VL3D_COORD_UTILS_MPI) {
            qx = min(max((int)floorf((tx0 + qc.x0) * qc.inv_cw), 0), qc.QW - 1); qy = min(max((int)floorf((ty0 + qc.y0) * qc.inv_ch), 0), qc.QH - 1);
        }
        if (!qc.keep[qy * qc.QW + qx]) t.cov = 0.0f;
    }
    return t;
}

template <int COORD, int BORDER>
__device__ __forceinline__ Taps2 make_taps2(const float *__restrict__ h, float px, float py, int Hs, int Ws,
                                            float sx, float sy, float ox, float oy, QuadCull qc = QuadCull{nullptr, 0, 0, 0.f, 0.f, 0.f, 0.f, 0},
                                            UvNoise nz = UvNoise{0u, 0}) {
    const TapsI ti = make_taps_i<COORD, BORDER>(h, px, py, Hs, Ws, sx, sy, ox, oy, qc, nz);
    Taps2 t;
    t.w = ti.w;
    t.cov = ti.cov; t.tx = ti.tx; t.ty = ti.ty;
    t.off = (__umul24((unsigned)ti.y0, (unsigned)Ws) + (unsigned)ti.x0) << 4;   // y0 < 2^24 (check_desc), full-rate multiply
    return t;
}

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// one texel = 16 bytes (fp32 stack) or 8 bytes (fp16 stack, cfg5 of BASELINE.json; arithmetic stays fp32).
// `plane` is uniform and off16 a zero-extended 32-bit lane offset: global_load ... v_off, s[base:base+1]
template <bool F16>
__device__ __forceinline__ f4 load_texel(const char *__restrict__ plane, unsigned off16) {
    if constexpr (F16) return __builtin_convertvector(*reinterpret_cast<const h4 *>(plane + (size_t)(off16 >> 1)), f4);
    else return *reinterpret_cast<const f4 *>(plane + (size_t)off16);
}

// The stack gradient has the dtype of the stack (8-byte fp16 texels for cfg5: 96 GB of stack + 96 GB of gradient per GPU fit
// in 288 GB; an fp32 gradient would not).  `tex16` is the byte offset the texel would have with 16-byte texels.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <bool F16>
__device__ __forceinline__ void store_grad_texel(char *gplane, unsigned tex16, f4 v) {
    if constexpr (F16) __builtin_nontemporal_store(__builtin_convertvector(v, h4), reinterpret_cast<h4 *>(gplane + (size_t)(tex16 >> 1)));
    else __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(gplane + (size_t)tex16));
}
template <bool F16>
__device__ __forceinline__ void atomic_add_grad_texel(char *gplane, size_t tex16, f4 c) {
    if constexpr (F16) {     // packed-half atomics (global_atomic_pk_add_f16); fallback path only
        auto *g = (__attribute__((address_space(1))) h2 *)(gplane + (tex16 >> 1));
        __builtin_amdgcn_global_atomic_fadd_v2f16(g, h2{(_Float16)c.x, (_Float16)c.y});
        __builtin_amdgcn_global_atomic_fadd_v2f16(g + 1, h2{(_Float16)c.z, (_Float16)c.w});
    } else {
        float *g = reinterpret_cast<float *>(gplane + tex16);
        atomicAdd(g + 0, c.x); atomicAdd(g + 1, c.y); atomicAdd(g + 2, c.z); atomicAdd(g + 3, c.w);
    }
}

// the four taps: one lane offset, four uniform bases (plane, +dx, +dy, +dx+dy) -- no per-tap address arithmetic
template <bool F16>
__device__ __forceinline__ void load_taps2(const char *__restrict__ plane, const Taps2 &t, TapStep st, f4 v[4]) {
    v[0] = load_texel<F16>(plane, t.off);
    v[1] = load_texel<F16>(plane + st.dx, t.off);
    v[2] = load_texel<F16>(plane + st.dy, t.off);
    v[3] = load_texel<F16>(plane + st.dy + st.dx, t.off);
}

// fp16 stacks in sample-then-activate order keep their taps packed (2 VGPRs per tap instead of 4) and convert inside the
// blend's FMAs: v_fma_mix_f32 reads an f16 half as one source and computes in fp32, i.e. exactly cvt + fma, in one plain-rate
// instruction.  The 16 v_cvt_f32_f16 per pixel and plane this removes were ~20 % of the fp16 forward's VALU issue time.
typedef unsigned u2 __attribute__((ext_vector_type(2)));
template <bool F16, int ORDER> struct TapVal { typedef f4 type; };
template <> struct TapVal<true, VL3D_ACT_POST> { typedef u2 type; };

template <bool F16>
__device__ __forceinline__ void load_taps2(const char *__restrict__ plane, const Taps2 &t, TapStep st, u2 v[4]) {
    static_assert(F16, "packed taps are fp16 texels");
    const size_t o = (size_t)(t.off >> 1);
    // texels x0 and x0+1 of a row are 16 contiguous bytes: ONE 16-byte load per row (8-byte aligned) instead of two 8-byte ones.
    // The fp16 forward was bound by the rate of tap-load instructions (~20 clk per wave-load per CU): 4.10 -> 3.12 ms.
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef u4 u4_a8 __attribute__((aligned(8)));
    const u4 r0 = *reinterpret_cast<const u4_a8 *>(plane + o);
    const u4 r1 = *reinterpret_cast<const u4_a8 *>(plane + st.dy + o);
    v[0] = u2{r0.x, r0.y}; v[1] = u2{r0.z, r0.w};
    v[2] = u2{r1.x, r1.y}; v[3] = u2{r1.z, r1.w};
}

template <int HI>
__device__ __forceinline__ float fma_mix(unsigned hpair, float w, float acc) {      // (float)half[HI] * w + acc
    float r;
    if constexpr (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(w), "v"(acc));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(w), "v"(acc));
    return r;
}

// The composite's scalar chains with their fused multiply-adds spelt out: under -ffp-contract=fast hipcc picks a different
// grouping in every kernel it inlines them into, and the backward kernels (tile / frame pairs / atomics) are compared bit for bit.
__device__ __forceinline__ float dot3p(float a0, float b0, float a1, float b1, float a2, float b2, float c) {
    return fmaf(a0, b0, fmaf(a1, b1, fmaf(a2, b2, c)));
}

template <int RACT, int AACT>
__device__ __forceinline__ f4 act4(f4 s) {
    return f4{act_fwd<RACT>(s.x), act_fwd<RACT>(s.y), act_fwd<RACT>(s.z), act_fwd<AACT>(s.w)};
}

// bilinear blend + activation (vector types so the blend compiles to packed FMAs)
template <int ORDER, int RACT, int AACT>
__device__ __forceinline__ f4 shade2(const Taps2 &t, const f4 v[4], f4 *pre_out = nullptr) {
    f4 s;
    if constexpr (ORDER == VL3D_ACT_POST) {
        s = v[0] * t.w[0] + (v[1] * t.w[1] + (v[2] * t.w[2] + v[3] * t.w[3]));
        if (pre_out) *pre_out = s;
        s = act4<RACT, AACT>(s);
    } else {
        s = act4<RACT, AACT>(v[0]) * t.w[0] + (act4<RACT, AACT>(v[1]) * t.w[1] + (act4<RACT, AACT>(v[2]) * t.w[2] + act4<RACT, AACT>(v[3]) * t.w[3]));
        if (pre_out) *pre_out = s;
    }
    s.w *= t.cov;    // uncovered: a = 0 (and c irrelevant) -> the plane drops out of the composite
    return s;
}

// the same blend (same association: v3 w3, then fma v2, v1, v0 -- bit-identical to the f4 overload on the converted values)
template <int ORDER, int RACT, int AACT>
__device__ __forceinline__ f4 shade2(const Taps2 &t, const u2 v[4], f4 *pre_out = nullptr) {
    static_assert(ORDER == VL3D_ACT_POST, "packed taps: sample-then-activate only");
    f4 s;
    s.x = fma_mix<0>(v[0].x, t.w[0], fma_mix<0>(v[1].x, t.w[1], fma_mix<0>(v[2].x, t.w[2], fma_mix<0>(v[3].x, t.w[3], 0.0f))));
    s.y = fma_mix<1>(v[0].x, t.w[0], fma_mix<1>(v[1].x, t.w[1], fma_mix<1>(v[2].x, t.w[2], fma_mix<1>(v[3].x, t.w[3], 0.0f))));
    s.z = fma_mix<0>(v[0].y, t.w[0], fma_mix<0>(v[1].y, t.w[1], fma_mix<0>(v[2].y, t.w[2], fma_mix<0>(v[3].y, t.w[3], 0.0f))));
    s.w = fma_mix<1>(v[0].y, t.w[0], fma_mix<1>(v[1].y, t.w[1], fma_mix<1>(v[2].y, t.w[2], fma_mix<1>(v[3].y, t.w[3], 0.0f))));
    if (pre_out) *pre_out = s;
    s = act4<RACT, AACT>(s);
    s.w *= t.cov;
    return s;
}

// the loop-mask texture's four taps (4-byte texels at the stack's tap positions) and their blend, associated like shade2's
__device__ __forceinline__ void load_mask_taps(const float *__restrict__ mplane, unsigned off16, TapStep st, float m[4]) {
    const char *b = reinterpret_cast<const char *>(mplane) + (size_t)(off16 >> 2);
    m[0] = *reinterpret_cast<const float *>(b);
    m[1] = *reinterpret_cast<const float *>(b + (st.dx >> 2));
    m[2] = *reinterpret_cast<const float *>(b + (st.dy >> 2));
    m[3] = *reinterpret_cast<const float *>(b + (st.dy >> 2) + (st.dx >> 2));
}
__device__ __forceinline__ float mask_blend(const float m[4], f4 w) { return m[0] * w[0] + (m[1] * w[1] + (m[2] * w[2] + m[3] * w[3])); }

// =====================================================================================================
// Layer regularisers in hit-slot order (MPV.py:386-392, 441-449, 517-531; MPI.py:553-566, 608-622; utils.py:51-69).
// The reference's layer tensor mpi[T,h,w,K,4] is indexed by HIT SLOT: slot k of a pixel is the k-th nearest face the rasteriser hit
// there (masked_scatter over the z-sorted pix_to_face), i.e. its k-th nearest COVERED plane, and rgb_smooth / a_smooth difference
// neighbouring pixels per slot.  Where two neighbours are covered by the same planes that is the difference per plane; where they
// are not (a plane's edge inside the view; every quad border of a tile-culled model) every deeper slot pairs two DIFFERENT planes.
//   * reg_masks_k / reg_flags_k (frame independent, once per forward): per pixel the 128-bit coverage mask over the planes and four
//     flags "the pair with my right / lower / left / upper neighbour is irregular" (the two masks differ).
//   * the forward kernels with regularisers form the differences of REGULAR pairs plane by plane (both pixels covered by the same
//     planes: slot k is the same plane on both sides) exactly as before, and store their signs: 2 bits per channel, two's complement
//     (+1 = 01, -1 = 11, 0 = 00 -- equal values have gradient 0 like torch's abs), right pair in bits 0-7, lower pair in bits 8-15
//     of one uint16 per (plane, frame, pixel).  Irregular pairs are left out there (code 0) ...
//   * ... and taken by reg_slot_fwd_k (below), whose threads walk their own covered planes slot by slot: it adds |difference| of the
//     irregular pairs (including the slots only one of the two pixels has) to the sums, adds their signs to the owner's sign word of
//     ITS plane and writes the negated signs to the neighbour's patch word of the NEIGHBOUR's plane.  For tile-culled models, where
//     most pairs are irregular, the same kernel forms ALL pairs and the plane-by-plane kernel is not run at all.
//   * the backward kernels never see layer values of neighbours any more: d sum / d layer value = gx (s_right + s_left) + gy (s_down
//     + s_up), the four signs decoded from this pixel's word, the left / upper neighbours' words of the same plane (regular pairs:
//     sign(left - me) is the left pixel's right-pair code) or this pixel's patch word (irregular pairs).  No LDS exchange, no second
//     barrier, no 2-pixel halo: the kernels with regularisers are the plain ones plus 2-4 two-byte loads and a decode per plane.
__device__ __forceinline__ int xcd_remap(int b, int nb);      // (defined with the forward kernels below)
// Sign fields: 2 bits per channel holding sign + 1 (0: negative, 1: zero -- equal values have gradient 0 like torch's abs --, 2: positive).
// Encoding four of them: the float's bit pattern clamped to [-1, 1] as an INTEGER is its sign (one v_med3_i32: positive floats are
// positive integers, +0 is 0 and x - x is +0); sum c_k 4^k + 0b01010101 = sum (c_k + 1) 4^k has no borrows, so a byte costs four
// clamps and three shift-adds (the bias is added by the caller, once per word).
__device__ __forceinline__ int reg_sign(float v) {      // (min(max()) compiles to two compares and two selects)
    int r;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(__float_as_int(v)));
    return r;
}
__device__ __forceinline__ int reg_signs4(f4 diff) {      // sum c_k 4^k, WITHOUT the bias
    return reg_sign(diff.x) + (reg_sign(diff.y) << 2) + (reg_sign(diff.z) << 4) + (reg_sign(diff.w) << 6);
}
constexpr int REG_BIAS = 0x55;            // + 1 in each of a byte's four fields
constexpr unsigned REG_ZERO = 0x5555u;    // a word whose eight signs are all 0
// gradient of the four smoothness sums w.r.t. this pixel's activated layer value on one plane and frame.
// w_own / w_left / w_up: sign words of this pixel, its left and its upper neighbour (REG_ZERO where that neighbour does not exist);
// w_patch: this pixel's patch word (read when a left / upper pair is irregular); fl: the pixel's flags.
// A neighbour's word holds sign(neighbour - me) (its right / lower pair), the patch word sign(me - neighbour).
__device__ __forceinline__ f4 reg_grad(unsigned w_own, unsigned w_left, unsigned w_up, unsigned w_patch, unsigned fl, f4 gx, f4 gy) {
    const unsigned l = (fl & 4u) ? w_patch : w_left, u = (fl & 8u) ? (w_patch >> 8) : (w_up >> 8);
    // u_R - 1 -+ (u_L - 1): with the neighbour's word the biases cancel, with the patch word they add up to -2
    const int ls = (fl & 4u) ? 1 : -1, us = (fl & 8u) ? 1 : -1, lo = (fl & 4u) ? -2 : 0, uo = (fl & 8u) ? -2 : 0;
    f4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ix = (int)__builtin_amdgcn_ubfe(w_own, 2 * c, 2) + ls * (int)__builtin_amdgcn_ubfe(l, 2 * c, 2) + lo;
        const int iy = (int)__builtin_amdgcn_ubfe(w_own, 8 + 2 * c, 2) + us * (int)__builtin_amdgcn_ubfe(u, 2 * c, 2) + uo;
        r[c] = fmaf(gy[c], (float)iy, gx[c] * (float)ix);
    }
    return r;
}

// The same for the hot backward kernels: the words are the 32-bit halves of a group (two planes each) and `bo` (uniform: 0 or 16) the bit
// offset of this plane's fields, so a field is ONE v_bfe_u32 with a scalar offset -- no 64-bit shifts, no masks.  Regular pairs (all of a
// dense frame but plane edges) take the 10-instruction-per-channel path; a pixel with an irregular left / upper pair branches (rare, divergent).
// w_left / w_up must hold REG_ZERO's pattern (0x55555555) where that neighbour does not exist.
__device__ __forceinline__ f4 reg_grad32(unsigned w_own, unsigned w_left, unsigned w_up, unsigned w_patch, unsigned bo, unsigned fl, f4 gx, f4 gy) {
    f4 r;
    if (__builtin_expect((fl & 12u) == 0u, 1)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ix = (int)__builtin_amdgcn_ubfe(w_own, bo + 2 * c, 2) - (int)__builtin_amdgcn_ubfe(w_left, bo + 2 * c, 2);
            const int iy = (int)__builtin_amdgcn_ubfe(w_own, bo + 8 + 2 * c, 2) - (int)__builtin_amdgcn_ubfe(w_up, bo + 8 + 2 * c, 2);
            r[c] = fmaf(gy[c], (float)iy, gx[c] * (float)ix);
        }
    } else {
        r = reg_grad((w_own >> bo) & 0xffffu, (w_left >> bo) & 0xffffu, (w_up >> bo) & 0xffffu, (w_patch >> bo) & 0xffffu, fl, gx, gy);
    }
    return r;
}

template <int COORD, int BORDER>
__global__ __launch_bounds__(256) void reg_masks_k(RenderArgs a) {
    // (first kernel of every forward that accumulates the four regulariser sums, launch_reg_prepass: it clears them -- the entry points issued a
    // 32-byte hipMemsetAsync for that, one launch per iteration)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4 && a.reg_sums) a.reg_sums[threadIdx.x] = 0.0;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const float px = (float)(a.col0 + x) + a.pc, py = (float)(a.row0 + y) + a.pc;
    unsigned long long m0 = 0ull, m1 = 0ull;
    for (int d = 0; d < a.D; ++d) {
        float h[VL3D_HN];
        load_uniform(a.homos + VL3D_HS * d, h);
        const TapsI t = make_taps_i<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, plane_cull(a, d));
        if (t.cov > 0.0f) { if (d < 64) m0 |= 1ull << d; else m1 |= 1ull << (d - 64); }
    }
    a.reg_masks[2 * ((size_t)y * a.W + x)] = m0;
    a.reg_masks[2 * ((size_t)y * a.W + x) + 1] = m1;
}

__global__ __launch_bounds__(256) void reg_flags_k(RenderArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const size_t p = (size_t)y * a.W + x;
    const unsigned long long *M = a.reg_masks;
    const unsigned long long m0 = M[2 * p], m1 = M[2 * p + 1];
    auto differs = [&](size_t q) { return M[2 * q] != m0 || M[2 * q + 1] != m1; };
    unsigned f = 0;
    if (x + 1 < a.W && differs(p + 1)) f |= 1u;
    if (y + 1 < a.H && differs(p + a.W)) f |= 2u;
    if (x >= 1 && differs(p - 1)) f |= 4u;
    if (y >= 1 && differs(p - a.W)) f |= 8u;
    a.reg_flags[p] = (unsigned char)f;
}

// lowest set bit of a 128-bit mask, removed from it; -1 when empty
// (selects, not branches: with `if (m0) {...} else if (m1) {...}` hipcc turned the two words into a dynamically indexed private array --
// 24 bytes of scratch and a scratch store per slot)
__device__ __forceinline__ int reg_pop_plane(unsigned long long &m0, unsigned long long &m1) {
    const bool lo = m0 != 0ull;
    const unsigned long long m = lo ? m0 : m1;
    if (!m) return -1;
    const int d = __builtin_ctzll(m) + (lo ? 0 : 64);
    const unsigned long long c = m & (m - 1);
    m0 = lo ? c : m0;
    m1 = lo ? m1 : c;
    return d;
}

// activated layer value of pixel (px, py) on plane d in frame t, sampled in place (plane and pixel vary per lane)
// (hs: the homography table -- the slot kernel stages it in LDS: the plane varies per lane, and a global load per slot in front of the taps
// was a full memory latency on the slot loop's critical path)
template <int COORD, int BORDER, int ORDER, int RACT, int AACT, bool F16>
__device__ __forceinline__ f4 reg_sample(const RenderArgs &a, const float *hs, float px, float py, int d, int t) {
    float h[VL3D_HN];
#pragma unroll
    for (int i = 0; i < VL3D_HN; ++i) h[i] = hs[VL3D_HS * d + i];
    const Taps2 tp = make_taps2<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, plane_cull(a, d), UvNoise{a.uv_seed, d});
    const char *plane = reinterpret_cast<const char *>(a.stack) + ((size_t)d * a.T + t) * ((size_t)a.Hs * a.Ws * (F16 ? 8 : 16));
    typename TapVal<F16, ORDER>::type tv[4];
    load_taps2<F16>(plane, tp, make_tap_step<F16>(a.Hs, a.Ws), tv);
    return shade2<ORDER, RACT, AACT>(tp, tv) * tp.cov;
}

// Slot-synchronous regulariser forward.  A workgroup is a 64 x 8-pixel region whose last column / row are halo (63 x 7 pixels own their
// right / lower pairs); every thread walks ITS OWN covered planes near -> far, so iteration k handles slot k of every pixel: the
// layer value (and its plane) goes through a double-buffered LDS tile, one barrier per slot, and the pair owner reads its
// neighbour's slot-k value next to its own -- exactly the reference's per-slot difference, with no resampling.  Per slot it
//   - adds |difference| to the four sums,
//   - puts the pair's signs into its own sign word of ITS plane (right pair: bits 0-7, lower pair: bits 8-15),
//   - and, for an irregular pair, writes the negated signs into the NEIGHBOUR's patch word of the NEIGHBOUR's plane (one byte:
//     left field for the right neighbour, upper field for the lower one), including the slots only the neighbour has.
// PATCH = false: every pair of the frame (the whole regulariser forward: tile-culled models, where most pairs are irregular, and the
//   two-pass forward of dense ones); sign words are written whole, four planes per 8-byte store.
// PATCH = true: only the irregular pairs, after render_fwd_reg_k took the regular ones plane by plane: regions without an irregular
//   pair return at once; the signs are ADDED to the words that kernel wrote (it left sign 0 in those fields).
// RENDER (with PATCH = false): the composite as well -- a thread visits its covered planes nearest first, which is the order of the
//   over-composite, so rgb / alpha / the alpha sums of its pixel fall out of the same samples (the planes it skips have alpha exactly 0
//   for it: the bits of render_fwd2_k's culled composite).  The whole forward of a tile-culled model with regularisers in one pass
//   (vl3d_render_fwd_reg_culled) instead of the culled render + this kernel over the same taps.
template <int COORD, int BORDER, int ORDER, int RACT, int AACT, bool F16, bool PATCH, bool RENDER = false>
__global__ __launch_bounds__(512) void reg_slot_fwd_k(RenderArgs a, int tiles_x, int tiles_y) {
    static_assert(!(PATCH && RENDER), "the composite rides the kernel that visits every covered plane of every pixel");
    constexpr int FW = 64, FH = 8, NT = FW * FH;
    __shared__ float4 s_v[2][NT];
    __shared__ int s_d[2][NT];
    __shared__ float red[4][FH];
    __shared__ int s_kmax;
    __shared__ float s_h[128 * VL3D_HS];      // the planes' homography records (D <= 128: the coverage masks' limit)
    for (int i = threadIdx.x; i < a.D * VL3D_HS; i += NT) s_h[i] = a.homos[i];      // (visible behind the barriers below)
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_x = b % tiles_x, rest = b / tiles_x;
    const int tile_y = rest % tiles_y, t = rest / tiles_y;
    const int tid = threadIdx.x, lane = tid & 63, row = tid >> 6;
    const int x = tile_x * (FW - 1) + lane, y = tile_y * (FH - 1) + row;
    const bool inimg = (x < a.W) && (y < a.H);
    const bool owner = inimg && lane < FW - 1 && row < FH - 1;
    const size_t p = (size_t)min(y, a.H - 1) * a.W + min(x, a.W - 1);
    const unsigned fl = owner ? a.reg_flags[p] : 0u;
    if (PATCH && !__syncthreads_or((fl & 3u) != 0u)) return;
    // pairs this thread forms: its right / lower one, when it exists (and, in patch mode, is irregular)
    const bool irr_r = (fl & 1u) != 0u, irr_d = (fl & 2u) != 0u;
    const bool own_r = owner && x + 1 < a.W && (!PATCH || irr_r), own_d = owner && y + 1 < a.H && (!PATCH || irr_d);
    unsigned long long m0 = inimg ? a.reg_masks[2 * p] : 0ull, m1 = inimg ? a.reg_masks[2 * p + 1] : 0ull;
    if (tid == 0) s_kmax = 0;
    __syncthreads();
    atomicMax(&s_kmax, __builtin_popcountll(m0) + __builtin_popcountll(m1));
    __syncthreads();
    const int kmax = s_kmax;
    const float px = (float)(a.col0 + min(x, a.W - 1)) + a.pc, py = (float)(a.row0 + min(y, a.H - 1)) + a.pc;
    const size_t plane_px = (size_t)a.T * a.H * a.W, fpix = ((size_t)t * a.H + min(y, a.H - 1)) * a.W + min(x, a.W - 1);
    unsigned long long *const sg64 = reinterpret_cast<unsigned long long *>(a.reg_signs);
    unsigned char *const patch8 = reinterpret_cast<unsigned char *>(a.reg_patch);
    float sxc = 0.f, syc = 0.f, sxa = 0.f, sya = 0.f;
    float Tr = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, A = 0.f, n1 = 0.f, n2 = 0.f;      // RENDER: the composite state of this pixel
    unsigned long long sgw = 0ull;      // (full mode) sign words of the group of four planes being filled
    int sgg = -1;                       // ... and its index
    for (int k = 0; k < kmax; ++k) {
        const int buf = k & 1;
        const int d = reg_pop_plane(m0, m1);      // my plane in slot k (-1: I have no slot k)
        f4 v = f4{0.f, 0.f, 0.f, 0.f};
        if (d >= 0) v = reg_sample<COORD, BORDER, ORDER, RACT, AACT, F16>(a, s_h, px, py, d, t);
        if constexpr (RENDER) {      // (spelt like VL3D_COMPOSITE of render_fwd2_k; an absent slot adds exact zeros)
            const float w = v.w * Tr;
            cr += w * v.x; cg += w * v.y; cb += w * v.z; A += w;
            n1 += v.w; n2 = fmaf(v.w, v.w, n2);
            Tr *= (1.0f - v.w);
        }
        s_v[buf][tid] = make_float4(v.x, v.y, v.z, v.w);
        s_d[buf][tid] = d;
        __syncthreads();
        int code = PATCH ? 0 : (int)REG_ZERO;
        if (own_r) {
            const float4 r = s_v[buf][tid + 1];
            const int dn = s_d[buf][tid + 1];
            if (d >= 0 || dn >= 0) {
                const f4 df = v - f4{r.x, r.y, r.z, r.w};
                sxc += fabsf(df.x) + fabsf(df.y) + fabsf(df.z);
                sxa += fabsf(df.w);
                const int c = reg_signs4(df);
                code += c;
                if (irr_r && dn >= 0)      // the right neighbour's left field on ITS plane: sign(it - me)
                    patch8[((((size_t)(dn >> 2) * plane_px + fpix + 1) << 2) + (dn & 3)) * 2] = (unsigned char)(REG_BIAS - c);
            }
        }
        if (own_d) {
            const float4 r = s_v[buf][tid + FW];
            const int dn = s_d[buf][tid + FW];
            if (d >= 0 || dn >= 0) {
                const f4 df = v - f4{r.x, r.y, r.z, r.w};
                syc += fabsf(df.x) + fabsf(df.y) + fabsf(df.z);
                sya += fabsf(df.w);
                const int c = reg_signs4(df);
                code += c << 8;
                if (irr_d && dn >= 0)      // the lower neighbour's upper field on ITS plane
                    patch8[((((size_t)(dn >> 2) * plane_px + fpix + a.W) << 2) + (dn & 3)) * 2 + 1] = (unsigned char)(REG_BIAS - c);
            }
        }
        if (owner && d >= 0) {
            if constexpr (PATCH) {
                if (code) {
                    unsigned short *w = a.reg_signs + ((((size_t)(d >> 2) * plane_px + fpix) << 2) + (d & 3));
                    *w = (unsigned short)((int)*w + code);
                }
            } else {
                if ((d >> 2) != sgg) {
                    if (sgg >= 0) sg64[(size_t)sgg * plane_px + fpix] = sgw;
                    sgg = d >> 2; sgw = 0ull;
                }
                sgw |= (unsigned long long)(unsigned)code << (16 * (d & 3));
            }
        }
    }
    if (!PATCH && owner && sgg >= 0) sg64[(size_t)sgg * plane_px + fpix] = sgw;
    if constexpr (RENDER) if (owner) {
        a.rgb[fpix * 3 + 0] = cr; a.rgb[fpix * 3 + 1] = cg; a.rgb[fpix * 3 + 2] = cb;
        a.alpha[fpix] = A;
        if (a.asum) { a.asum[fpix * 2 + 0] = n1; a.asum[fpix * 2 + 1] = n2; }
    }
    float v4[4] = {sxc, syc, sxa, sya};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v4[k] += __shfl_down(v4[k], off, 64);
        if (lane == 0) red[k][row] = v4[k];
    }
    __syncthreads();
    if (tid < 4) {
        double sum = 0.0;
        for (int r = 0; r < FH; ++r) sum += (double)red[tid][r];
        if (sum != 0.0) atomicAdd(a.reg_sums + tid, sum);
    }
}

template <int COORD, int BORDER, int ORDER, int RACT, int AACT, bool F16>
void launch_reg_prepass(const RenderArgs &a, hipStream_t s) {
    const dim3 g((a.W + 63) / 64, (a.H + 3) / 4);
    hipLaunchKernelGGL((reg_masks_k<COORD, BORDER>), g, dim3(256), 0, s, a);
    hipLaunchKernelGGL(reg_flags_k, g, dim3(256), 0, s, a);
}
template <int COORD, int BORDER, int ORDER, int RACT, int AACT, bool F16, bool PATCH, bool RENDER = false>
void launch_reg_slots(const RenderArgs &a, hipStream_t s) {
    const int tx = (a.W + 62) / 63, ty = (a.H + 6) / 7;
    hipLaunchKernelGGL((reg_slot_fwd_k<COORD, BORDER, ORDER, RACT, AACT, F16, PATCH, RENDER>), dim3((unsigned)(tx * ty * a.T)), dim3(512), 0, s, a, tx, ty);
}

constexpr int TILE_X = 64, TILE_Y = 4;

template <int COORD, int BORDER, int ORDER, int RACT, int AACT, bool F16, bool MASK = false>
__global__ __launch_bounds__(TILE_X *TILE_Y) void render_bwd_k(RenderArgs a) {
    if (a.plan && reinterpret_cast<const int *>(a.plan)[0]) return;   // the tile path owns this call
    const int x = blockIdx.x * TILE_X + (threadIdx.x & (TILE_X - 1));
    const int y = blockIdx.y * TILE_Y + (threadIdx.x / TILE_X);
    const int t = blockIdx.z;
    if (x >= a.W || y >= a.H) return;
    const float px = (float)(a.col0 + x) + a.pc, py = (float)(a.row0 + y) + a.pc;
    constexpr size_t TEXB = F16 ? 8 : 16;
    const size_t frame = (size_t)a.Hs * a.Ws * 4;
    const size_t plane_stride = (size_t)a.T * frame;
    const char *plane = reinterpret_cast<const char *>(a.stack) + (size_t)t * a.Hs * a.Ws * TEXB;
    char *gplane = reinterpret_cast<char *>(a.g_stack) + (size_t)t * a.Hs * a.Ws * TEXB;     // gradient texels = stack texels
    const size_t pix = ((size_t)t * a.H + y) * a.W + x;
    const float Gr = a.g_rgb[pix * 3 + 0], Gg = a.g_rgb[pix * 3 + 1], Gb = a.g_rgb[pix * 3 + 2];
    const float gA = a.g_alpha ? a.g_alpha[pix] : 0.0f;
    // S = sum_k w_k q_k with q_k = G.c_k + gA  ==  G.C + gA*A from the saved forward outputs
    const float S = dot3p(Gr, a.rgb[pix * 3 + 0], Gg, a.rgb[pix * 3 + 1], Gb, a.rgb[pix * 3 + 2], gA * a.alpha[pix]);
    const float gN1 = a.g_asum ? a.g_asum[pix * 2 + 0] : 0.0f, gN2 = a.g_asum ? 2.0f * a.g_asum[pix * 2 + 1] : 0.0f;
    const unsigned fl = a.g_reg ? a.reg_flags[(size_t)y * a.W + x] : 0u;
    float Tr = 1.0f, P = 0.0f;
    const TapStep st = make_tap_step<F16>(a.Hs, a.Ws), gst = make_tap_step<false>(a.Hs, a.Ws);
    const float gL = MASK ? a.g_label[pix] : 0.0f;
    const size_t mframe = (size_t)a.Hs * a.Ws;
    for (int d = 0; d < a.D; ++d, plane += (size_t)a.T * a.Hs * a.Ws * TEXB, gplane += (size_t)a.T * a.Hs * a.Ws * TEXB) {
        const Taps2 tp = make_taps2<COORD, BORDER>(a.homos + VL3D_HS * d, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, plane_cull(a, d), UvNoise{a.uv_seed, d});
        if (tp.cov == 0.0f) continue;
        typename TapVal<F16, ORDER>::type tv[4];
        f4 pre;
        load_taps2<F16>(plane, tp, st, tv);
        const f4 o = shade2<ORDER, RACT, AACT>(tp, tv, &pre);
        const float q = dot3p(Gr, o.x, Gg, o.y, Gb, o.z, gA);
        const float w = o.w * Tr;
        if constexpr (MASK) {      // d label / d mask texel = g_label w_k sigmoid'(m_k) x tap weight (the weights w_k are detached, MPI.py:577-579)
            float mt[4];
            load_mask_taps(a.mask + ((size_t)d * a.T + t) * mframe, tp.off, gst, mt);
            const float sm = act_fwd<VL3D_ACT_SIGMOID>(mask_blend(mt, tp.w));
            const float gm = gL * w * (sm * (1.0f - sm));
            float *gmp = a.g_mask + ((size_t)d * a.T + t) * mframe + (tp.off >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tp.w[i] != 0.0f) atomicAdd(gmp + ((i & 1) ? (gst.dx >> 4) : 0u) + ((i & 2) ? (gst.dy >> 4) : 0u), gm * tp.w[i]);
        }
        P = fmaf(w, q, P);
        const float om = 1.0f - o.w;
        // dL/da_k = T_k q_k - (sum_{j>k} w_j q_j)/(1-a_k); everything behind a fully opaque plane has zero weight
        const float behind = (om > 1e-12f) ? (S - P) * fast_rcp(om) : 0.0f;
        f4 go = f4{w * Gr, w * Gg, w * Gb, fmaf(Tr, q, -behind) + fmaf(gN2, o.w, gN1)};   // grad wrt activated (c, a)
        Tr *= om;
        if (a.g_reg) {   // smoothness regularisers: the signs the forward stored (hit-slot order)
            const f4 gx = f4{a.g_reg[0], a.g_reg[0], a.g_reg[0], a.g_reg[2]}, gy = f4{a.g_reg[1], a.g_reg[1], a.g_reg[1], a.g_reg[3]};
            const size_t idx = (((((size_t)(d >> 2) * a.T + t) * a.H + y) * a.W + x) << 2) + (d & 3);
            const unsigned w_left = x >= 1 ? a.reg_signs[idx - 4] : REG_ZERO, w_up = y >= 1 ? a.reg_signs[idx - 4 * (size_t)a.W] : REG_ZERO;
            go += reg_grad(a.reg_signs[idx], w_left, w_up, (fl & 12u) ? a.reg_patch[idx] : 0u, fl, gx, gy);
        }
        if constexpr (ORDER == VL3D_ACT_POST)
            go = f4{go.x * act_bwd<RACT>(pre.x, o.x), go.y * act_bwd<RACT>(pre.y, o.y), go.z * act_bwd<RACT>(pre.z, o.z),
                    go.w * act_bwd<AACT>(pre.w, o.w)};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (tp.w[i] != 0.0f) {
                f4 c = go * tp.w[i];
                if constexpr (ORDER == VL3D_ACT_PRE) {
                    const f4 sv = tv[i];
                    c = f4{c.x * act_bwd<RACT>(sv.x, act_fwd<RACT>(sv.x)), c.y * act_bwd<RACT>(sv.y, act_fwd<RACT>(sv.y)),
                           c.z * act_bwd<RACT>(sv.z, act_fwd<RACT>(sv.z)), c.w * act_bwd<AACT>(sv.w, act_fwd<AACT>(sv.w))};
                }
                atomic_add_grad_texel<F16>(gplane, (size_t)tp.off + ((i & 1) ? gst.dx : 0u) + ((i & 2) ? gst.dy : 0u), c);
            }
        }
    }
}

// =====================================================================================================
// Forward, variant 2: same arithmetic as render_fwd_k with a leaner instruction stream
//   - one Newton-refined reciprocal instead of two IEEE divisions for the perspective divide,
//   - branch-free taps: addresses clamped into the plane, invalid taps get weight 0, uncovered planes get a = 0,
//     so the plane loop has uniform control flow and the taps of plane d+1 are issued before plane d is shaded,
//   - 32-bit byte offsets against a scalar plane base (global_load_dwordx4 v, s[base:base+1]),
//   - taller tiles (TY rows: vertical tap reuse inside the workgroup) on a 1-D grid with an XCD-aware
//     bijective remap so that vertically adjacent tiles share one XCD's L2.
// bijective XCD remap (cdna guide T1): workgroup b runs on XCD b % 8; give every XCD a contiguous chunk of tiles
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// forward plan of the tile culling: one thread per (64 x TY pixel workgroup, plane) projects the workgroup's corners and sets
// the plane's bit when the footprint touches a kept quad.  Frame independent, once per call.
template <int COORD>
__global__ __launch_bounds__(256) void cull_fwd_plan_k(RenderArgs a, int TY, int tiles_x, int tiles_y, unsigned long long *masks) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= tiles_x * tiles_y * a.D) return;
    const int d = i % a.D, tile = i / a.D, tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int x0 = tile_x * 64, x1 = min(x0 + 63, a.W - 1), y0 = tile_y * TY, y1 = min(y0 + TY - 1, a.H - 1);
    const float *h = a.homos + VL3D_HS * d;
    float tnx = 1e30f, txx = -1e30f, tny = 1e30f, txy = -1e30f;
    for (int c = 0; c < 4; ++c) {
        const float cx = (float)a.col0 + a.pc + (float)((c & 1) ? x1 : x0), cy = (float)a.row0 + a.pc + (float)((c & 2) ? y1 : y0);
        const float X = h[0] * cx + h[1] * cy + h[2], Y = h[3] * cx + h[4] * cy + h[5], Z = h[6] * cx + h[7] * cy + h[8];
        const float ctx = texel_coord<COORD>(X / Z, (float)a.Ws / 2.0f, (float)(a.Ws - 1), a.sx, a.ox);
        const float cty = texel_coord<COORD>(Y / Z, (float)a.Hs / 2.0f, (float)(a.Hs - 1), a.sy, a.oy);
        tnx = fminf(tnx, ctx); txx = fmaxf(txx, ctx); tny = fminf(tny, cty); txy = fmaxf(txy, cty);
    }
    if (box_touches_kept_quad(a, d, tnx, txx, tny, txy)) atomicOr(masks + (size_t)tile * 2 + (d >> 6), 1ull << (d & 63));
}

template <int COORD, int BORDER, int ORDER, int RACT, int AACT, int TY, bool SWZ, bool F16, bool CULL = false, bool MASK = false>
__global__ __launch_bounds__(64 * TY) void render_fwd2_k(RenderArgs a, int tiles_x, int tiles_y) {
    static_assert(!(MASK && (CULL || F16)), "the loop-mask channel: dense fp32 stage-1 stacks");
    int b = blockIdx.x;
    if constexpr (SWZ) b = xcd_remap(b, gridDim.x);
    const int tile_x = b % tiles_x, rest = b / tiles_x;
    const int tile_y = rest % tiles_y, t = rest / tiles_y;
    const int x = tile_x * 64 + (threadIdx.x & 63);
    const int y = tile_y * TY + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const float px = (float)(a.col0 + x) + a.pc, py = (float)(a.row0 + y) + a.pc;
    const size_t frame_b = (size_t)a.Hs * a.Ws * (F16 ? 8 : 16);
    const size_t plane_stride_b = (size_t)a.Tstride * frame_b;
    const char *plane = reinterpret_cast<const char *>(a.stack) + (size_t)t * frame_b;
    float Tr = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, A = 0.f, n1 = 0.f, n2 = 0.f;
    const TapStep st = make_tap_step<F16>(a.Hs, a.Ws);
    typedef typename TapVal<F16, ORDER>::type tapv_t;
    tapv_t vA[4], vB[4];
    float mA[4] = {0.f, 0.f, 0.f, 0.f}, mB[4] = {0.f, 0.f, 0.f, 0.f}, lab = 0.f;      // MASK: the loop-mask texture's taps, the composited label
    const float *mplane = MASK ? a.mask + (size_t)t * a.Hs * a.Ws : nullptr;
    const size_t mplane_stride = (size_t)a.T * a.Hs * a.Ws;
#define VL3D_COMPOSITE(T_, V_, M_)                                    \
    {                                                                 \
        const f4 o = shade2<ORDER, RACT, AACT>(T_, V_);               \
        const float w = o.w * Tr;                                     \
        cr += w * o.x; cg += w * o.y; cb += w * o.z; A += w;          \
        n1 += o.w; n2 = fmaf(o.w, o.w, n2);                           \
        Tr *= (1.0f - o.w);                                           \
        if constexpr (MASK) lab = fmaf(w, act_fwd<VL3D_ACT_SIGMOID>(mask_blend(M_, T_.w)), lab);      \
    }
    if constexpr (CULL) {
        // tile culling: walk only the planes whose bit is set for this workgroup (two 64-bit words in SGPRs, scalar bit scans);
        // the skipped planes' taps are all culled texels (alpha exactly 0), so the result is bit-identical to walking them
        const unsigned long long *mk = a.cull_masks + (size_t)(tile_y * tiles_x + tile_x) * 2;
        unsigned long long m0 = ((const __attribute__((address_space(4))) unsigned long long *)mk)[0];
        unsigned long long m1 = ((const __attribute__((address_space(4))) unsigned long long *)mk)[1];
        auto next = [&]() {
            int d = -1;
            if (m0) { d = __builtin_ctzll(m0); m0 &= m0 - 1; }
            else if (m1) { d = 64 + __builtin_ctzll(m1); m1 &= m1 - 1; }
            return d;
        };
        auto fetch = [&](int d, Taps2 &t, tapv_t *v) {
            float h[VL3D_HN];
            load_uniform(a.homos + VL3D_HS * d, h);
            t = make_taps2<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, plane_cull(a, d), UvNoise{a.uv_seed, d});
            load_taps2<F16>(plane + (size_t)d * plane_stride_b, t, st, v);
            asm volatile("" ::: "memory");
        };
        int dA = next();
        if (dA >= 0) {
            Taps2 tA, tB;
            fetch(dA, tA, vA);
            for (;;) {
                const int dB = next();
                fetch(dB < 0 ? dA : dB, tB, vB);      // unconditional prefetch (re-reads the current plane past the end)
                VL3D_COMPOSITE(tA, vA, mA)
                if (dB < 0) break;
                const int dC = next();
                fetch(dC < 0 ? dB : dC, tA, vA);
                VL3D_COMPOSITE(tB, vB, mB)
                if (dC < 0) break;
                dA = dC;
            }
        }
    } else {
    // two register sets (A/B) so the taps of plane d+1 are in flight while plane d is shaded, without register copies
    const QuadCull noq = QuadCull{nullptr, 0, 0, 0.f, 0.f, 0.f, 0.f, 0};
    Taps2 tA = make_taps2<COORD, BORDER>(a.homos, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, noq, UvNoise{a.uv_seed, 0}), tB = tA;
    load_taps2<F16>(plane, tA, st, vA);
    if constexpr (MASK) load_mask_taps(mplane, tA.off, st, mA);
    // The prefetch of the next plane is unconditional (past the end it re-reads the last plane): with a branch around the
    // loads hipcc merges the two paths' counters and waits with vmcnt(0), i.e. for the taps it has just issued as well.
    for (int d = 0;; d += 2) {
        {
            const int dn = min(d + 1, a.D - 1);
            float h[VL3D_HN];
            load_uniform(a.homos + VL3D_HS * dn, h);
            tB = make_taps2<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, noq, UvNoise{a.uv_seed, dn});
            load_taps2<F16>(plane + (size_t)dn * plane_stride_b, tB, st, vB);
            if constexpr (MASK) load_mask_taps(mplane + (size_t)dn * mplane_stride, tB.off, st, mB);
            asm volatile("" ::: "memory");   // keep the loads here: hipcc otherwise sinks them below the composite
        }
        VL3D_COMPOSITE(tA, vA, mA)
        if (d + 1 >= a.D) break;
        {
            const int dn = min(d + 2, a.D - 1);
            float h[VL3D_HN];
            load_uniform(a.homos + VL3D_HS * dn, h);
            tA = make_taps2<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, noq, UvNoise{a.uv_seed, dn});
            load_taps2<F16>(plane + (size_t)dn * plane_stride_b, tA, st, vA);
            if constexpr (MASK) load_mask_taps(mplane + (size_t)dn * mplane_stride, tA.off, st, mA);
            asm volatile("" ::: "memory");
        }
        VL3D_COMPOSITE(tB, vB, mB)
        if (d + 2 >= a.D) break;
    }
    }
#undef VL3D_COMPOSITE
    const size_t pix = ((size_t)t * a.H + y) * a.W + x;
    a.rgb[pix * 3 + 0] = cr; a.rgb[pix * 3 + 1] = cg; a.rgb[pix * 3 + 2] = cb;
    a.alpha[pix] = A;
    if (a.asum) { a.asum[pix * 2 + 0] = n1; a.asum[pix * 2 + 1] = n2; }
    if constexpr (MASK) a.label[pix] = lab;
}

// Forward, two frames per thread.  Both render kernels are VALU-issue bound, and half of the forward's instruction
// stream depends on the plane and the pixel only -- homography, perspective divide, base tap, tent weights, coverage, tap offset --
// not on the frame: a thread that composites frames t and t+1 of its pixel side by side pays for it once.  The per-frame
// arithmetic is that of render_fwd2_k instruction for instruction (same results bit for bit); the 8 tap loads per plane and
// thread also replace occupancy as the source of memory parallelism (<= 128 VGPRs, 4 waves per SIMD).
// CULL: tile culling (render_fwd2_k's plan and plane walk): only the planes whose bit is set for this workgroup, two frames each.
template <int COORD, int BORDER, int ORDER, int RACT, int AACT, int TY, bool F16, bool CULL = false>
__global__ __launch_bounds__(64 * TY, VL3D_FWD2X_MIN_WAVES) void render_fwd2x_k(RenderArgs a, int tiles_x, int tiles_y) {      // >= 4 waves per SIMD: <= 128 VGPRs
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_x = b % tiles_x, rest = b / tiles_x;
    const int tile_y = rest % tiles_y, t0 = (rest / tiles_y) * 2;
    const bool has1 = t0 + 1 < a.T;          // odd T: the last pair composites frame t0 twice and stores it once
    const int x = tile_x * 64 + (threadIdx.x & 63);
    const int y = tile_y * TY + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    const float px = (float)(a.col0 + x) + a.pc, py = (float)(a.row0 + y) + a.pc;
    const size_t frame_b = (size_t)a.Hs * a.Ws * (F16 ? 8 : 16);
    const size_t plane_stride_b = (size_t)a.Tstride * frame_b;
    const char *plane0 = reinterpret_cast<const char *>(a.stack) + (size_t)t0 * frame_b;
    const char *plane1 = plane0 + (has1 ? frame_b : 0);
    float Tr0 = 1.0f, cr0 = 0.f, cg0 = 0.f, cb0 = 0.f, A0 = 0.f, n10 = 0.f, n20 = 0.f;
    float Tr1 = 1.0f, cr1 = 0.f, cg1 = 0.f, cb1 = 0.f, A1 = 0.f, n11 = 0.f, n21 = 0.f;
    const TapStep st = make_tap_step<F16>(a.Hs, a.Ws);
    typedef typename TapVal<F16, ORDER>::type tapv_t;
    tapv_t vA0[4], vA1[4], vB0[4], vB1[4];
#define VL3D_COMPOSITE2(T_, V0_, V1_)                                        \
    {                                                                        \
        const f4 o0 = shade2<ORDER, RACT, AACT>(T_, V0_);                    \
        const f4 o1 = shade2<ORDER, RACT, AACT>(T_, V1_);                    \
        const float w0 = o0.w * Tr0, w1 = o1.w * Tr1;                        \
        cr0 += w0 * o0.x; cg0 += w0 * o0.y; cb0 += w0 * o0.z; A0 += w0;      \
        cr1 += w1 * o1.x; cg1 += w1 * o1.y; cb1 += w1 * o1.z; A1 += w1;      \
        n10 += o0.w; n20 = fmaf(o0.w, o0.w, n20);                            \
        n11 += o1.w; n21 = fmaf(o1.w, o1.w, n21);                            \
        Tr0 *= (1.0f - o0.w); Tr1 *= (1.0f - o1.w);                          \
    }
    if constexpr (CULL) {
        // the workgroup's plane list (cull_fwd_plan_k): two 64-bit words in SGPRs, scalar bit scans; a skipped plane's taps are all culled
        // texels (alpha exactly 0), so the result is bit-identical to walking it
        const unsigned long long *mk = a.cull_masks + (size_t)(tile_y * tiles_x + tile_x) * 2;
        unsigned long long m0 = ((const __attribute__((address_space(4))) unsigned long long *)mk)[0];
        unsigned long long m1 = ((const __attribute__((address_space(4))) unsigned long long *)mk)[1];
        auto next = [&]() {
            int d = -1;
            if (m0) { d = __builtin_ctzll(m0); m0 &= m0 - 1; }
            else if (m1) { d = 64 + __builtin_ctzll(m1); m1 &= m1 - 1; }
            return d;
        };
        auto fetch = [&](int d, Taps2 &t, tapv_t *v0, tapv_t *v1) {
            float h[VL3D_HN];
            load_uniform(a.homos + VL3D_HS * d, h);
            t = make_taps2<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, plane_cull(a, d));
            load_taps2<F16>(plane0 + (size_t)d * plane_stride_b, t, st, v0);
            load_taps2<F16>(plane1 + (size_t)d * plane_stride_b, t, st, v1);
            asm volatile("" ::: "memory");
        };
        int dA = next();
        if (dA >= 0) {
            Taps2 tA, tB;
            fetch(dA, tA, vA0, vA1);
            for (;;) {
                const int dB = next();
                fetch(dB < 0 ? dA : dB, tB, vB0, vB1);      // unconditional prefetch (re-reads the current plane past the end)
                VL3D_COMPOSITE2(tA, vA0, vA1)
                if (dB < 0) break;
                const int dC = next();
                fetch(dC < 0 ? dB : dC, tA, vA0, vA1);
                VL3D_COMPOSITE2(tB, vB0, vB1)
                if (dC < 0) break;
                dA = dC;
            }
        }
    } else {
    Taps2 tA = make_taps2<COORD, BORDER>(a.homos, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy), tB = tA;
    load_taps2<F16>(plane0, tA, st, vA0);
    load_taps2<F16>(plane1, tA, st, vA1);
    for (int d = 0;; d += 2) {
        {
            const int dn = min(d + 1, a.D - 1);
            float h[VL3D_HN];
            load_uniform(a.homos + VL3D_HS * dn, h);
            tB = make_taps2<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy);
            load_taps2<F16>(plane0 + (size_t)dn * plane_stride_b, tB, st, vB0);
            load_taps2<F16>(plane1 + (size_t)dn * plane_stride_b, tB, st, vB1);
            asm volatile("" ::: "memory");
        }
        VL3D_COMPOSITE2(tA, vA0, vA1)
        if (d + 1 >= a.D) break;
        {
            const int dn = min(d + 2, a.D - 1);
            float h[VL3D_HN];
            load_uniform(a.homos + VL3D_HS * dn, h);
            tA = make_taps2<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy);
            load_taps2<F16>(plane0 + (size_t)dn * plane_stride_b, tA, st, vA0);
            load_taps2<F16>(plane1 + (size_t)dn * plane_stride_b, tA, st, vA1);
            asm volatile("" ::: "memory");
        }
        VL3D_COMPOSITE2(tB, vB0, vB1)
        if (d + 2 >= a.D) break;
    }
    }
#undef VL3D_COMPOSITE2
    size_t pix = ((size_t)t0 * a.H + y) * a.W + x;
    a.rgb[pix * 3 + 0] = cr0; a.rgb[pix * 3 + 1] = cg0; a.rgb[pix * 3 + 2] = cb0;
    a.alpha[pix] = A0;
    if (a.asum) { a.asum[pix * 2 + 0] = n10; a.asum[pix * 2 + 1] = n20; }
    if (has1) {
        pix += (size_t)a.H * a.W;
        a.rgb[pix * 3 + 0] = cr1; a.rgb[pix * 3 + 1] = cg1; a.rgb[pix * 3 + 2] = cb1;
        a.alpha[pix] = A1;
        if (a.asum) { a.asum[pix * 2 + 0] = n11; a.asum[pix * 2 + 1] = n21; }
    }
}

template <int COORD, int BORDER, int ORDER, int RACT, int AACT, bool F16>
void launch_fwd2x(const RenderArgs &a, hipStream_t s) {
    constexpr int TY = 8;
    const int tiles_x = (a.W + 63) / 64, tiles_y = (a.H + TY - 1) / TY;
    const dim3 grid((unsigned)(tiles_x * tiles_y * ((a.T + 1) / 2))), block(64 * TY);
    if (a.quad_keep && a.cull_masks) {       // tile culling: plan (frame independent; the workgroups are render_fwd2_k's 64 x 8 tiles), then the plane-list kernel
        auto *masks = const_cast<unsigned long long *>(a.cull_masks);
        (void)hipMemsetAsync(masks, 0, (size_t)tiles_x * tiles_y * 16, s);
        const int n = tiles_x * tiles_y * a.D;
        hipLaunchKernelGGL((cull_fwd_plan_k<COORD>), dim3((n + 255) / 256), dim3(256), 0, s, a, TY, tiles_x, tiles_y, masks);
        hipLaunchKernelGGL((render_fwd2x_k<COORD, BORDER, ORDER, RACT, AACT, TY, F16, true>), grid, block, 0, s, a, tiles_x, tiles_y);
        return;
    }
    hipLaunchKernelGGL((render_fwd2x_k<COORD, BORDER, ORDER, RACT, AACT, TY, F16>), grid, block, 0, s, a, tiles_x, tiles_y);
}

template <int COORD, int BORDER, int ORDER, int RACT, int AACT, int TY, bool SWZ, bool F16 = false>
void launch_fwd2(const RenderArgs &a, hipStream_t s) {
    const int tiles_x = (a.W + 63) / 64, tiles_y = (a.H + TY - 1) / TY;
    if (a.quad_keep && a.cull_masks) {       // tile culling: plan (frame independent), then the plane-list kernel
        auto *masks = const_cast<unsigned long long *>(a.cull_masks);
        (void)hipMemsetAsync(masks, 0, (size_t)tiles_x * tiles_y * 16, s);
        const int n = tiles_x * tiles_y * a.D;
        hipLaunchKernelGGL((cull_fwd_plan_k<COORD>), dim3((n + 255) / 256), dim3(256), 0, s, a, TY, tiles_x, tiles_y, masks);
        hipLaunchKernelGGL((render_fwd2_k<COORD, BORDER, ORDER, RACT, AACT, TY, SWZ, F16, true>), dim3((unsigned)(tiles_x * tiles_y * a.T)),
                           dim3(64 * TY), 0, s, a, tiles_x, tiles_y);
        return;
    }
    hipLaunchKernelGGL((render_fwd2_k<COORD, BORDER, ORDER, RACT, AACT, TY, SWZ, F16>), dim3((unsigned)(tiles_x * tiles_y * a.T)),
                       dim3(64 * TY), 0, s, a, tiles_x, tiles_y);
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
// frame are zero-filled by bwd_owner_table_k (run first; it also zeroes a 1-pixel safety band that the
// tile kernel then overwrites).  bwd_plan_k checks the geometric preconditions per call ON DEVICE
// (Z>0 over the frame, magnification < 1.4x, window fits); if they fail, these kernels exit and the
// universal atomics kernel above runs instead -- no host synchronisation either way.
constexpr int RW = 64;        // region width in pixels = one wave
constexpr int PLAN_HDR = 16;  // floats before the per-plane records
constexpr int PLAN_REC = 12;  // per plane: 9 floats inverse texel homography, 2 floats gather radius (x,y), 1 pad
// after the per-plane records (16-byte aligned): one int4 per (tile, plane) = texel window of the tile's owned pixels on
// that plane: X0, Y0, width | height << 16, float bits of 1/width
__host__ __device__ inline int plan_win_off(int D) { return (PLAN_HDR + PLAN_REC * D + 3) & ~3; }

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
        texel_homography<COORD>(a.homos + VL3D_HS * d, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, M);
        const double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
                           M[2] * (M[3] * M[7] - M[4] * M[6]);
        bool ok = (det == det) && fabs(det) > 1e-30;
        double I[9];
        I[0] = (M[4] * M[8] - M[5] * M[7]) / det; I[1] = (M[2] * M[7] - M[1] * M[8]) / det; I[2] = (M[1] * M[5] - M[2] * M[4]) / det;
        I[3] = (M[5] * M[6] - M[3] * M[8]) / det; I[4] = (M[0] * M[8] - M[2] * M[6]) / det; I[5] = (M[2] * M[3] - M[0] * M[5]) / det;
        I[6] = (M[3] * M[7] - M[4] * M[6]) / det; I[7] = (M[1] * M[6] - M[0] * M[7]) / det; I[8] = (M[0] * M[4] - M[1] * M[3]) / det;
        // normalise so the pixel-space w of the frame centre is ~1 (keeps fp32 well scaled)
        for (int i = 0; i < 9; ++i) plan[PLAN_HDR + PLAN_REC * d + i] = (float)I[i];
        double rxm = 0.0, rym = 0.0;
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
                rxm = fmax(rxm, i_r0); rym = fmax(rym, i_r1);
                // keep the owned footprint of a tile a small multiple of the workgroup (pure efficiency guard)
                if (!(fabs(j00) + fabs(j01) < 4.0 && fabs(j10) + fabs(j11) < 4.0)) ok = false;
            }
        // gather radius per axis: a pixel p contributes to texel tau only if |p - H^-1 tau| < |J^-1|_inf-row (2% safety)
        plan[PLAN_HDR + PLAN_REC * d + 9] = (float)fmin(1.02 * rxm + 1e-3, 1.45);
        plan[PLAN_HDR + PLAN_REC * d + 10] = (float)fmin(1.02 * rym + 1e-3, 1.45);
        if (!ok) atomicAnd(&ok_all, 0);
    }
    __syncthreads();
    if (threadIdx.x == 0) { reinterpret_cast<int *>(plan)[0] = ok_all; reinterpret_cast<int *>(plan)[1] = 0; }
    if (threadIdx.x < 4) plan[4 + threadIdx.x] = 0.0f;     // four zero floats: stand-in for g_reg when only the sparsity sums have a gradient
    if (threadIdx.x >= 8 && threadIdx.x < 16) plan[threadIdx.x] = 0.0f;      // (the rest of the header: the caller need not clear the scratch buffer)
    if (threadIdx.x == 2 || threadIdx.x == 3) plan[threadIdx.x] = 0.0f;
}

// Texel window of every (tile, plane): the footprint of the tile's owned pixels, from the image of its four corners
// (convex image of a rectangle, Z>0).  Texels owned by a tile have their owner pixel inside it, i.e. H^-1(tau) within 0.5 px
// of the tile (0.55 here: the margin absorbs the fp32 error of the corner images); beyond a frame border the owner is the
// clamped border pixel, so only texels within the 1.4 px contribution range matter (1.6).  Frame independent: computed
// once per call here instead of by wave 0 of every workgroup for every plane (90 VALU instructions on the barrier path).
template <int COORD>
__global__ __launch_bounds__(256) void bwd_windows_k(RenderArgs a, int iw, int ih, int rh, int tiles_x, int tiles_y, int *win) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= tiles_x * tiles_y * a.D) return;
    const int d = i % a.D, tile = i / a.D, tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int ix0 = tile_x * iw, ix1 = min(ix0 + iw - 1, a.W - 1), iy0 = tile_y * ih, iy1 = min(iy0 + ih - 1, a.H - 1);
    const float el = (ix0 == 0) ? 1.6f : 0.55f, er = (ix1 == a.W - 1) ? 1.6f : 0.55f;
    const float et = (iy0 == 0) ? 1.6f : 0.55f, eb = (iy1 == a.H - 1) ? 1.6f : 0.55f;
    const float *h = a.homos + VL3D_HS * d;
    float mnx = 1e30f, mxx = -1e30f, mny = 1e30f, mxy = -1e30f;
    for (int c = 0; c < 4; ++c) {
        const float cx = (float)a.col0 + a.pc + ((c & 1) ? (float)ix1 + er : (float)ix0 - el);
        const float cy = (float)a.row0 + a.pc + ((c & 2) ? (float)iy1 + eb : (float)iy0 - et);
        const float X = h[0] * cx + h[1] * cy + h[2], Y = h[3] * cx + h[4] * cy + h[5], Z = h[6] * cx + h[7] * cy + h[8];
        const float ctx = texel_coord<COORD>(X / Z, (float)a.Ws / 2.0f, (float)(a.Ws - 1), a.sx, a.ox);
        const float cty = texel_coord<COORD>(Y / Z, (float)a.Hs / 2.0f, (float)(a.Hs - 1), a.sy, a.oy);
        mnx = fminf(mnx, ctx); mxx = fmaxf(mxx, ctx); mny = fminf(mny, cty); mxy = fmaxf(mxy, cty);
    }
    if (a.q_th) {
        // tile-exact layout: the corners' images are LATTICE coordinates (window-local; + q_x0 = in the plane); a texel of quad q sits q
        // columns to the right of its lattice position, and a lattice column on a quad border is TWO texel columns (the last of one tile,
        // the first of the next): the lower end of the box takes the quad strictly below its coordinate, the upper end the quad at or
        // above it (1e-3 quads of slack for the fp32 product: a larger window is only a few more table entries)
        const float qlo_x = fminf(fmaxf(ceilf((mnx - 0.01f + a.q_x0) * a.q_inv_cw - 1e-3f) - 1.0f, 0.0f), (float)(a.QW - 1));
        const float qhi_x = fminf(fmaxf(floorf((mxx + 0.01f + a.q_x0) * a.q_inv_cw + 1e-3f), 0.0f), (float)(a.QW - 1));
        const float qlo_y = fminf(fmaxf(ceilf((mny - 0.01f + a.q_y0) * a.q_inv_ch - 1e-3f) - 1.0f, 0.0f), (float)(a.QH - 1));
        const float qhi_y = fminf(fmaxf(floorf((mxy + 0.01f + a.q_y0) * a.q_inv_ch + 1e-3f), 0.0f), (float)(a.QH - 1));
        mnx += qlo_x; mxx += qhi_x; mny += qlo_y; mxy += qhi_y;
    }
    const int wX0 = max(0, (int)ceilf(fmaxf(mnx - 0.01f, -2.0f))), wY0 = max(0, (int)ceilf(fmaxf(mny - 0.01f, -2.0f)));
    const int wX1 = min(a.Ws - 1, (int)floorf(fminf(mxx + 0.01f, (float)a.Ws)));
    const int wY1 = min(a.Hs - 1, (int)floorf(fminf(mxy + 0.01f, (float)a.Hs)));
    const int ww = min(max(0, wX1 - wX0 + 1), 0xffff), wh = min(max(0, wY1 - wY0 + 1), 0x3fff);
    int4 rec;
    const bool empty = ww == 0 || wh == 0;    // keep the corner a valid texel: the gather prefetches relative to it
    rec.x = empty ? 0 : wX0; rec.y = empty ? 0 : wY0; rec.z = empty ? 0 : (ww | (wh << 16)); rec.w = __float_as_int(1.0f / (float)max(ww, 1));
    // bit 30: "pixels are at least a texel apart" on this tile -- J = d texel / d pixel has J00 - |J01| >= 1 and J11 - |J10| >= 1
    // on the tile's region.  Then, for a texel tau with owner pixel p0 and u = tau - t(p0), the pixel p0 + e with e_x = -sign(u_x)
    // lies at |t_x(p0+e) - tau_x| = J00 + |u_x| -/+ J01 e_y >= 1, i.e. has tent weight exactly 0 (same along y): only the 2x2
    // block of p0 towards tau contributes, and the gather reads 4 staged pixels instead of 9 (no-minification views, e.g.
    // stacks stored at >= the frame's resolution as the reference's are: mpi_h/w_scale 1.1, configs/mpv_base.txt:10-11).
    {
        float ax, ay;
        if constexpr (COORD == VL3D_COORD_UTILS_MPI) { ax = (float)(a.Ws - 1) / (float)a.Ws; ay = (float)(a.Hs - 1) / (float)a.Hs; }
        else { ax = a.sx; ay = a.sy; }
        bool apart = !a.gather9;
        for (int c = 0; c < 4; ++c) {     // region corners (tile + halo); J is monotone enough over <= 70 pixels for the 1e-3 margin
            const float cx = (float)a.col0 + a.pc + (float)((c & 1) ? tile_x * iw + iw - 1 + rh + 1 : ix0 - rh - 1);
            const float cy = (float)a.row0 + a.pc + (float)((c & 2) ? tile_y * ih + ih - 1 + rh + 1 : iy0 - rh - 1);
            const float X = h[0] * cx + h[1] * cy + h[2], Y = h[3] * cx + h[4] * cy + h[5], Z = h[6] * cx + h[7] * cy + h[8];
            const float iz2 = 1.0f / (Z * Z);
            const float j00 = ax * (h[0] * Z - X * h[6]) * iz2, j01 = ax * (h[1] * Z - X * h[7]) * iz2;
            const float j10 = ay * (h[3] * Z - Y * h[6]) * iz2, j11 = ay * (h[4] * Z - Y * h[7]) * iz2;
            apart = apart && (Z > 0.0f) && (j00 - fabsf(j01) >= 1.001f) && (j11 - fabsf(j10) >= 1.001f);
        }
        if (apart && !empty) rec.z |= 0x40000000;
    }
    if (a.quad_keep) {
        // tile culling: bit 31 of the size word = no pixel of the tile's region (interior + halo) can see a kept quad of this
        // plane -> the tile kernel skips the plane's sweep and writes zeros to the texels it owns
        const int qx0 = max(ix0 - rh, 0), qx1 = min(tile_x * iw + iw - 1 + rh, a.W - 1);
        const int qy0 = max(iy0 - rh, 0), qy1 = min(tile_y * ih + ih - 1 + rh, a.H - 1);
        float tnx = 1e30f, txx = -1e30f, tny = 1e30f, txy = -1e30f;
        for (int c = 0; c < 4; ++c) {
            const float cx = (float)a.col0 + a.pc + (float)((c & 1) ? qx1 : qx0), cy = (float)a.row0 + a.pc + (float)((c & 2) ? qy1 : qy0);
            const float X = h[0] * cx + h[1] * cy + h[2], Y = h[3] * cx + h[4] * cy + h[5], Z = h[6] * cx + h[7] * cy + h[8];
            const float ctx = texel_coord<COORD>(X / Z, (float)a.Ws / 2.0f, (float)(a.Ws - 1), a.sx, a.ox);
            const float cty = texel_coord<COORD>(Y / Z, (float)a.Hs / 2.0f, (float)(a.Hs - 1), a.sy, a.oy);
            tnx = fminf(tnx, ctx); txx = fmaxf(txx, ctx); tny = fminf(tny, cty); txy = fmaxf(txy, cty);
        }
        if (!box_touches_kept_quad(a, d, tnx, txx, tny, txy)) rec.z |= (int)0x80000000;
    }
    reinterpret_cast<int4 *>(win)[i] = rec;     // [tile][plane]: one contiguous run per workgroup
}

// tile-exact layout: the lattice position of texel (x, y) of the (window of the) plane -- texel coordinate minus the index of the tile
// that holds it (tiles of q_th x q_tw texels; (x + 1/2) / tw in fp32 is exact to the quad for plane coordinates < 2^22).  The inverse
// homographies of the plan map LATTICE positions to pixels: both copies of a border sample have the same owner pixel.
__device__ __forceinline__ void lattice_texel(const RenderArgs &a, int x, int y, float &lx, float &ly) {
    lx = (float)x; ly = (float)y;
    if (a.q_th) {      // uniform
        lx -= fminf(floorf(((float)x + a.q_x0 + 0.5f) * (1.0f / (float)a.q_tw)), (float)(a.QW - 1));
        ly -= fminf(floorf(((float)y + a.q_y0 + 0.5f) * (1.0f / (float)a.q_th)), (float)(a.QH - 1));
    }
}

// owner pixel (float, before rounding) of texel (tx,ty) on plane d, relative to this window's pixel origin
__device__ __forceinline__ void owner_pixel(const float *__restrict__ hi, float tx, float ty, float pc, int col0, int row0,
                                            float &px, float &py) {
    const float X = hi[0] * tx + hi[1] * ty + hi[2];
    const float Y = hi[3] * tx + hi[4] * ty + hi[5];
    const float Z = hi[6] * tx + hi[7] * ty + hi[8];
    const float rz = fast_rcp(Z);
    px = X * rz - pc - (float)col0;
    py = Y * rz - pc - (float)row0;
}

// ---- the optimiser step in the owner's store (vl3d_render_bwd_adam) ---------------------------------------------------------------------
// Texel (cx, cy) of the compact window on plane d: false when it lies outside its plane's box (no tap of this iteration can reach it: its
// update stays deferred like that of every texel outside the window); else the step its bookkeeping tile is current for and the texel's
// BYTE offset inside a frame of the full (D,T,Hs,Ws,4) parameter / moment tensors (32 bits: one lane offset for p, m and v of every frame,
// the frame's base is uniform -- global_load / global_store v_off, s[base:base+1]).
__device__ __forceinline__ bool adam_texel(const RenderArgs &a, int d, int cx, int cy, int &from, unsigned &off) {
    const int X = a.ad.x0 + cx, Y = a.ad.y0 + cy;
    if (a.ad.boxes) {      // (d is uniform: four scalar loads)
        const cint_p b = (cint_p)a.ad.boxes + 4 * d;
        if (Y < b[0] || Y >= b[1] || X < b[2] || X >= b[3]) return false;
    }
    static_assert(vl3d_adam::TS == 8, "bookkeeping tiles of 8 x 8 texels");
    from = a.ad.last_step[((size_t)d * a.ad.tiles_y + (Y >> 3)) * a.ad.tiles_x + (X >> 3)];
    off = (unsigned)(Y * a.ad.Ws + X) << 4;
    return true;
}
// ... for a caller that already knows the texel is one the optimiser steps now (the class table of a tile-culled model holds the box test).
// PACKED storage: the texel's byte offset inside frame 0 of its block's slots in the pool (the frames of a dynamic block are 1 KiB apart).
__device__ __forceinline__ void adam_texel_inbox(const RenderArgs &a, int d, int cx, int cy, int &from, size_t &off) {
    const int X = a.ad.x0 + cx, Y = a.ad.y0 + cy;
    const size_t tile = ((size_t)d * a.ad.tiles_y + (Y >> 3)) * a.ad.tiles_x + (X >> 3);
    from = a.ad.last_step[tile];
    if (a.ad.blocks) off = ((size_t)(a.ad.blocks[tile] >> 1) * 64 + (size_t)((Y & 7) * 8 + (X & 7))) << 4;      // (a dynamic texel's block: >= 0, dynamic)
    else off = (size_t)((unsigned)(Y * a.ad.Ws + X) << 4);
}
// byte offset of frame t of plane d (uniform): dense tensors / packed pools
__device__ __forceinline__ size_t adam_frame_base(const RenderArgs &a, int d, int t) {
    return a.ad.blocks ? (size_t)t * 1024 : ((size_t)d * a.T + t) * ((size_t)a.ad.Hs * a.ad.Ws * 16);
}
// One Adam step of one texel and frame with gradient g: `pcur` is the parameter current for step - 1 (the compact copy holds it: the
// catch-up replayed the deferred steps into it, so only the two moments are replayed here -- multiplications), (p, m, v) are written.
// fb: byte offset of the frame (uniform), off: of the texel inside it.
// (m0, v0: the two moments as stored -- the caller may have requested them long before the gradient was complete)
template <typename OFF>
__device__ __forceinline__ void adam_texel_step_mv(const RenderArgs &a, size_t fb, OFF off, int from, f4 pcur, f4 g, f4 m0, f4 v0) {
    char *pb = reinterpret_cast<char *>(a.ad.p) + fb, *mb = reinterpret_cast<char *>(a.ad.m) + fb, *vb = reinterpret_cast<char *>(a.ad.v) + fb;
    float4 mm = make_float4(m0.x, m0.y, m0.z, m0.w), vv = make_float4(v0.x, v0.y, v0.z, v0.w);
    vl3d_adam::replay_moments(mm, vv, from, a.ad.step - 1, a.ad.beta1, a.ad.beta2);
    float4 pp = make_float4(pcur.x, pcur.y, pcur.z, pcur.w);
    vl3d_adam::adam_upd4(pp, make_float4(g.x, g.y, g.z, g.w), mm, vv, a.ad.lr_bc1, a.ad.beta1, a.ad.beta2, a.ad.eps, a.ad.bc2s);
    __builtin_nontemporal_store(f4{pp.x, pp.y, pp.z, pp.w}, reinterpret_cast<f4 *>(pb + (size_t)off));
    __builtin_nontemporal_store(f4{mm.x, mm.y, mm.z, mm.w}, reinterpret_cast<f4 *>(mb + (size_t)off));
    __builtin_nontemporal_store(f4{vv.x, vv.y, vv.z, vv.w}, reinterpret_cast<f4 *>(vb + (size_t)off));
}
template <typename OFF>
__device__ __forceinline__ void adam_texel_step(const RenderArgs &a, size_t fb, OFF off, int from, f4 pcur, f4 g) {
    const char *mb = reinterpret_cast<const char *>(a.ad.m) + fb, *vb = reinterpret_cast<const char *>(a.ad.v) + fb;
    adam_texel_step_mv(a, fb, off, from, pcur, g, *reinterpret_cast<const f4 *>(mb + (size_t)off), *reinterpret_cast<const f4 *>(vb + (size_t)off));
}

// Owner table: for every texel of every plane, the tile that owns it (the tile of its owner pixel p0 = clamp_to_frame(
// round(H_d^-1 tau))) and p0's index in that tile's pixel region, packed as tile << 10 | index.  Frame independent, so it
// is built once per call (D*Hs*Ws entries) and read once per frame by the gather, which then needs no inverse homography,
// reciprocal, rounding or range tests per texel (~200 of its ~400 VALU issue cycles, profiles/microbench/isa_cost.py).
__global__ __launch_bounds__(256) void bwd_owner_table_k(RenderArgs a, int iw, int ih, int rh, int tiles_x, unsigned short *owner, int rw = 64,
                                                         int slot_bits = 10) {
    if (!reinterpret_cast<const int *>(a.plan)[0]) return;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int d = blockIdx.z;
    if (x >= a.Ws || y >= a.Hs) return;
    float qx, qy, lx, ly;
    lattice_texel(a, x, y, lx, ly);
    owner_pixel(a.plan + PLAN_HDR + PLAN_REC * d, lx, ly, a.pc, a.col0, a.row0, qx, qy);
    // owner = nearest FRAME pixel: texels just outside the frame still collect taps of the border pixels
    const float rxf = fminf(fmaxf(rintf(qx), 0.0f), (float)(a.W - 1)), ryf = fminf(fmaxf(rintf(qy), 0.0f), (float)(a.H - 1));
    const int rx = (int)rxf, ry = (int)ryf;
    // tile of the owner pixel: floor((r + 0.5) / size) in fp32 is exact for frame coordinates (< 2^22) -- no integer division
    const int tx = (int)((rxf + 0.5f) * (1.0f / (float)iw)), ty = (int)((ryf + 0.5f) * (1.0f / (float)ih));
    const unsigned lc = (unsigned)((ry - ty * ih + rh) * rw + (rx - tx * iw + rh));     // rw: the tile kernel's region width
    // a tile's window only holds texels owned by itself or tiles a few steps away (the window is the bounding box of the tile's
    // image; under the plan's rotation / magnification limits its corners reach < 4 tiles of the 16-row regions, < 8 of the flat
    // 64 x 8 ones along y), which the low bits of each tile coordinate tell apart: 16 bits per texel = slot_bits for the slot
    // (10 for the 1024-thread tile kernel, 9 for the 512-thread pair kernels), the rest for (tile_y & 7 | 15, tile_x & 7)
    const unsigned ymask = slot_bits == 9 ? 15u : 7u;
    const unsigned ent = ((((unsigned)ty & ymask) << 3 | (unsigned)(tx & 7)) << slot_bits) | lc;
    // four lanes' entries leave as ONE 8-byte store from the first of them where the row is 8-byte aligned (the 2-byte stores were a third of
    // this pre-pass for a single frame); the lane -> texel mapping stays one to one: the zero fill below needs contiguous lanes
    const unsigned e1 = __shfl_down(ent, 1, 64), e2 = __shfl_down(ent, 2, 64), e3 = __shfl_down(ent, 3, 64);
    unsigned short *dst = owner + ((size_t)d * a.Hs + y) * a.Ws + x;
    if ((a.Ws & 3) == 0 && x + 3 < a.Ws) {      // (uniform per row of the block but for its last lanes; x is a multiple of 4 where lane & 3 == 0)
        if ((threadIdx.x & 3) == 0) *reinterpret_cast<uint2 *>(dst) = make_uint2(ent | e1 << 16, e2 | e3 << 16);
    } else {
        *dst = (unsigned short)ent;
    }
    // same pass (it already has the texel's owner pixel): texels no tile is certain to own -- owner pixel on or outside the frame's
    // border ring -- are zero-filled for all T frames here, so nothing memsets the gradient; the tile kernel runs after this
    // kernel and overwrites the border ring's texels it does own
    int cls = 1;
    if (a.ad.p && a.ad.cls) {      // tile-culled model under the fused step: classify the texel once (frame independent), the gather reads a byte
        const int X = a.ad.x0 + x, Y = a.ad.y0 + y;
        bool inbox = true;
        if (a.ad.boxes) {
            const cint_p b = (cint_p)a.ad.boxes + 4 * d;
            inbox = !(Y < b[0] || Y >= b[1] || X < b[2] || X >= b[3]);
        }
        cls = inbox ? vl3d_adam::texel_class(vl3d_adam::Quads{a.quad_keep, a.ad.quad_dyn, a.QH, a.QW, a.q_th, a.q_tw}, d, X, Y, a.ad.Hs, a.ad.Ws) : 3;
        // the record the gather reads with its owner entry, long before it needs it: class | (step the texel's bookkeeping tile is current
        // for) << 2, and the texel's 16-byte slot inside a frame of the parameter / moment tensors (dense) or pools (packed) -- so that the
        // owner's store asks for nothing but the two moments (no class byte -> step table / block table -> moments chain of dependent loads)
        int rfrom = 0;
        size_t roff = 0;
        if (cls == 1) adam_texel_inbox(a, d, x, y, rfrom, roff);
        a.ad.cls[((size_t)d * a.Hs + y) * a.Ws + x] = make_uint2((unsigned)cls | ((unsigned)rfrom << 2), (unsigned)(roff >> 4));
    }
    const bool safe = (qx > 0.5f) && (qx < (float)a.W - 1.5f) && (qy > 0.5f) && (qy < (float)a.H - 1.5f);
    if (safe) return;
    const size_t frame = (size_t)a.Hs * a.Ws;
    if (a.ad.p) {
        // fused optimiser step: the texels the tile kernel will NOT reach take their zero-gradient step here, exactly once: the tile kernel's
        // gather visits the texels of its window (bwd_windows_k, run before this kernel) the table assigns to it, so a texel outside its owner
        // tile's window is visited by nobody (tiles far enough apart to share a code have disjoint windows: the table's premise)
        const int4 rec = reinterpret_cast<const int4 *>(reinterpret_cast<const int *>(a.plan) + plan_win_off(a.D))[(size_t)(ty * tiles_x + tx) * a.D + d];
        const int ww = rec.z & 0xffff, wh = (rec.z >> 16) & 0x3fff;
        // (bit 31: a tile-culled model's tile that skips this plane altogether -- none of its pixels sees a kept quad -- visits nothing)
        if (rec.z >= 0 && x >= rec.x && x < rec.x + ww && y >= rec.y && y < rec.y + wh) return;
        int from;
        unsigned off;
        if (cls == 2) {      // a static texel's gradient is summed over the frames by the step kernel: zeros, as without the fused step
            float4 *g = reinterpret_cast<float4 *>(a.g_stack) + (size_t)d * a.T * frame + (size_t)y * a.Ws + x;
            for (int t = 0; t < a.T; ++t, g += frame) *g = make_float4(0.f, 0.f, 0.f, 0.f);
            return;
        }
        if (cls != 1) return;
        const f4 *pc = reinterpret_cast<const f4 *>(a.stack) + (size_t)d * a.T * frame + (size_t)y * a.Ws + x;
        if (a.ad.cls) {
            size_t offp;
            adam_texel_inbox(a, d, x, y, from, offp);
            for (int t = 0; t < a.T; ++t, pc += frame) adam_texel_step(a, adam_frame_base(a, d, t), offp, from, *pc, f4{0.f, 0.f, 0.f, 0.f});
            return;
        }
        if (!adam_texel(a, d, x, y, from, off)) return;
        for (int t = 0; t < a.T; ++t, pc += frame) adam_texel_step(a, adam_frame_base(a, d, t), off, from, *pc, f4{0.f, 0.f, 0.f, 0.f});
        return;
    }
    if (a.g_f16) {
        float2 *g = reinterpret_cast<float2 *>(a.g_stack) + (size_t)d * a.T * frame + (size_t)y * a.Ws + x;
        for (int t = 0; t < a.T; ++t, g += frame) *g = make_float2(0.f, 0.f);
    } else {
        float4 *g = reinterpret_cast<float4 *>(a.g_stack) + (size_t)d * a.T * frame + (size_t)y * a.Ws + x;
        for (int t = 0; t < a.T; ++t, g += frame) *g = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (a.g_mask) {      // the loop-mask gradient follows the stack gradient's ownership
        float *g = a.g_mask + (size_t)d * a.T * frame + (size_t)y * a.Ws + x;
        for (int t = 0; t < a.T; ++t, g += frame) *g = 0.f;
    }
}

// ... for a SINGLE frame (T = 1: cfg2, every stage-1 iteration), four consecutive texels of a row per thread: the table is written and read
// once there, so this pass is a quarter of the backward -- one 8-byte store of four entries without the three shuffles, row / plane
// arithmetic once per four texels (0.414 -> 0.389 ms for a 720p backward when it was tried in round 3; not usable for T > 1, where the zero
// fill below walks the frames at a 64-byte lane stride: 15.8 -> 19.4 ms on a 1.1x stack).  Same entries and the same zero fill as
// bwd_owner_table_k (the owner pixel of every texel is computed by the same function on the same inputs).
__global__ __launch_bounds__(256) void bwd_owner_table4_k(RenderArgs a, int iw, int ih, int rh, unsigned short *owner, int rw, int slot_bits) {
    if (!reinterpret_cast<const int *>(a.plan)[0]) return;
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int d = blockIdx.z;
    if (x0 >= a.Ws || y >= a.Hs) return;
    const float *hi = a.plan + PLAN_HDR + PLAN_REC * d;
    const unsigned ymask = slot_bits == 9 ? 15u : 7u;
    const float inv_iw = 1.0f / (float)iw, inv_ih = 1.0f / (float)ih;
    unsigned ent[4];
    bool unsafe[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float qx, qy, lx, ly;
        lattice_texel(a, x0 + k, y, lx, ly);
        owner_pixel(hi, lx, ly, a.pc, a.col0, a.row0, qx, qy);
        const float rxf = fminf(fmaxf(rintf(qx), 0.0f), (float)(a.W - 1)), ryf = fminf(fmaxf(rintf(qy), 0.0f), (float)(a.H - 1));
        const int rx = (int)rxf, ry = (int)ryf;
        const int tx = (int)((rxf + 0.5f) * inv_iw), ty = (int)((ryf + 0.5f) * inv_ih);
        const unsigned lc = (unsigned)((ry - ty * ih + rh) * rw + (rx - tx * iw + rh));
        ent[k] = ((((unsigned)ty & ymask) << 3 | (unsigned)(tx & 7)) << slot_bits) | lc;
        unsafe[k] = !((qx > 0.5f) && (qx < (float)a.W - 1.5f) && (qy > 0.5f) && (qy < (float)a.H - 1.5f));
    }
    const size_t row = ((size_t)d * a.Hs + y) * a.Ws;
    unsigned short *dst = owner + row + x0;
    if ((a.Ws & 3) == 0) {      // (x0 + 3 < Ws then; rows are 8-byte aligned)
        *reinterpret_cast<uint2 *>(dst) = make_uint2(ent[0] | ent[1] << 16, ent[2] | ent[3] << 16);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (x0 + k < a.Ws) dst[k] = (unsigned short)ent[k];
    }
    // texels no tile is certain to own: zero gradient here (T = 1: one frame), the tile kernel overwrites those it does own
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!unsafe[k] || x0 + k >= a.Ws) continue;
        if (a.g_f16) reinterpret_cast<float2 *>(a.g_stack)[row + x0 + k] = make_float2(0.f, 0.f);
        else reinterpret_cast<float4 *>(a.g_stack)[row + x0 + k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.g_mask) a.g_mask[row + x0 + k] = 0.f;
    }
}

__global__ __launch_bounds__(256) void bwd_fill_zero_if_infeasible_k(float2 *g, size_t n8, const float *plan) {      // n8: 8-byte units
    if (reinterpret_cast<const int *>(plan)[0]) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) g[i] = make_float2(0.f, 0.f);
}
__global__ __launch_bounds__(256) void bwd_fill_zero_f32_if_infeasible_k(float *g, size_t n, const float *plan) {
    if (reinterpret_cast<const int *>(plan)[0]) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) g[i] = 0.f;
}

template <int COORD, int BORDER, int ORDER, int RACT, int AACT, int ROWS, bool REG, bool F16, bool CULL = false, bool MASK = false, bool ADAM = false, int RWT = 64>
// (the culled instantiation of the utils_mpi coordinate convention -- a cross-check convention, its texel coordinates cost a
// reciprocal more -- does not fit 64 VGPRs: it takes the 128-register budget (one workgroup per CU) rather than spill)
// MASK: stage 1's loop-mask texture as a fifth channel of the same sweep and gather (a fifth staged value, a fifth accumulator, one
// 4-byte store per owned texel); T = 1 there, so the instantiation simply takes the 128-register budget.
// RWT: region width in pixels.  64 = one wave per region row (1024 threads at 16 rows); 32 = the frame pairs' region shape for ONE frame
// (512 threads: two workgroups per CU at the 128-register budget, so that one workgroup's tap / moment latencies run under the other's
// arithmetic -- the shape of the tile-culled fused step, where a workgroup sweeps a handful of planes with nothing else resident).
__global__ __launch_bounds__(RWT *ROWS, ((REG || MASK || (CULL && COORD == VL3D_COORD_UTILS_MPI)) ? 4 : ((CULL && RWT == 32) ? 6 : 8))) void render_bwd_tile_k(RenderArgs a) {
    static_assert(!(MASK && (CULL || F16 || ORDER != VL3D_ACT_POST)), "the loop-mask channel: dense fp32 stage-1 stacks, sample-then-activate");
    static_assert(!(ADAM && (F16 || MASK || !REG)), "the fused optimiser step: fp32 stacks, the instantiation with the 128-register budget");
    static_assert(RWT == 64 || RWT == 32, "region width");
    if (!reinterpret_cast<const int *>(a.plan)[0]) return;
    constexpr int RW = RWT;                                  // (shadows the 64-wide default of the namespace)
    constexpr int SLOT_BITS = RW * ROWS > 512 ? 10 : 9;      // the owner table's slot field (bwd_owner_table_k)
    constexpr unsigned SLOT_MASK = (1u << SLOT_BITS) - 1u;
    constexpr int NT = RW * ROWS;
    __shared__ float s_gm[MASK ? 2 : 1][MASK ? NT : 1];     // MASK: gradient w.r.t. the sampled mask logit of this pixel on this plane
    // REG: the layer-space smoothness regularisers (MPV.py:517-531) are differentiated here as well: their gradient at a pixel is
    // decoded from the sign words the forward stored (reg_grad) -- no neighbours' layer values, no extra halo.  The sparsity-sum
    // gradients ride in this instantiation too (g_reg == NULL: sparsity only).
    constexpr int RH = 1;
    // per-plane staging of the region's pixels, double buffered so one barrier per plane suffices
    __shared__ float4 s_g[2][NT];     // gradient w.r.t. the sampled (POST) / activated (PRE) value of this pixel on this plane
    __shared__ float2 s_t[2][NT];     // its texel coordinates (tx,ty) (true ones also where the plane does not cover it: then g = 0)
    const int tid = threadIdx.x, lane = tid & (RW - 1), row = tid / RW;
    // 1-D grid, XCD-aware order: every XCD walks a contiguous run of tiles (row-major within a frame), so a tile's halo
    // rows and its neighbours' taps hit the same L2
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_x = bid % a.tiles_x, rest = bid / a.tiles_x;
    const int tile_y = rest % a.tiles_y, t = rest / a.tiles_y;
    const int rx0 = tile_x * (RW - 2 * RH) - RH, ry0 = tile_y * (ROWS - 2 * RH) - RH;
    const int x = rx0 + lane, y = ry0 + row;
    const bool inimg = (x >= 0) && (x < a.W) && (y >= 0) && (y < a.H);
    // owned (interior) pixel range of this workgroup, clipped to the frame: [ix0,ix1] x [iy0,iy1]
    const int ix0 = max(rx0 + RH, 0), ix1 = min(rx0 + RW - 1 - RH, a.W - 1);
    const int iy0 = max(ry0 + RH, 0), iy1 = min(ry0 + ROWS - 1 - RH, a.H - 1);
    const float px = (float)(a.col0 + x) + a.pc, py = (float)(a.row0 + y) + a.pc;
    constexpr size_t TEXB = F16 ? 8 : 16;
    const size_t frame = (size_t)a.Hs * a.Ws * 4;
    const size_t plane_stride = (size_t)a.T * frame;
    const size_t plane_stride_b = (size_t)a.T * a.Hs * a.Ws * TEXB;
    const char *plane = reinterpret_cast<const char *>(a.stack) + (size_t)t * a.Hs * a.Ws * TEXB;
    char *gplane = reinterpret_cast<char *>(a.g_stack) + (size_t)t * a.Hs * a.Ws * TEXB;       // gradient texels = stack texels
    float Gr = 0.f, Gg = 0.f, Gb = 0.f, gA = 0.f, S = 0.f, gN1 = 0.f, gN2 = 0.f;   // gN1 + gN2*a_k = d(sparsity sums)/da_k
    if (inimg) {
        const size_t pix = ((size_t)t * a.H + y) * a.W + x;
        Gr = a.g_rgb[pix * 3 + 0]; Gg = a.g_rgb[pix * 3 + 1]; Gb = a.g_rgb[pix * 3 + 2];
        gA = a.g_alpha ? a.g_alpha[pix] : 0.0f;
        S = dot3p(Gr, a.rgb[pix * 3 + 0], Gg, a.rgb[pix * 3 + 1], Gb, a.rgb[pix * 3 + 2], gA * a.alpha[pix]);
        // the sparsity-sum gradients ride in the REG instantiation only (launch_t): the plain one is at its 64-VGPR budget
        if constexpr (REG) if (a.g_asum) { gN1 = a.g_asum[pix * 2 + 0]; gN2 = 2.0f * a.g_asum[pix * 2 + 1]; }
    }
    float gL = 0.f;
    if constexpr (MASK) if (inimg) gL = a.g_label[((size_t)t * a.H + y) * a.W + x];
    const float *mplane = MASK ? a.mask + (size_t)t * a.Hs * a.Ws : nullptr;
    float *gmplane = MASK ? a.g_mask + (size_t)t * a.Hs * a.Ws : nullptr;
    const size_t mplane_stride = (size_t)a.T * a.Hs * a.Ws;
    float Tr = 1.0f, P = 0.0f;
    float gsx_c = 0.f, gsy_c = 0.f, gsx_a = 0.f, gsy_a = 0.f;
    unsigned fl = 0u;
    const uint2 *sgp = nullptr;      // this pixel's sign words of planes 0-3 (clamped into the frame: the loads are unconditional)
    uint2 wg_own = make_uint2(0u, 0u), wg_l = wg_own, wg_u = wg_own, wg_p = wg_own;      // sign words of the current group of four planes (planes 0, 1 | 2, 3)
    int wg_idx = -1;
    const bool has_l = inimg && x >= 1, has_u = inimg && y >= 1;
    const bool reg_on = REG && a.g_reg != nullptr;
    if constexpr (REG) if (reg_on) {
        gsx_c = a.g_reg[0]; gsy_c = a.g_reg[1]; gsx_a = a.g_reg[2]; gsy_a = a.g_reg[3];
        const int xc = min(max(x, 0), a.W - 1), yc = min(max(y, 0), a.H - 1);
        sgp = reinterpret_cast<const uint2 *>(a.reg_signs) + ((size_t)t * a.H + yc) * a.W + xc;
        if (inimg) fl = a.reg_flags[(size_t)y * a.W + x];
    }
    const size_t sg_plane = (size_t)a.T * a.H * a.W;
    constexpr bool provider = true;
    const TapStep st = make_tap_step<F16>(a.Hs, a.Ws);
    // this tile's texel windows, one int4 per plane (bwd_windows_k)
    const unsigned my_tile_id = (unsigned)(tile_y * a.tiles_x + tile_x);
    const unsigned my_tile = (unsigned)((tile_y & (SLOT_BITS == 9 ? 15 : 7)) << 3 | (tile_x & 7));      // the owner table's code of this tile
    const unsigned toff_thread = (unsigned)(row * a.Ws + lane);   // texel (lane, row) of a window, relative to its corner
    const cint_p wrec = (cint_p)a.plan + plan_win_off(a.D) + (size_t)my_tile_id * a.D * 4;
    int nswept = 0;
    for (int d = 0; d < a.D; ++d, plane += plane_stride_b, gplane += plane_stride_b, mplane += mplane_stride, gmplane += mplane_stride) {
        float h[VL3D_HN];
        load_uniform(a.homos + VL3D_HS * d, h);
        // texel window of this tile on plane d (wave = window row, lane = window column); bit 31: culled for this tile
        const int X0 = wrec[4 * d], Y0 = wrec[4 * d + 1], wwh = wrec[4 * d + 2];
        const int ww = wwh & 0xffff, wh = (wwh >> 16) & 0x3fff;
        const bool culled = CULL && wwh < 0;
        const bool apart = (wwh & 0x40000000) != 0;      // pixels >= 1 texel apart on this tile: 2x2 gather (bwd_windows_k)
        // staging buffer of this plane: alternates over the planes that are actually swept (a culled plane has no barrier)
        const int buf = CULL ? (nswept & 1) : (d & 1);
        if constexpr (CULL) nswept += culled ? 0 : 1;
        // (a skipped plane whose owned texels need no zero fill -- grad_culled_unwritten -- costs one scalar record, not a table load per
        // thread.  In the instantiation with regularisers only, the one a shipped stage-2 iteration runs: the plain culled kernel sits
        // exactly at its 64-register budget and the extra branch spilled 12 bytes; it keeps filling zeros, which is always correct.)
        if constexpr (REG || RWT == 32) { if (CULL && culled && (ADAM || a.grad_culled_unwritten)) continue; }
        const unsigned win0 = (unsigned)(Y0 * a.Ws + X0);          // frame texel index of the window's corner (uniform)
        const unsigned short *oplane = a.owner + (size_t)d * a.Hs * a.Ws;
        // this thread's first owner-table entry, requested now so that it arrives in the shadow of the sweep.  Unconditional
        // (threads outside the window read a neighbouring entry -- the table is padded by ROWS rows -- and ignore it): no
        // branch around the load, so no merged wait counters.
        const unsigned e0 = oplane[win0 + toff_thread];
        // (fused step of a tile-culled model) this thread's record of the same texel: class, step, slot (bwd_owner_table_k)
        const uint2 *rplane = nullptr;
        uint2 rec0 = make_uint2(0u, 0u);
        if constexpr (ADAM && CULL) {
            rplane = a.ad.cls + (size_t)d * a.Hs * a.Ws;
            rec0 = rplane[win0 + toff_thread];
        }
        if (CULL && culled) {
            // tile culling: no pixel of the region sees a kept quad of this plane -- its alpha is exactly 0 for all of them, the
            // composite state does not move, and the texels this tile owns get a zero gradient (written: nothing memsets it)
            // (... unless the caller never reads the gradient of culled texels -- every texel this tile owns on this plane is one: the box
            // test of bwd_windows_k is two texels wider than the region's footprint, a texel's class looks one texel around it)
            const f4 z = f4{0.f, 0.f, 0.f, 0.f};
            if (row < wh && lane < ww && (e0 >> SLOT_BITS) == my_tile)
                store_grad_texel<F16>(gplane, (win0 + toff_thread) << 4, z);
            for (int wy = row; wy < wh; wy += ROWS)
                for (int wx = lane + (wy == row ? RW : 0); wx < ww; wx += RW) {
                    const unsigned tix = win0 + (unsigned)(wy * a.Ws + wx);
                    if ((oplane[tix] >> SLOT_BITS) == my_tile)
                        store_grad_texel<F16>(gplane, tix << 4, z);
                }
            continue;
        }
        // (2) sample this pixel on plane d, composite backward, stage (tx,ty,g) in LDS   (branch-free taps)
        float2 tc = make_float2(0.f, 0.f);        // pixels outside the frame: any finite coordinate (their gradient is 0)
        float4 gval = make_float4(0.f, 0.f, 0.f, 0.f);
        f4 o = f4{0.f, 0.f, 0.f, 0.f}, pre = o;
        Taps2 tp{};
        float gm = 0.f, msig = 0.f;
        if (inimg) {
            if constexpr (CULL) tp = make_taps2<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, plane_cull(a, d));
            else tp = make_taps2<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy);
            const char *src = plane;
            if VL3D_ABLATE(a.ablate, 4) {   // measurement only: all taps from a 64 KiB cache-resident window
                src = reinterpret_cast<const char *>(a.stack);
                tp.off &= 0xfff0u;
            }
            typename TapVal<F16, ORDER>::type tv[4];
            load_taps2<F16>(src, tp, st, tv);
            float mt[4];
            if constexpr (MASK) load_mask_taps(mplane, tp.off, st, mt);
            o = shade2<ORDER, RACT, AACT>(tp, tv, &pre);                 // o.w already 0 when the plane does not cover the pixel
            if constexpr (MASK) msig = act_fwd<VL3D_ACT_SIGMOID>(mask_blend(mt, tp.w));
        }
        f4 sg = f4{0.f, 0.f, 0.f, 0.f};
        if constexpr (REG) if (reg_on) {      // uniform
            if ((d >> 2) != wg_idx) {      // (not "d & 3 == 0": a culled plane skips this block)
                wg_idx = d >> 2;
                const uint2 *w = sgp + (size_t)(d >> 2) * sg_plane;
                const uint2 zz = make_uint2(0x55555555u, 0x55555555u);      // a neighbour that does not exist: sign 0 everywhere
                wg_own = w[0]; wg_l = w[has_l ? -1 : 0]; wg_u = w[has_u ? -(ptrdiff_t)a.W : 0];
                if (!has_l) wg_l = zz;
                if (!has_u) wg_u = zz;
                if (fl & 12u) wg_p = reinterpret_cast<const uint2 *>(a.reg_patch)[w - reinterpret_cast<const uint2 *>(a.reg_signs)];
            }
            const bool hi = (d & 2) != 0;       // uniform
            sg = reg_grad32(hi ? wg_own.y : wg_own.x, hi ? wg_l.y : wg_l.x, hi ? wg_u.y : wg_u.x, hi ? wg_p.y : wg_p.x, 16u * (d & 1), fl,
                            f4{gsx_c, gsx_c, gsx_c, gsx_a}, f4{gsy_c, gsy_c, gsy_c, gsy_a});
        }
        if (inimg) {
            const float q = dot3p(Gr, o.x, Gg, o.y, Gb, o.z, gA);
            const float w = o.w * Tr;
            // loop mask: d label / d (sampled logit) = g_label w_k sigmoid'; w_k is 0 where the plane does not cover the pixel, and the
            // colour composite's weights get nothing back from the label (detached, MPI.py:577-579)
            if constexpr (MASK) gm = gL * w * (msig * (1.0f - msig));
            P = fmaf(w, q, P);
            const float om = 1.0f - o.w;
            const float behind = (om > 1e-12f) ? (S - P) * fast_rcp(om) : 0.0f;
            // grad wrt activated (c, a); spelt exactly like VL3D_PAIR_GRAD of the frame-pair kernels (compared bit for bit)
            if constexpr (REG) sg.w += fmaf(gN2, o.w, gN1);
            gval = make_float4(fmaf(w, Gr, sg.x), fmaf(w, Gg, sg.y), fmaf(w, Gb, sg.z), fmaf(Tr, q, -behind) + sg.w);
            Tr *= om;
            if constexpr (ORDER == VL3D_ACT_POST)
                gval = make_float4(gval.x * act_bwd<RACT>(pre.x, o.x), gval.y * act_bwd<RACT>(pre.y, o.y),
                                   gval.z * act_bwd<RACT>(pre.z, o.z), gval.w * act_bwd<AACT>(pre.w, o.w));
            // every frame pixel stages its true coordinates (the 2x2 gather picks its block from the owner pixel's); a pixel the
            // plane does not cover, or a layer-only halo pixel, provides a zero gradient instead of a zero weight
            tc = make_float2(tp.tx, tp.ty);
            if (!(tp.cov > 0.0f && provider)) gval = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        s_t[buf][tid] = tc;
        s_g[buf][tid] = gval;
        if constexpr (MASK) s_gm[buf][tid] = gm;
        // (fused step of a tile-culled model) the moments and the current parameter of this thread's first texel are requested HERE, in front
        // of the barrier: they arrive while the workgroup meets and the gather reads LDS.  Unconditional (a thread without a dynamic texel of
        // its own reads slot 0 and drops it): no branch around the loads, no merged wait counters.
        f4 pre_m = f4{0.f, 0.f, 0.f, 0.f}, pre_v = pre_m, pre_p = pre_m;
        if constexpr (ADAM && CULL) {
            const bool mine0 = row < wh && lane < ww && (e0 >> SLOT_BITS) == my_tile && (rec0.x & 3u) == 1u;
            const size_t fb = adam_frame_base(a, d, t), o = mine0 ? (size_t)rec0.y << 4 : 0;
            pre_m = *reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(a.ad.m) + fb + o);
            pre_v = *reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(a.ad.v) + fb + o);
            pre_p = load_texel<false>(plane, mine0 ? (win0 + toff_thread) << 4 : 0u);
            asm volatile("" ::: "memory");      // keep the requests in front of the barrier
        }
        __syncthreads();   // staging of plane d visible (the other buffer may still be read by slower waves: not touched here)
        // (3) every texel of this tile's window that the owner table assigns to this tile gathers its taps from the 3x3
        //     pixels around its owner pixel.  Wave = window row, lane = window column: uniform row bases, no index arithmetic.
        if VL3D_ABLATE(a.ablate, 1) continue;
        auto gather = [&](unsigned e, int wx, int wy, unsigned tix, uint2 rec = make_uint2(0u, 0u), bool pre = false) {   // tix = frame texel index of window texel (wx, wy)
            if ((e >> SLOT_BITS) != my_tile) return;
            // fixed trip count, constant LDS offsets; weights clamp to 0 for non-contributing pixels (|J^-1|_inf < 1.4)
            const int lc = (int)(e & SLOT_MASK);
            const f2 tau = f2{(float)(X0 + wx), (float)(Y0 + wy)};
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
            float accm = 0.f;
            if VL3D_ABLATE(a.ablate, 8) {
            } else if (apart) {
                // 2x2 block of the owner pixel towards tau, summed in the 3x3 loop's order: the five pixels left out have weight
                // exactly 0 there, so both gathers give the same bits
                const f2 c0 = *reinterpret_cast<const f2 *>(&s_t[buf][lc]);
                const int li0 = lc - (tau.x < c0.x ? 1 : 0) - (tau.y < c0.y ? RW : 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int li = li0 + (k >> 1) * RW + (k & 1);
                    const f2 dc = *reinterpret_cast<const f2 *>(&s_t[buf][li]) - tau;
                    const float wgt = tent_weight(dc.x) * tent_weight(dc.y);
                    acc += *reinterpret_cast<const f4 *>(&s_g[buf][li]) * wgt;
                    if constexpr (MASK) accm = fmaf(s_gm[buf][li], wgt, accm);
                }
            } else {
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int li = lc + dy * RW + dx;
                    const f2 dc = *reinterpret_cast<const f2 *>(&s_t[buf][li]) - tau;
                    const float wgt = tent_weight(dc.x) * tent_weight(dc.y);
                    if (!a.gather9 && __builtin_amdgcn_ballot_w64(wgt != 0.0f) == 0ull) continue;      // (exact zeros for the whole wave: pair_gather_plane)
                    acc += *reinterpret_cast<const f4 *>(&s_g[buf][li]) * wgt;
                    if constexpr (MASK) accm = fmaf(s_gm[buf][li], wgt, accm);
                }
            }
            if constexpr (ORDER == VL3D_ACT_PRE) {   // d act(s_tau)/d s_tau factors out of the tap sum
                const f4 sv = load_texel<F16>(plane, tix << 4);
                acc = f4{acc.x * act_bwd<RACT>(sv.x, act_fwd<RACT>(sv.x)), acc.y * act_bwd<RACT>(sv.y, act_fwd<RACT>(sv.y)),
                         acc.z * act_bwd<RACT>(sv.z, act_fwd<RACT>(sv.z)), acc.w * act_bwd<AACT>(sv.w, act_fwd<AACT>(sv.w))};
            }
            if constexpr (ADAM) {      // the optimiser's step instead of the gradient store (vl3d_render_bwd_adam; t = this workgroup's frame)
                int from;
                if constexpr (CULL) {
                    // tile-culled model: the pre-pass's class byte -- a dynamic texel is stepped here, a static texel's gradient is stored (the step
                    // kernel sums it over the frames), culled texels and texels outside their plane's box are nobody's business
                    const unsigned c = rec.x & 3u;
                    if (c == 2) store_grad_texel<false>(gplane, tix << 4, acc);
                    if (c != 1) return;
                    from = (int)(rec.x >> 2);
                    const size_t offp = (size_t)rec.y << 4;
                    if (pre) adam_texel_step_mv(a, adam_frame_base(a, d, t), offp, from, pre_p, acc, pre_m, pre_v);
                    else adam_texel_step(a, adam_frame_base(a, d, t), offp, from, load_texel<false>(plane, tix << 4), acc);
                } else {
                    unsigned off;
                    if (!adam_texel(a, d, X0 + wx, Y0 + wy, from, off)) return;
                    adam_texel_step(a, adam_frame_base(a, d, t), off, from, load_texel<false>(plane, tix << 4), acc);
                }
            } else if (!VL3D_ABLATE(a.ablate, 2)) store_grad_texel<F16>(gplane, tix << 4, acc);
            if constexpr (MASK) __builtin_nontemporal_store(accm, gmplane + tix);
        };
        auto rec_at = [&](unsigned tix) { if constexpr (ADAM && CULL) return rplane[tix]; else return make_uint2(0u, 0u); };
        if (row < wh && lane < ww) gather(e0, lane, row, win0 + toff_thread, rec0, true);
        // rest of a window larger than 64 x ROWS (stacks stored above the frame's resolution, frame-border tiles, rotations)
        const int nec = ww - RW;                     // columns right of the first 64 (uniform)
        if (nec > 0 && nec <= RW) {
            // right strip of the first ROWS rows, packed: 64 >> sh rows per wave (sh = ceil log2 of its width), so a 6-texel strip
            // of 16 rows is 2 wave passes instead of 16 passes with 6 active lanes each
            const int sh = nec > 1 ? 32 - __builtin_clz((unsigned)(nec - 1)) : 0, rpw = RW >> sh, rmain = min(wh, ROWS);
            const int c = lane & ((1 << sh) - 1), r = lane >> sh;
            for (int wy0 = row * rpw; wy0 < rmain; wy0 += ROWS * rpw) {
                const int wy = wy0 + r, wx = RW + c;
                if (c < nec && wy < rmain) {
                    const unsigned tix = win0 + (unsigned)(wy * a.Ws + wx);
                    gather(oplane[tix], wx, wy, tix, rec_at(tix));
                }
            }
        } else if (nec > RW) {
            for (int wx = lane + RW; wx < ww && row < wh; wx += RW) {
                const unsigned tix = win0 + (unsigned)(row * a.Ws + wx);
                gather(oplane[tix], wx, row, tix, rec_at(tix));
            }
        }
        for (int wy = row + ROWS; wy < wh; wy += ROWS)      // rows below the first ROWS: one wave per row
            for (int wx = lane; wx < ww; wx += RW) {
                const unsigned tix = win0 + (unsigned)(wy * a.Ws + wx);
                gather(oplane[tix], wx, wy, tix, rec_at(tix));
            }
    }
}



// =====================================================================================================
// Backward, two frames per thread (dense stacks, no layer regularisers).  render_bwd_tile_k is VALU-issue bound, and almost
// half of its instruction stream does not depend on the frame: the sweep's homography / divide / base tap / tents / coverage /
// offset, the gather's owner decode and all of its tent weights.  A workgroup of this kernel owns a 30 x 14-pixel tile of
// frames t and t+1 (32 x 16 region, 512 threads, 2 workgroups per CU at <= 128 VGPRs): one set of coordinates and weights,
// two composite states, two staged gradients, two accumulators, two stores.  Per frame the arithmetic is that of
// render_bwd_tile_k in the same order (same bits).  Pre-pass kernels, owner table and window records are shared (region
// width 32 in the owner table's slots).
//
// Region shape: 32 x 16 pixels (30 x 14 owned).  Round 2 measured the flat 64 x 8 alternative in process on one resident stack
// (profiles/ab_inproc.py): the memory pattern alone prefers 1-KiB row segments (profiles/microbench/rw_bw.hip `bwd_like`: 4.51 vs
// 4.18 TB/s of algorithmic bytes), but its halo (x1.38 instead of x1.22 pixels swept per pixel owned) costs more than that:
// 12.58 vs 11.96 ms.  An L2 prefetch of the next plane's tap rows in front of the barrier (4-byte LDS-DMA loads): +0.5 ms.
constexpr int PROWS = 16;      // region rows.  The region width PW is a template parameter (PW * 16 threads): 32 is shipped; 64 -- 992-byte owner
// segments, x1.18 instead of x1.22 pixels swept per pixel owned, but ONE 1024-thread workgroup per CU (80 KB LDS, <= 128 VGPRs) -- was built and
// measured in round 4 (same bits): cfg3 13.3 against 12.6-12.9 ms, reference geometry 16.3 against 15.9, fused schedule 201 against 214 it/s:
// with a single workgroup per CU nothing runs while its 16 waves meet at the per-plane barrier.  Not instantiated.

// gather of one plane for a frame pair: every texel of the tile's window that the owner table assigns to this tile sums its taps
// from the 3 x 3 (or, where pixels are >= 1 texel apart, 2 x 2) staged pixels around its owner pixel -- one set of weights, two
// accumulators, two stores.  sg0 / sg1: staged gradients of frames t and t+1, st: staged texel coordinates (this plane's buffers).
// ADAM: the owner applies the optimiser's step where it would have stored the gradient (vl3d_render_bwd_adam): (d, t0) = the plane and the
// pair's first frame.
template <int ORDER, int RACT, int AACT, bool F16, bool ADAM = false, int PW = 32>
__device__ __forceinline__ void pair_gather_plane(const RenderArgs &a, const float4 *sg0, const float4 *sg1, const float2 *st, int X0, int Y0,
                                                  int ww, int wh, bool apart, unsigned my_tile, unsigned e0, const unsigned short *oplane,
                                                  const char *plane0, char *gplane0, size_t f1, size_t frame_b, bool has1, int col, int row,
                                                  int d = 0, int t0 = 0) {
    const unsigned win0 = (unsigned)(Y0 * a.Ws + X0);
    auto gather = [&](unsigned e, int wx, int wy, unsigned tix) {
        constexpr int SLOT_BITS = PW == 64 ? 10 : 9;      // PW * 16 slots of the region (bwd_owner_table_k)
        if ((e >> SLOT_BITS) != my_tile) return;
        const int lc = (int)(e & ((1u << SLOT_BITS) - 1u));
        const f2 tau = f2{(float)(X0 + wx), (float)(Y0 + wy)};
        f4 acc0 = f4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        if (apart) {
            const f2 c0 = *reinterpret_cast<const f2 *>(&st[lc]);
            const int li0 = lc - (tau.x < c0.x ? 1 : 0) - (tau.y < c0.y ? PW : 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int li = li0 + (k >> 1) * PW + (k & 1);
                const f2 dc = *reinterpret_cast<const f2 *>(&st[li]) - tau;
                const float wgt = tent_weight(dc.x) * tent_weight(dc.y);
                acc0 += *reinterpret_cast<const f4 *>(&sg0[li]) * wgt;
                acc1 += *reinterpret_cast<const f4 *>(&sg1[li]) * wgt;
            }
        } else {
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int li = lc + dy * PW + dx;
                    const f2 dc = *reinterpret_cast<const f2 *>(&st[li]) - tau;
                    const float wgt = tent_weight(dc.x) * tent_weight(dc.y);
                    // a staged pixel whose weight is 0 for EVERY texel of the wave adds exact zeros: its 16-byte LDS reads and multiply-adds are
                    // skipped (uniform branch).  Near unit magnification -- stacks at the frame's resolution, J within a fraction of a percent
                    // of 1 -- the 2x2 test of bwd_windows_k stays undecided, yet five of the nine pixels almost never reach a texel: cfg3
                    // backward -2 % (fp32) / -4 % (fp16 stacks), same bits.  (variant 4 -- the plain 3x3 definition -- does not skip.)
                    if (!a.gather9 && __builtin_amdgcn_ballot_w64(wgt != 0.0f) == 0ull) continue;
                    acc0 += *reinterpret_cast<const f4 *>(&sg0[li]) * wgt;
                    acc1 += *reinterpret_cast<const f4 *>(&sg1[li]) * wgt;
                }
        }
        if constexpr (ORDER == VL3D_ACT_PRE) {
            const f4 sv0 = load_texel<F16>(plane0, tix << 4), sv1 = load_texel<F16>(plane0 + f1, tix << 4);
            acc0 = f4{acc0.x * act_bwd<RACT>(sv0.x, act_fwd<RACT>(sv0.x)), acc0.y * act_bwd<RACT>(sv0.y, act_fwd<RACT>(sv0.y)),
                      acc0.z * act_bwd<RACT>(sv0.z, act_fwd<RACT>(sv0.z)), acc0.w * act_bwd<AACT>(sv0.w, act_fwd<AACT>(sv0.w))};
            acc1 = f4{acc1.x * act_bwd<RACT>(sv1.x, act_fwd<RACT>(sv1.x)), acc1.y * act_bwd<RACT>(sv1.y, act_fwd<RACT>(sv1.y)),
                      acc1.z * act_bwd<RACT>(sv1.z, act_fwd<RACT>(sv1.z)), acc1.w * act_bwd<AACT>(sv1.w, act_fwd<AACT>(sv1.w))};
        }
        if constexpr (ADAM) {
            static_assert(!F16, "the fused optimiser step: fp32 stacks");
            int from;
            unsigned off;
            if (!adam_texel(a, d, X0 + wx, Y0 + wy, from, off)) return;      // outside its plane's box: deferred (its gradient is exactly 0)
            const size_t fbytes = (size_t)a.ad.Hs * a.ad.Ws * 16, fb = ((size_t)d * a.T + t0) * fbytes;      // uniform
            adam_texel_step(a, fb, off, from, load_texel<false>(plane0, tix << 4), acc0);
            if (has1) adam_texel_step(a, fb + fbytes, off, from, load_texel<false>(plane0 + f1, tix << 4), acc1);
        } else {
            store_grad_texel<F16>(gplane0, tix << 4, acc0);
            if (has1) store_grad_texel<F16>(gplane0 + frame_b, tix << 4, acc1);
        }
    };
    if (row < wh && col < ww) gather(e0, col, row, win0 + (unsigned)(row * a.Ws + col));
    // rest of a window larger than 32 x 16: columns beyond 32 as a packed strip, rows beyond 16 one half-wave per row
    const int nec = ww - PW;
    if (nec > 0) {
        const int necp = min(nec, PW);
        const int sh = necp > 1 ? 32 - __builtin_clz((unsigned)(necp - 1)) : 0, rpg = PW >> sh, rmain = min(wh, PROWS);
        const int c = col & ((1 << sh) - 1), r = col >> sh;          // a 32-thread row group takes rpg window rows of the strip
        for (int wxb = PW; wxb < ww; wxb += (1 << sh))
            for (int wy0 = row * rpg; wy0 < rmain; wy0 += PROWS * rpg) {
                const int wy = wy0 + r, wx = wxb + c;
                if (c < necp && wx < ww && wy < rmain) {
                    const unsigned tix = win0 + (unsigned)(wy * a.Ws + wx);
                    gather(oplane[tix], wx, wy, tix);
                }
            }
    }
    for (int wy = row + PROWS; wy < wh; wy += PROWS)
        for (int wx = col; wx < ww; wx += PW) {
            const unsigned tix = win0 + (unsigned)(wy * a.Ws + wx);
            gather(oplane[tix], wx, wy, tix);
        }
}

// composite backward of one plane for one frame: o = activated sample (alpha already 0 where uncovered), pre = the
// value before the activation, extra = the regularisers' part of the gradient w.r.t. the activated layer value (0 without them)
#define VL3D_PAIR_GRAD(o, pre, Gr, Gg, Gb, gA, S, P, Tr, gv, ex)                                                               \
    {                                                                                                                          \
        const float q = dot3p(Gr, o.x, Gg, o.y, Gb, o.z, gA);                                                                  \
        const float w = o.w * Tr;                                                                                              \
        P = fmaf(w, q, P);                                                                                                     \
        const float om = 1.0f - o.w;                                                                                           \
        const float behind = (om > 1e-12f) ? (S - P) * fast_rcp(om) : 0.0f;                                                    \
        gv = make_float4(fmaf(w, Gr, ex.x), fmaf(w, Gg, ex.y), fmaf(w, Gb, ex.z), fmaf(Tr, q, -behind) + ex.w);                \
        Tr *= om;                                                                                                              \
        if constexpr (ORDER == VL3D_ACT_POST)                                                                                  \
            gv = make_float4(gv.x * act_bwd<RACT>(pre.x, o.x), gv.y * act_bwd<RACT>(pre.y, o.y),                              \
                             gv.z * act_bwd<RACT>(pre.z, o.z), gv.w * act_bwd<AACT>(pre.w, o.w));                              \
    }

// REG: with the layer regularisers (MPV.py:511-531: rgb_smooth / a_smooth / sparsity -- what a shipped stage-2 iteration runs,
// configs/mpv_base.txt:33-34): the smoothness gradient is decoded from the sign words the forward stored (reg_grad), the
// sparsity-sum gradient is gN1 + gN2 a_k.  Until round 3 this was a kernel of its own (render_bwd_pair_reg_k: 2-pixel halo, the
// neighbours' layer values through a second LDS stage, sampling pipelined one plane ahead, 128 VGPRs, 18.2 ms at cfg3 against 12.0
// without the regularisers -- VALU bound on re-deriving signs the forward had already formed).
template <int COORD, int BORDER, int ORDER, int RACT, int AACT, bool F16, bool REG = false, bool ADAM = false, int PW = 32>
__global__ __launch_bounds__(PW * PROWS, VL3D_PAIR_MIN_WAVES) void render_bwd_pair_k(RenderArgs a) {      // >= 4 waves per SIMD (2 workgroups per CU at PW = 32): <= 128 VGPRs
    static_assert(PW == 32 || PW == 64, "region width");
    constexpr int PNT = PW * PROWS;
    if (!reinterpret_cast<const int *>(a.plan)[0]) return;
    __shared__ float4 s_g[2][2][PNT];   // [buffer][frame][pixel]
    __shared__ float2 s_t[2][PNT];
    const int tid = threadIdx.x, col = tid & (PW - 1), row = tid / PW;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_x = bid % a.tiles_x, rest = bid / a.tiles_x;
    const int tile_y = rest % a.tiles_y, t0 = (rest / a.tiles_y) * 2;
    const bool has1 = t0 + 1 < a.T;          // odd T: the last pair sweeps frame t0 twice and stores it once
    const int rx0 = tile_x * (PW - 2) - 1, ry0 = tile_y * (PROWS - 2) - 1;
    const int x = rx0 + col, y = ry0 + row;
    const bool inimg = (x >= 0) && (x < a.W) && (y >= 0) && (y < a.H);
    const float px = (float)(a.col0 + x) + a.pc, py = (float)(a.row0 + y) + a.pc;
    constexpr size_t TEXB = F16 ? 8 : 16;
    const size_t frame_b = (size_t)a.Hs * a.Ws * TEXB;
    const size_t plane_stride_b = (size_t)a.T * frame_b;
    const char *plane0 = reinterpret_cast<const char *>(a.stack) + (size_t)t0 * frame_b;
    char *gplane0 = reinterpret_cast<char *>(a.g_stack) + (size_t)t0 * frame_b;
    const size_t f1 = has1 ? frame_b : 0;
    float Gr0 = 0.f, Gg0 = 0.f, Gb0 = 0.f, gA0 = 0.f, S0 = 0.f, Gr1 = 0.f, Gg1 = 0.f, Gb1 = 0.f, gA1 = 0.f, S1 = 0.f;
    if (inimg) {
        size_t pix = ((size_t)t0 * a.H + y) * a.W + x;
        Gr0 = a.g_rgb[pix * 3 + 0]; Gg0 = a.g_rgb[pix * 3 + 1]; Gb0 = a.g_rgb[pix * 3 + 2];
        gA0 = a.g_alpha ? a.g_alpha[pix] : 0.0f;
        S0 = dot3p(Gr0, a.rgb[pix * 3 + 0], Gg0, a.rgb[pix * 3 + 1], Gb0, a.rgb[pix * 3 + 2], gA0 * a.alpha[pix]);
        if (has1) pix += (size_t)a.H * a.W;
        Gr1 = a.g_rgb[pix * 3 + 0]; Gg1 = a.g_rgb[pix * 3 + 1]; Gb1 = a.g_rgb[pix * 3 + 2];
        gA1 = a.g_alpha ? a.g_alpha[pix] : 0.0f;
        S1 = dot3p(Gr1, a.rgb[pix * 3 + 0], Gg1, a.rgb[pix * 3 + 1], Gb1, a.rgb[pix * 3 + 2], gA1 * a.alpha[pix]);
    }
    // layer regularisers: sparsity-sum gradients gN1 + gN2 a_k, smoothness coefficients, this pixel's flags and sign words
    float gN10 = 0.f, gN20 = 0.f, gN11 = 0.f, gN21 = 0.f;
    f4 gx = f4{0.f, 0.f, 0.f, 0.f}, gy = gx;
    unsigned fl = 0u;
    const uint2 *sgp = nullptr;
    uint2 go0 = make_uint2(0u, 0u), gl0 = go0, gu0 = go0, gp0 = go0, go1 = go0, gl1 = go0, gu1 = go0, gp1 = go0;   // the group's sign words (planes 0, 1 | 2, 3)
    const bool has_l = inimg && x >= 1, has_u = inimg && y >= 1;
    const bool reg_on = REG && a.g_reg != nullptr;
    const size_t sg_plane = (size_t)a.T * a.H * a.W, sg_f1 = has1 ? (size_t)a.H * a.W : 0;
    if constexpr (REG) {
        if (inimg && a.g_asum) {
            size_t pix = ((size_t)t0 * a.H + y) * a.W + x;
            gN10 = a.g_asum[pix * 2 + 0]; gN20 = 2.0f * a.g_asum[pix * 2 + 1];
            pix += sg_f1;
            gN11 = a.g_asum[pix * 2 + 0]; gN21 = 2.0f * a.g_asum[pix * 2 + 1];
        }
        if (reg_on) {
            gx = f4{a.g_reg[0], a.g_reg[0], a.g_reg[0], a.g_reg[2]}; gy = f4{a.g_reg[1], a.g_reg[1], a.g_reg[1], a.g_reg[3]};
            const int xc = min(max(x, 0), a.W - 1), yc = min(max(y, 0), a.H - 1);
            sgp = reinterpret_cast<const uint2 *>(a.reg_signs) + ((size_t)t0 * a.H + yc) * a.W + xc;      // clamped into the frame: the loads below are unconditional
            if (inimg) fl = a.reg_flags[(size_t)y * a.W + x];
        }
    }
    float Tr0 = 1.0f, P0 = 0.0f, Tr1 = 1.0f, P1 = 0.0f;
    const TapStep st = make_tap_step<F16>(a.Hs, a.Ws);
    const unsigned my_tile_id = (unsigned)(tile_y * a.tiles_x + tile_x);
    // 9-bit slots (PW = 32): 4 + 3 bits of tile code; 10-bit slots (PW = 64): 3 + 3 (bwd_owner_table_k)
    const unsigned my_tile = (unsigned)((tile_y & (PW == 64 ? 7 : 15)) << 3 | (tile_x & 7));
    const unsigned toff_thread = (unsigned)(row * a.Ws + col);
    const cint_p wrec = (cint_p)a.plan + plan_win_off(a.D) + (size_t)my_tile_id * a.D * 4;
    typedef typename TapVal<F16, ORDER>::type tapv_t;
    for (int d = 0; d < a.D; ++d, plane0 += plane_stride_b, gplane0 += plane_stride_b) {
        float h[VL3D_HN];
        load_uniform(a.homos + VL3D_HS * d, h);
        const int X0 = wrec[4 * d], Y0 = wrec[4 * d + 1], wwh = wrec[4 * d + 2];
        const int ww = wwh & 0xffff, wh = (wwh >> 16) & 0x3fff;
        const bool apart = (wwh & 0x40000000) != 0;
        const int buf = d & 1;
        const unsigned short *oplane = a.owner + (size_t)d * a.Hs * a.Ws;
        const unsigned e0 = oplane[(unsigned)(Y0 * a.Ws + X0) + toff_thread];     // unconditional (padded table), arrives in the shadow of the sweep
        // sign words of this group of four planes (own, left, upper; two frames): requested with the taps of its first plane
        if constexpr (REG) if (reg_on && (d & 3) == 0 && !VL3D_REGAB(16)) {      // uniform
            const uint2 *w = sgp;      // (advanced by one group of planes below: no 64-bit multiply per group)
            sgp += sg_plane;
            const ptrdiff_t ol = has_l ? -1 : 0, ou = has_u ? -(ptrdiff_t)a.W : 0;
            const uint2 zz = make_uint2(0x55555555u, 0x55555555u);      // a neighbour that does not exist: sign 0 everywhere
            go0 = w[0]; gl0 = w[ol]; gu0 = w[ou];
            go1 = w[sg_f1]; gl1 = w[sg_f1 + ol]; gu1 = w[sg_f1 + ou];
            if (!has_l) { gl0 = zz; gl1 = zz; }
            if (!has_u) { gu0 = zz; gu1 = zz; }
            if (fl & 12u) {
                const uint2 *pw = reinterpret_cast<const uint2 *>(a.reg_patch) + (w - reinterpret_cast<const uint2 *>(a.reg_signs));
                gp0 = pw[0]; gp1 = pw[sg_f1];
            }
        }
        // (2) sweep: one set of taps, two frames
        float2 tc = make_float2(0.f, 0.f);
        float4 gv0 = make_float4(0.f, 0.f, 0.f, 0.f), gv1 = gv0;
        if (inimg) {
            const Taps2 tp = make_taps2<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy);
            tapv_t tv0[4], tv1[4];
            load_taps2<F16>(plane0, tp, st, tv0);
            load_taps2<F16>(plane0 + f1, tp, st, tv1);
            f4 pre0, pre1;
            const f4 o0 = shade2<ORDER, RACT, AACT>(tp, tv0, &pre0);
            const f4 o1 = shade2<ORDER, RACT, AACT>(tp, tv1, &pre1);
            f4 ex0 = f4{0.f, 0.f, 0.f, 0.f}, ex1 = ex0;
            if constexpr (REG) {
                if (reg_on && !VL3D_REGAB(8)) {
                    const bool hi = (d & 2) != 0;       // uniform
                    const unsigned bo = 16u * (d & 1);
                    ex0 = reg_grad32(hi ? go0.y : go0.x, hi ? gl0.y : gl0.x, hi ? gu0.y : gu0.x, hi ? gp0.y : gp0.x, bo, fl, gx, gy);
                    ex1 = reg_grad32(hi ? go1.y : go1.x, hi ? gl1.y : gl1.x, hi ? gu1.y : gu1.x, hi ? gp1.y : gp1.x, bo, fl, gx, gy);
                }
                ex0.w += fmaf(gN20, o0.w, gN10);
                ex1.w += fmaf(gN21, o1.w, gN11);
            }
            VL3D_PAIR_GRAD(o0, pre0, Gr0, Gg0, Gb0, gA0, S0, P0, Tr0, gv0, ex0)
            VL3D_PAIR_GRAD(o1, pre1, Gr1, Gg1, Gb1, gA1, S1, P1, Tr1, gv1, ex1)
            tc = make_float2(tp.tx, tp.ty);
            if (!(tp.cov > 0.0f)) { gv0 = make_float4(0.f, 0.f, 0.f, 0.f); gv1 = gv0; }
        }
        s_t[buf][tid] = tc;
        s_g[buf][0][tid] = gv0;
        s_g[buf][1][tid] = gv1;
        __syncthreads();
        // (3) gather: one set of weights, two accumulators
        pair_gather_plane<ORDER, RACT, AACT, F16, ADAM, PW>(a, s_g[buf][0], s_g[buf][1], s_t[buf], X0, Y0, ww, wh, apart, my_tile, e0, oplane, plane0,
                                                        gplane0, f1, frame_b, has1, col, row, d, t0);
    }
}

#undef VL3D_PAIR_GRAD

template <int COORD, int BORDER, int ORDER, int RACT, int AACT, bool F16, bool REG = false, bool ADAM = false, int PW = 32>
void launch_pair(const RenderArgs &a, hipStream_t s) {
    constexpr int RH = 1, IW = PW - 2 * RH, IH = PROWS - 2 * RH;
    RenderArgs b = a;
    b.tiles_x = (a.W + IW - 1) / IW; b.tiles_y = (a.H + IH - 1) / IH;
    const int nwin = b.tiles_x * b.tiles_y * a.D;
    hipLaunchKernelGGL((bwd_windows_k<COORD>), dim3((nwin + 255) / 256), dim3(256), 0, s, b, IW, IH, RH, b.tiles_x, b.tiles_y,
                       reinterpret_cast<int *>(const_cast<float *>(a.plan)) + plan_win_off(a.D));
    hipLaunchKernelGGL(bwd_owner_table_k, dim3((a.Ws + 63) / 64, (a.Hs + 3) / 4, a.D), dim3(256), 0, s, b, IW, IH, RH, b.tiles_x,
                       const_cast<unsigned short *>(a.owner), PW, PW == 64 ? 10 : 9);
    hipLaunchKernelGGL((render_bwd_pair_k<COORD, BORDER, ORDER, RACT, AACT, F16, REG, ADAM, PW>),
                       dim3((unsigned)(b.tiles_x * b.tiles_y * ((a.T + 1) / 2))), dim3(PW * PROWS), 0, s, b);
}

// =====================================================================================================
// Forward WITH the layer regularisers in one pass: the render (rgb, alpha, alpha sums) and the four smoothness sums
// (MPV.py:517-531) from ONE sweep over the stack.  render_fwd2(x)_k + a separate sums kernel read every texel twice (5.1 + 7.1 ms at
// cfg3 for what a shipped stage-2 iteration calls once per step); here a workgroup is a 64 x 8-pixel region whose last column and
// row are halo (63 x 7 pixels owned: their outputs and the |o - o_right|, |o - o_down| pairs), each plane's activated layer values
// go through a double-buffered LDS tile (one barrier per plane), and the taps of plane d+1 are in flight across that barrier (two
// register sets, as in render_fwd2_k).  Per pixel the composite is render_fwd2_k's, instruction for instruction (same bits).
template <int COORD, int BORDER, int ORDER, int RACT, int AACT, bool F16, bool MASK = false>
__global__ __launch_bounds__(512) void render_fwd_reg_k(RenderArgs a, int tiles_x, int tiles_y) {
    constexpr int FW = 64, FH = 8, NT = FW * FH;
    __shared__ float4 s_o[2][NT];
    __shared__ float red[4][FH];
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_x = b % tiles_x, rest = b / tiles_x;
    const int tile_y = rest % tiles_y, t = rest / tiles_y;
    const int tid = threadIdx.x, lane = tid & 63, row = tid >> 6;
    const int x = tile_x * (FW - 1) + lane, y = tile_y * (FH - 1) + row;
    const bool inimg = (x < a.W) && (y < a.H);
    const bool owner = inimg && lane < FW - 1 && row < FH - 1;        // the last column / row are owned by the next tile (its column / row 0)
    // pairs whose two pixels are covered by different planes are not formed here (reg_slot_fwd_k takes them, slot by slot)
    const unsigned fl = owner ? a.reg_flags[(size_t)y * a.W + x] : 0u;
    const bool own_r = owner && x + 1 < a.W && !(fl & 1u), own_d = owner && y + 1 < a.H && !(fl & 2u);
    const size_t sg_plane = (size_t)a.T * a.H * a.W;
    uint2 *sgq = reinterpret_cast<uint2 *>(a.reg_signs) + ((size_t)t * a.H + min(y, a.H - 1)) * a.W + min(x, a.W - 1);
    unsigned sg_lo = 0u, sg_hi = 0u;      // sign words of the current group of four planes (planes 0, 1 | 2, 3)
    // pixels outside the frame sample the frame's last pixel (valid addresses, results dropped): no branch around the loads
    const float px = (float)(a.col0 + min(x, a.W - 1)) + a.pc, py = (float)(a.row0 + min(y, a.H - 1)) + a.pc;
    const size_t frame_b = (size_t)a.Hs * a.Ws * (F16 ? 8 : 16), plane_stride_b = (size_t)a.T * frame_b;
    const char *plane = reinterpret_cast<const char *>(a.stack) + (size_t)t * frame_b;
    float Tr = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, A = 0.f, n1 = 0.f, n2 = 0.f;
    float sxc = 0.f, syc = 0.f, sxa = 0.f, sya = 0.f;
    const TapStep st = make_tap_step<F16>(a.Hs, a.Ws);
    typedef typename TapVal<F16, ORDER>::type tapv_t;
    tapv_t vA[4], vB[4];
    static_assert(!(MASK && F16), "the loop-mask channel: fp32 stage-1 stacks");
    float mA[4] = {0.f, 0.f, 0.f, 0.f}, mB[4] = {0.f, 0.f, 0.f, 0.f}, lab = 0.f;      // MASK: the loop-mask texture's taps, the composited label
    const float *mplane = MASK ? a.mask + (size_t)t * a.Hs * a.Ws : nullptr;
    const size_t mplane_stride = (size_t)a.T * a.Hs * a.Ws;
#define VL3D_PLANE(T_, V_, BUF_, S_, M_)                                                                  \
    {                                                                                                     \
        const f4 o = shade2<ORDER, RACT, AACT>(T_, V_);                                                   \
        const f4 ol = inimg ? o * T_.cov : f4{0.f, 0.f, 0.f, 0.f};                                        \
        if (!VL3D_REGAB(4)) s_o[BUF_][tid] = make_float4(ol.x, ol.y, ol.z, ol.w);                         \
        const float w = o.w * Tr;                                                                         \
        cr += w * o.x; cg += w * o.y; cb += w * o.z; A += w;                                              \
        n1 += o.w; n2 = fmaf(o.w, o.w, n2);                                                               \
        Tr *= (1.0f - o.w);                                                                               \
        if constexpr (MASK) lab = fmaf(w, act_fwd<VL3D_ACT_SIGMOID>(mask_blend(M_, T_.w)), lab);          \
        if (!VL3D_REGAB(4)) __syncthreads();                                                              \
        int code = (int)REG_ZERO;                                                                         \
        if (own_r && !VL3D_REGAB(2)) {                                                                    \
            const float4 r = s_o[BUF_][tid + 1];                                                          \
            const f4 df = ol - f4{r.x, r.y, r.z, r.w};                                                    \
            sxc += fabsf(df.x) + fabsf(df.y) + fabsf(df.z);                                               \
            sxa += fabsf(df.w);                                                                           \
            if (!VL3D_REGAB(1)) code += reg_signs4(df);                                                   \
        }                                                                                                 \
        if (own_d && !VL3D_REGAB(2)) {                                                                    \
            const float4 r = s_o[BUF_][tid + FW];                                                         \
            const f4 df = ol - f4{r.x, r.y, r.z, r.w};                                                    \
            syc += fabsf(df.x) + fabsf(df.y) + fabsf(df.z);                                               \
            sya += fabsf(df.w);                                                                           \
            if (!VL3D_REGAB(1)) code += reg_signs4(df) << 8;                                              \
        }                                                                                                 \
        if (VL3D_REGAB(1)) { }                                                                            \
        else if (S_ == 0) sg_lo = (unsigned)code;                                                         \
        else if (S_ == 1) sg_lo |= (unsigned)code << 16;                                                  \
        else if (S_ == 2) sg_hi = (unsigned)code;                                                         \
        else {                                                                                            \
            sg_hi |= (unsigned)code << 16;                                                                \
            if (owner) *sgq = make_uint2(sg_lo, sg_hi);                                                   \
            sgq += sg_plane;                                                                              \
        }                                                                                                 \
    }
    const QuadCull noq = QuadCull{nullptr, 0, 0, 0.f, 0.f, 0.f, 0.f, 0};
    Taps2 tA = make_taps2<COORD, BORDER>(a.homos, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, noq, UvNoise{a.uv_seed, 0}), tB = tA;
    load_taps2<F16>(plane, tA, st, vA);
    if constexpr (MASK) load_mask_taps(mplane, tA.off, st, mA);
#define VL3D_FETCH(T_, V_, DN_, M_)                                                                       \
    {                                                                                                     \
        const int dn = min(DN_, a.D - 1);                                                                 \
        float h[VL3D_HN];                                                                                 \
        load_uniform(a.homos + VL3D_HS * dn, h);                                                          \
        T_ = make_taps2<COORD, BORDER>(h, px, py, a.Hs, a.Ws, a.sx, a.sy, a.ox, a.oy, noq, UvNoise{a.uv_seed, dn});  \
        load_taps2<F16>(plane + (size_t)dn * plane_stride_b, T_, st, V_);                                 \
        if constexpr (MASK) load_mask_taps(mplane + (size_t)dn * mplane_stride, T_.off, st, M_);          \
        asm volatile("" ::: "memory");                                                                    \
    }
    // four planes per trip: the slot of a plane inside its group of sign words is a compile-time constant
    int d = 0;
    for (;; d += 4) {
        VL3D_FETCH(tB, vB, d + 1, mB)
        VL3D_PLANE(tA, vA, 0, 0, mA)
        if (d + 1 >= a.D) break;
        VL3D_FETCH(tA, vA, d + 2, mA)
        VL3D_PLANE(tB, vB, 1, 1, mB)
        if (d + 2 >= a.D) break;
        VL3D_FETCH(tB, vB, d + 3, mB)
        VL3D_PLANE(tA, vA, 0, 2, mA)
        if (d + 3 >= a.D) break;
        VL3D_FETCH(tA, vA, d + 4, mA)
        VL3D_PLANE(tB, vB, 1, 3, mB)
        if (d + 4 >= a.D) break;
    }
    if ((a.D & 3) && owner) *sgq = make_uint2(sg_lo, (a.D & 3) == 3 ? sg_hi : 0u);      // the last, partial group
#undef VL3D_FETCH
#undef VL3D_PLANE
    if (owner) {
        const size_t pix = ((size_t)t * a.H + y) * a.W + x;
        a.rgb[pix * 3 + 0] = cr; a.rgb[pix * 3 + 1] = cg; a.rgb[pix * 3 + 2] = cb;
        a.alpha[pix] = A;
        if (a.asum) { a.asum[pix * 2 + 0] = n1; a.asum[pix * 2 + 1] = n2; }
        if constexpr (MASK) a.label[pix] = lab;
    }
    float v[4] = {sxc, syc, sxa, sya};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
        if (lane == 0) red[k][row] = v[k];
    }
    __syncthreads();
    if (tid < 4) {
        double sum = 0.0;
        for (int r = 0; r < FH; ++r) sum += (double)red[tid][r];
        atomicAdd(a.reg_sums + tid, sum);
    }
}

// ---- launch templates ---------------------------------------------------------------------------------------------
template <int COORD, int BORDER, int ORDER, int RACT, int AACT, int ROWS, bool REG, bool F16 = false, bool MASK = false, bool ADAM = false, int RWT = 64>
void launch_tile(const RenderArgs &a, hipStream_t s) {
    constexpr int RH = 1, IW = RWT - 2 * RH, IH = ROWS - 2 * RH;
    constexpr int SLOT_BITS = RWT * ROWS > 512 ? 10 : 9;
    RenderArgs b = a;
    b.tiles_x = (a.W + IW - 1) / IW; b.tiles_y = (a.H + IH - 1) / IH;
    const int nwin = b.tiles_x * b.tiles_y * a.D;
    hipLaunchKernelGGL((bwd_windows_k<COORD>), dim3((nwin + 255) / 256), dim3(256), 0, s, b, IW, IH, RH, b.tiles_x, b.tiles_y,
                       reinterpret_cast<int *>(const_cast<float *>(a.plan)) + plan_win_off(a.D));
    if (a.T == 1 && !a.ad.p && a.owner4)      // (a single frame: four texels per thread; the fused optimiser step keeps its per-texel records)
        hipLaunchKernelGGL(bwd_owner_table4_k, dim3(((a.Ws + 3) / 4 + 63) / 64, (a.Hs + 3) / 4, a.D), dim3(256), 0, s, b, IW, IH, RH,
                           const_cast<unsigned short *>(a.owner), RWT, SLOT_BITS);
    else
        hipLaunchKernelGGL(bwd_owner_table_k, dim3((a.Ws + 63) / 64, (a.Hs + 3) / 4, a.D), dim3(256), 0, s, b, IW, IH, RH, b.tiles_x,
                           const_cast<unsigned short *>(a.owner), RWT, SLOT_BITS);
    const dim3 grid((unsigned)(b.tiles_x * b.tiles_y * a.T)), block(RWT * ROWS);
    if constexpr (ADAM) {      // (tile-culled models only: the dense fused step rides the frame pairs -- the one-frame form measured 202 against 213-218 it/s)
        hipLaunchKernelGGL((render_bwd_tile_k<COORD, BORDER, ORDER, RACT, AACT, ROWS, true, false, true, false, true, RWT>), grid, block, 0, s, b);
    } else if constexpr (MASK) {       // (dense models only: the entry point refuses a quad map)
        hipLaunchKernelGGL((render_bwd_tile_k<COORD, BORDER, ORDER, RACT, AACT, ROWS, REG, F16, false, true, false, RWT>), grid, block, 0, s, b);
    } else if (a.quad_keep)
        hipLaunchKernelGGL((render_bwd_tile_k<COORD, BORDER, ORDER, RACT, AACT, ROWS, REG, F16, true, false, false, RWT>), grid, block, 0, s, b);
    else
        hipLaunchKernelGGL((render_bwd_tile_k<COORD, BORDER, ORDER, RACT, AACT, ROWS, REG, F16, false, false, false, RWT>), grid, block, 0, s, b);
}

template <bool BWD, int COORD, int BORDER, int ORDER, int RACT, int AACT, bool F16>
void launch_t(const RenderArgs &a, hipStream_t s) {
    dim3 grid((a.W + TILE_X - 1) / TILE_X, (a.H + TILE_Y - 1) / TILE_Y, a.T), block(TILE_X * TILE_Y);
    // the loop-mask channel (a.mask != NULL) is built for the convention stage 1 ships (MPI.py planar path, sigmoid / sigmoid, fp32); the
    // entry points refuse every other descriptor
    constexpr bool MASKABLE = COORD == VL3D_COORD_AFFINE && BORDER == VL3D_BORDER_HARDCUT && ORDER == VL3D_ACT_POST &&
                              RACT == VL3D_ACT_SIGMOID && AACT == VL3D_ACT_SIGMOID && !F16;
    if constexpr (BWD) {
        if (a.tile_rows) {
            hipLaunchKernelGGL((bwd_plan_k<COORD>), dim3(1), dim3(64), 0, s, a, 16, const_cast<float *>(a.plan));
            const size_t n8 = (size_t)a.D * a.T * a.Hs * a.Ws * (a.g_f16 ? 1 : 2);          // fp16 texels are 8 bytes, fp32 ones 16
            hipLaunchKernelGGL(bwd_fill_zero_if_infeasible_k, dim3(4096), dim3(256), 0, s, reinterpret_cast<float2 *>(a.g_stack), n8, a.plan);
            bool done = false;
            if constexpr (MASKABLE) {
                if (a.mask) {      // one frame per thread, fifth channel in the sweep and the gather
                    hipLaunchKernelGGL(bwd_fill_zero_f32_if_infeasible_k, dim3(1024), dim3(256), 0, s, a.g_mask, (size_t)a.D * a.T * a.Hs * a.Ws, a.plan);
                    // (flat 64 x 8 regions, tile_rows 8: 512 threads at this instantiation's 128-register budget are TWO workgroups per CU -- the
                    // 1024-thread regions run alone on theirs; variant 3 keeps the 16 rows)
                    if (a.tile_rows == 8) {
                        if (a.g_reg || a.g_asum) launch_tile<COORD, BORDER, ORDER, RACT, AACT, 8, true, false, true>(a, s);
                        else launch_tile<COORD, BORDER, ORDER, RACT, AACT, 8, false, false, true>(a, s);
                    } else if (a.g_reg || a.g_asum) launch_tile<COORD, BORDER, ORDER, RACT, AACT, 16, true, false, true>(a, s);
                    else launch_tile<COORD, BORDER, ORDER, RACT, AACT, 16, false, false, true>(a, s);
                    hipLaunchKernelGGL((render_bwd_k<COORD, BORDER, ORDER, RACT, AACT, false, true>), grid, block, 0, s, a);
                    return;
                }
            }
            // the optimiser step in the owner store (vl3d_render_bwd_adam; the entry point admits this convention only): always the frame pairs
            if constexpr (COORD == VL3D_COORD_AFFINE && BORDER == VL3D_BORDER_HARDCUT && ORDER == VL3D_ACT_POST && RACT == VL3D_ACT_SIGMOID &&
                          AACT == VL3D_ACT_SIGMOID && !F16 && VL3D_HS == 9) {
                if (a.ad.p) {
                    // tile-culled models: one frame per thread, 64-wide regions (the frame pairs are built for dense stacks)
                    // (32-wide regions unless variant 3 asks for the 64-wide ones: two workgroups per CU, 0.60 against 0.71 ms per iteration of the
                    // tile-culled schedule, docs/kernels/K2_render_backward.md round 5)
                    if (a.quad_keep) {
                        if (a.tile_rows != 16) launch_tile<COORD, BORDER, ORDER, RACT, AACT, 16, true, false, false, true, 32>(a, s);
                        else launch_tile<COORD, BORDER, ORDER, RACT, AACT, 16, true, false, false, true>(a, s);
                    }
                    else if (a.g_reg || a.g_asum) launch_pair<COORD, BORDER, ORDER, RACT, AACT, false, true, true>(a, s);
                    else launch_pair<COORD, BORDER, ORDER, RACT, AACT, false, false, true>(a, s);
                    hipLaunchKernelGGL((render_bwd_k<COORD, BORDER, ORDER, RACT, AACT, F16>), grid, block, 0, s, a);
                    return;
                }
            }
            if constexpr (RACT == VL3D_ACT_SIGMOID && AACT == VL3D_ACT_SIGMOID) {
                // two frames per thread: dense stacks without layer regularisers (tile_rows 17 = "16 rows, pairs allowed"), when a
                // 30 x 14-pixel tile's texel window fits the 32 x 16 threads of its workgroup -- judged by the sizes alone (the
                // homographies live on the device): along one axis at least the stack is no larger than the frame (+7 %) -- full frames
                // and row bands (dist.render_band: full width, rows = band + halo) of a stack at the frame's resolution.  Beyond that
                // the extra gather passes of the small tiles cost more than the pairs save (1.1x: 13.3 ms tile kernel, 13.9 ms
                // pairs): crops of a larger stack and the reference's 1.1x stacks keep the 64 x 16 tile kernel.
                const bool fits = (int64_t)a.Hs * 100 <= (int64_t)a.H * 107 || (int64_t)a.Ws * 100 <= (int64_t)a.W * 107;
                if (a.tile_rows == 17 && a.T >= 2 && !a.g_reg && !a.g_asum && !a.quad_keep && fits) {
                    launch_pair<COORD, BORDER, ORDER, RACT, AACT, F16>(a, s);
                    done = true;
                }
                // with the layer regularisers the pair kernel wins at every stack size (round 3, in process: 15.7 ms against 20.0 ms for
                // the one-frame tile kernel on a 1.1x stack, 15.4 against 19.6 ms at the frame's resolution: decoding the forward's sign
                // words is frame-pair work the tile kernel does once per frame).  The utils_mpi cross-check convention keeps the tile
                // kernel: its texel coordinates cost a reciprocal more and the pair instantiation spilled 8-28 bytes at 128 VGPRs.
                if constexpr (COORD != VL3D_COORD_UTILS_MPI) {
                    if (!done && a.tile_rows == 17 && a.T >= 2 && (a.g_reg || a.g_asum) && !a.quad_keep) {
                        launch_pair<COORD, BORDER, ORDER, RACT, AACT, F16, true>(a, s);
                        done = true;
                    }
                }
            }
            // 32-wide one-frame regions (variant 5; the shipped planar convention only): half the workgroup, twice as many of them
            if constexpr (MASKABLE && VL3D_HS == 9) {
                // (... and the default of a tile-culled call under VL3D_GRAD_CULLED_UNWRITTEN, which SKIPS the planes a tile cannot see: cfg3 at
                // 16.5 % kept quads 3.7 ms at the plain kernel's register budget (70 VGPRs, three workgroups per CU), 4.5-4.8 ms in the
                // instantiation with the regularisers' 128 (variant 5), 5.1 ms in its 64-wide form, profiles/r05b_cull_lean.txt.  A per-tile work list of the swept planes -- bit masks written by
                // bwd_windows_k, scalar bit scans instead of one record load per skipped plane -- measured the same 4.49 ms / 0.60 ms per
                // schedule iteration: the skipped planes' scalar loads are hidden, not built)
                if (!done && (a.tile_rows == 18 || (a.tile_rows == 17 && a.quad_keep && a.grad_culled_unwritten))) {
                    if (a.g_reg || a.g_asum || (a.tile_rows == 18 && a.quad_keep && a.grad_culled_unwritten)) launch_tile<COORD, BORDER, ORDER, RACT, AACT, 16, true, false, false, false, 32>(a, s);
                    else launch_tile<COORD, BORDER, ORDER, RACT, AACT, 16, false, false, false, false, 32>(a, s);
                    done = true;
                }
            }
            // flat 64 x 8 one-frame regions (512 threads, four workgroups per CU at the plain kernel's 64 registers) for a SINGLE frame (cfg2, the
            // stage-1 shape): 2520 workgroups on 1024 slots instead of 1092 on 512 -- the x1.33 halo costs less than the 2.13-round tail of the
            // 16-row regions: backward 0.380 against 0.394 ms at 720p, D = 32, same bits (profiles/r05d_cfg2_rows.txt; 10 rows 0.442, 12 rows
            // 0.393: measured, not instantiated).  Variant 2 forces them at any T, variant 3 keeps the 16 rows.
            if constexpr (MASKABLE && VL3D_HS == 9) {
                if (!done && !a.g_reg && !a.g_asum && !a.quad_keep && (a.tile_rows == 8 || (a.tile_rows == 17 && a.T == 1))) {
                    launch_tile<COORD, BORDER, ORDER, RACT, AACT, 8, false, false>(a, s);
                    done = true;
                }
            }
            if (!done) {
                // layer regularisers and / or sparsity sums: the REG instantiation (128-VGPR budget); a tile-culled call whose consumer never
                // reads culled texels takes it too -- it is the one that SKIPS the planes a tile cannot see instead of zero-filling them
                if (a.g_reg || a.g_asum || (a.quad_keep && a.grad_culled_unwritten)) {
                    launch_tile<COORD, BORDER, ORDER, RACT, AACT, 16, true, F16>(a, s);
                } else {
                    launch_tile<COORD, BORDER, ORDER, RACT, AACT, 16, false, F16>(a, s);
                }
            }
        }
        if constexpr (MASKABLE) {
            if (a.mask) {
                hipLaunchKernelGGL((render_bwd_k<COORD, BORDER, ORDER, RACT, AACT, false, true>), grid, block, 0, s, a);
                return;
            }
        }
        hipLaunchKernelGGL((render_bwd_k<COORD, BORDER, ORDER, RACT, AACT, F16>), grid, block, 0, s, a);
    } else {
        // with the regularisers: coverage masks + pair flags (frame independent), the plane-by-plane kernel over the regular pairs,
        // then the slot-by-slot kernel over the irregular ones
        if (a.reg_fwd == 2) {       // render + regulariser sums in one pass (dense stacks)
            launch_reg_prepass<COORD, BORDER, ORDER, RACT, AACT, F16>(a, s);
            const int tx = (a.W + 62) / 63, ty = (a.H + 6) / 7;
            bool with_mask = false;
            if constexpr (MASKABLE) {
                if (a.mask) {
                    hipLaunchKernelGGL((render_fwd_reg_k<COORD, BORDER, ORDER, RACT, AACT, false, true>), dim3((unsigned)(tx * ty * a.T)), dim3(512), 0, s, a, tx, ty);
                    with_mask = true;
                }
            }
            if (!with_mask)
                hipLaunchKernelGGL((render_fwd_reg_k<COORD, BORDER, ORDER, RACT, AACT, F16>), dim3((unsigned)(tx * ty * a.T)), dim3(512), 0, s, a, tx, ty);
            launch_reg_slots<COORD, BORDER, ORDER, RACT, AACT, F16, true>(a, s);
            return;
        }
        if (a.reg_fwd == 3) {       // render + regulariser sums of a tile-culled model in one pass: the slot kernel composites as it goes
            launch_reg_prepass<COORD, BORDER, ORDER, RACT, AACT, F16>(a, s);
            launch_reg_slots<COORD, BORDER, ORDER, RACT, AACT, F16, false, true>(a, s);
            return;
        }
        if (a.reg_fwd) {        // the sums alone (tile-culled models; the two-pass forward of dense ones): every pair slot by slot
            launch_reg_prepass<COORD, BORDER, ORDER, RACT, AACT, F16>(a, s);
            launch_reg_slots<COORD, BORDER, ORDER, RACT, AACT, F16, false>(a, s);
            return;
        }
        // frame pairs (shipped activations, dense stacks, T >= 2); forward variant 6 (desc->variant bits 8..11) keeps the one-frame
        // kernel (A/B, bitwise tests).  The workgroup-shape variants of round 1 (64x4, 64x16, no XCD remap) measured within the
        // noise of the default and are no longer built.
        if constexpr (MASKABLE) {
            if (a.mask) {       // one frame per thread with the fifth channel
                const int tiles_x = (a.W + 63) / 64, tiles_y = (a.H + 7) / 8;
                hipLaunchKernelGGL((render_fwd2_k<COORD, BORDER, ORDER, RACT, AACT, 8, true, false, false, true>), dim3((unsigned)(tiles_x * tiles_y * a.T)),
                                   dim3(64 * 8), 0, s, a, tiles_x, tiles_y);
                return;
            }
        }
        if constexpr (RACT == VL3D_ACT_SIGMOID && AACT == VL3D_ACT_SIGMOID) {
            // (tile-culled models too; add_uv_noise: the jitter lives in the one-frame kernel)
            if (a.T >= 2 && a.fwd_variant != 6 && !a.uv_seed) return launch_fwd2x<COORD, BORDER, ORDER, RACT, AACT, F16>(a, s);
        }
        launch_fwd2<COORD, BORDER, ORDER, RACT, AACT, 8, true, F16>(a, s);
    }
}

// fp16 plane stacks (cfg5) are instantiated for the shipped (sigmoid, sigmoid) activations only
template <bool BWD, int COORD, int BORDER, int ORDER, int RACT, int AACT>
void launch(const RenderArgs &a, hipStream_t s) {
    if constexpr (RACT == VL3D_ACT_SIGMOID && AACT == VL3D_ACT_SIGMOID) {
        if (a.g_f16) return launch_t<BWD, COORD, BORDER, ORDER, RACT, AACT, true>(a, s);
    }
    launch_t<BWD, COORD, BORDER, ORDER, RACT, AACT, false>(a, s);
}

}  // namespace
