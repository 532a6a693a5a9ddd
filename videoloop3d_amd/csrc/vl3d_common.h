// Shared device/host helpers for the vl3d HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "vl3d.h"

extern "C" void vl3d_set_error(const char *msg);

#define VL3D_REQUIRE(cond, msg)          \
    do {                                 \
        if (!(cond)) {                   \
            vl3d_set_error(msg);         \
            return VL3D_EINVAL;          \
        }                                \
    } while (0)

#define VL3D_CHECK_LAUNCH()                                  \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) {                             \
            vl3d_set_error(hipGetErrorString(e__));          \
            return VL3D_ELAUNCH;                             \
        }                                                    \
    } while (0)

#define VL3D_HIP(call)                                       \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) {                             \
            vl3d_set_error(hipGetErrorString(e__));          \
            return VL3D_ELAUNCH;                             \
        }                                                    \
    } while (0)

// Timing-only ablation switches (skip the gather, the stores, the tap loads ...: WRONG RESULTS, used to price the parts of a kernel) exist
// only in a measurement build (-DVL3D_VARIANTS, profiles/build_variant.sh).  In the product they are compile-time false and the ABI
// refuses desc->variant bits 4-7 (vl3d_check_variant).
#ifdef VL3D_VARIANTS
#define VL3D_ABLATE(word, bit) (((word) & (bit)) != 0)
#else
#define VL3D_ABLATE(word, bit) (false)
#endif
static inline int vl3d_check_variant(int variant) {
#ifndef VL3D_VARIANTS
    if (variant & 0xf0) {
        vl3d_set_error("desc->variant bits 4-7 select timing-only ablations (wrong results): they exist only in a -DVL3D_VARIANTS measurement build");
        return VL3D_EINVAL;
    }
#endif
    (void)variant;
    return VL3D_OK;
}

// ---- activations (MPI.py:21-31) -----------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ float act_fwd(float v) {
    if constexpr (ACT == VL3D_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __expf(-v));   // 4 instrs, ~1 ulp
    else if constexpr (ACT == VL3D_ACT_RELU) return fmaxf(v, 0.0f);
    else if constexpr (ACT == VL3D_ACT_CLAMP) return fminf(fmaxf(v, 0.0f), 1.0f);
    else if constexpr (ACT == VL3D_ACT_ABS) return fabsf(v);
    else return v;
}

// derivative given pre-activation input `v` and activated output `o`
template <int ACT>
__device__ __forceinline__ float act_bwd(float v, float o) {
    if constexpr (ACT == VL3D_ACT_SIGMOID) return o * (1.0f - o);
    else if constexpr (ACT == VL3D_ACT_RELU) return v > 0.0f ? 1.0f : 0.0f;
    else if constexpr (ACT == VL3D_ACT_CLAMP) return (v > 0.0f && v < 1.0f) ? 1.0f : 0.0f;
    else if constexpr (ACT == VL3D_ACT_ABS) return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
    else return 1.0f;
}

// ---- texel coordinates ----------------------------------------------------------------------------
// UTILS_MPI: exactly the reference's fp32 op sequence (utils_mpi.py:173 then ATen's align_corners=True
// un-normalisation ((g+1)/2)*(size-1)) so that tap weights match the reference bit-for-bit where the
// homography product does.  AFFINE: tex = p*s + o.
template <int COORD>
__device__ __forceinline__ float texel_coord(float p, float half_size, float size_m1, float s, float o) {
    if constexpr (COORD == VL3D_COORD_UTILS_MPI) {
        // p / half_size by refined reciprocal + residual correction (matches the IEEE quotient of the reference)
        float r = __builtin_amdgcn_rcpf(half_size);
        r = fmaf(fmaf(-half_size, r, 1.0f), r, r);
        float q = p * r;
        q = fmaf(fmaf(-q, half_size, p), r, q);
        float g = q - 1.0f;
        return ((g + 1.0f) * 0.5f) * size_m1;
    } else {
        return p * s + o;
    }
}

// quad index of texel coordinate a (an integer, possibly -1 or S) on an axis of S texels and n quads: floor(a * n / (S - 1)), clamped --
// in INTEGER arithmetic, so that a texel exactly on a quad border lands on the same side in every kernel and in tiles.py
__host__ __device__ inline int quad_index(int a, int S, int n) {
    const int den = S > 1 ? S - 1 : 1;
    const long long num = (long long)a * n;
    int i = (int)(num >= 0 ? num / den : -((-num + den - 1) / den));
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
