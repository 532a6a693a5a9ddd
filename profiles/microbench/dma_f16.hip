// Three facts the split-f16 patch-NN kernel (patchnn6_k) is built on, measured on the part:
//   1. global_load_lds_dwordx4 from global sources that are only 2 / 4 / 8-byte aligned: correct?  at what rate?
//   2. v_mfma_f32_16x16x32_f16 with SUBNORMAL f16 inputs: flushed or not?
//   3. issue rates of v_mfma_f32_16x16x32_f16 / 16x16x16_f16 with 5 independent accumulators per wave.
// hipcc --offload-arch=gfx950 -O3 dma_f16.hip -o dma_f16
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(const void *g, void *lds_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)lds_wave_uniform, 16, 0, 0);
}

// every lane fetches 16 bytes from src + lane * stride + mis; the LDS image is copied out for the host to check
__global__ __launch_bounds__(64) void dma_check_k(const unsigned char *src, int stride, int mis, unsigned char *out) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[1024];
    lds_dma16(src + (size_t)threadIdx.x * stride + mis, sm);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = sm[threadIdx.x * 16 + i];
}

// rate: each wave issues `pieces` DMAs per round from a buffer that stays in L2 (lane stride like the NN kernel's: one line per lane)
__global__ __launch_bounds__(256) void dma_rate_k(const unsigned char *src, size_t span, int stride, int mis, int rounds, float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char *p = src + ((size_t)blockIdx.x * 65536) % span + (size_t)lane * stride + mis;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k) lds_dma16(p + (size_t)((r * 8 + k) * 16 % 4096) + (size_t)wave * 1048576, sm + (wave * 8 + k) * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += reinterpret_cast<float *>(sm)[threadIdx.x];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(64) void denorm_k(float *out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.0f; b[e] = (_Float16)0.0f; }
    // A[row][k = 0] = 2^-20 (subnormal in f16), B[k = 0][col] = 1024: product 2^-10 unless the input is flushed
    if ((lane >> 4) == 0) { a[0] = (_Float16)9.5367431640625e-07f; b[0] = (_Float16)1024.0f; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[lane] = c[0];
}

template <int K32>
__global__ __launch_bounds__(256, 2) void mfma_rate_k(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 r[5];
    for (int j = 0; j < 5; ++j) r[j] = f32x4{0, 0, 0, 0};
    f16x8 a8, b8[5];
    f16x4 a4, b4[5];
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(lane * 0.01f + e); for (int j = 0; j < 5; ++j) b8[j][e] = (_Float16)(j + e * 0.5f); }
    for (int e = 0; e < 4; ++e) { a4[e] = (_Float16)(lane * 0.01f + e); for (int j = 0; j < 5; ++j) b4[j][e] = (_Float16)(j + e * 0.5f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (K32) r[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8[j], r[j], 0, 0, 0);
            else r[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4[j], r[j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < 5; ++j) s += r[j][0] + r[j][1] + r[j][2] + r[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const size_t N = 256u << 20;
    unsigned char *src, *out;
    hipMalloc(&src, N + 4096);
    hipMalloc(&out, 1 << 20);
    std::vector<unsigned char> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char)((i * 2654435761u) >> 13);
    hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice);
    // 1a. correctness
    for (int stride : {16, 80, 8640})
        for (int mis : {0, 2, 4, 6, 8, 12, 1}) {
            hipLaunchKernelGGL(dma_check_k, dim3(1), dim3(64), 0, 0, src, stride, mis, out);
            std::vector<unsigned char> o(1024);
            hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 16; ++i) bad += o[l * 16 + i] != h[(size_t)l * stride + mis + i];
            printf("dma16 stride %5d mis %2d: %s (%d bad bytes)\n", stride, mis, bad ? "WRONG" : "ok", bad);
        }
    // 1b. rate (64 MiB span: L2 / MALL resident after the first round)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float *fout = reinterpret_cast<float *>(out);
    hipFuncSetAttribute((const void *)dma_rate_k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int stride : {16, 8640})
        for (int mis : {0, 8, 4, 2}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(dma_rate_k, dim3(768), dim3(256), 32 * 1024, 0, src, (size_t)(192u << 20), stride, mis, 400, fout);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("dma rate stride %5d mis %d: %.3f ms  %.2f TB/s into LDS\n", stride, mis, ms, 768.0 * 4 * 400 * 8 * 1024 / ms * 1e-9);
            }
        }
    // 2. subnormal inputs
    hipLaunchKernelGGL(denorm_k, dim3(1), dim3(64), 0, 0, fout);
    float c0;
    hipMemcpy(&c0, fout, 4, hipMemcpyDeviceToHost);
    printf("mfma f16 subnormal input: 2^-20 * 1024 = %.9g (expected %.9g if not flushed)\n", c0, 9.5367431640625e-07 * 1024);
    // 3. rates
    const int iters = 20000;
    for (int k32 = 1; k32 >= 0; --k32)
        for (int blocks : {256, 512, 1024}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (k32) hipLaunchKernelGGL(mfma_rate_k<1>, dim3(blocks), dim3(256), 0, 0, fout, iters);
                else hipLaunchKernelGGL(mfma_rate_k<0>, dim3(blocks), dim3(256), 0, 0, fout, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep) {
                    const double n = (double)blocks * 4 * iters * 5;
                    printf("mfma 16x16x%d f16  blocks %5d: %.3f ms  %.2f ns per MFMA per SIMD  (%.0f TFLOP/s)\n", k32 ? 32 : 16, blocks, ms,
                           ms * 1e6 / (n / 1024.0), n * (k32 ? 16384.0 : 8192.0) / ms * 1e-9);
                }
            }
        }
    return 0;
}
