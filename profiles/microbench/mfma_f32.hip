// Throughput of v_mfma_f32_16x16x4_f32 as the patch-NN kernel issues it: 5 independent accumulators per wave, operands either
// constant registers (bare) or read from LDS every column (lds), 1-4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_f32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LDS>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters) {
    __shared__ float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 r[5];
    for (int j = 0; j < 5; ++j) r[j] = f32x4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    float a = lane * 0.01f, b[5] = {1.f, 2.f, 3.f, 4.f, 5.f};
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            const float *p = sm + ((it * 512) & 4095) + lane;
            a = p[0];
#pragma unroll
            for (int j = 0; j < 5; ++j) b[j] = p[256 + j * 64];
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) r[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], r[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < 5; ++j) s += r[j][0] + r[j][1] + r[j][2] + r[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *out;
    hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int lds = 0; lds < 2; ++lds)
        for (int blocks : {256, 512, 768, 1024, 2048}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (lds) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep) {
                    const double fl = (double)blocks * 4 * iters * 5 * 2048.0;
                    printf("%s  blocks %5d (%.0f waves/SIMD)  %.3f ms  %.1f TFLOP/s\n", lds ? "lds " : "bare", blocks, blocks / 256.0, ms, fl / ms * 1e-9);
                }
            }
        }
    return 0;
}
