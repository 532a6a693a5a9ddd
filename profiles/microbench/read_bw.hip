// Read-only HBM streaming ceiling on MI355X: what a hand-written kernel reaches when it only READS (the forward render's
// pattern: 24 GB in, 0.55 GB out).  Sweeps loads in flight per thread, workgroup size and grid size over a 24 GiB buffer.
//   hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip && ./read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ void read_k(const f4 *__restrict__ p, size_t n, float *out) {
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    for (; i < n; i += stride) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 1234.5678f) out[0] = acc.x;
}

// tile-shaped variant: a workgroup streams one contiguous 1 MiB chunk after the other (block-contiguous instead of grid-strided)
template <int UNROLL>
__global__ void read_chunk_k(const f4 *__restrict__ p, size_t n, float *out) {
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    const size_t chunk = 65536;     // f4 elements = 1 MiB
    for (size_t c = blockIdx.x; c * chunk < n; c += gridDim.x) {
        const f4 *q = p + c * chunk;
        for (size_t i = threadIdx.x; i + (UNROLL - 1) * blockDim.x < chunk; i += UNROLL * blockDim.x) {
            f4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = q[i + u * blockDim.x];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1234.5678f) out[0] = acc.x;
}

template <typename F>
static void run(const char *name, F launch, size_t bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-44s %7.3f ms  %5.2f TB/s\n", name, ms, bytes / ms / 1e9);
}

int main() {
    const size_t bytes = 24ull << 30, n = bytes / 16;
    f4 *p; float *out;
    hipMalloc(&p, bytes); hipMalloc(&out, 4);
    hipMemset(p, 0, bytes);
    char name[128];
#define SWEEP(K, U)                                                                                     \
    for (int bs : {256, 512, 1024})                                                                      \
        for (int per_cu : {2, 4, 8, 16})                                                                 \
            if (bs * per_cu <= 2048 * 4) {                                                               \
                const int grid = 256 * per_cu;                                                           \
                snprintf(name, sizeof name, #K " unroll %d  block %4d  grid %5d", U, bs, grid);          \
                run(name, [&] { hipLaunchKernelGGL((K<U>), dim3(grid), dim3(bs), 0, 0, p, n, out); }, bytes); \
            }
    SWEEP(read_k, 1) SWEEP(read_k, 4) SWEEP(read_k, 8)
    SWEEP(read_chunk_k, 4) SWEEP(read_chunk_k, 8)
    // one-shot grid (a thread reads 4 x 16 B, no loop): the forward render's launch shape
    {
        const int bs = 512;
        const size_t grid = n / (4 * bs);
        run("read_k unroll 4  one-shot grid", [&] { hipLaunchKernelGGL((read_k<4>), dim3((unsigned)grid), dim3(bs), 0, 0, p, n, out); }, bytes);
    }
    return 0;
}
