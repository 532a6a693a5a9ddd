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

int main() {
    const size_t unit = 8ull << 30, n = unit / 16;     // 8 GiB per stream
    f4 *src, *dst;
    if (hipMalloc(&src, 3 * unit) != hipSuccess || hipMalloc(&dst, 3 * unit) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 0, 3 * unit); hipMemset(dst, 0, 3 * unit);
    char name[160];
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
    return 0;
}
