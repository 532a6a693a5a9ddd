// Issue-rate microbenchmark for the VALU instructions the render kernels are made of (gfx950).
// Every workgroup = 1024 threads, 2 per CU -> 8 waves/SIMD, the occupancy of the render kernels.  Each thread runs
// ITER x 32 copies of one instruction on 4 independent register chains.  Prints cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define DEF_KERNEL(NAME, ASM)                                                                       \
    __global__ __launch_bounds__(1024) void NAME(float *out, int iters) {                           \
        asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 s[22:23], 0" ::: "s20", "s21", "s22", "s23");      \
        float a = threadIdx.x * 1e-3f + 1.0f, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;               \
        float e = 1.0001f, f = 0.9999f, g = 1.5f, h = 0.5f;                                          \
        for (int i = 0; i < iters; ++i) { REP8(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) :: "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) } \
        if (a + b + c + d + e + f + g + h == 12345.678f) out[0] = a;                                  \
    }

// 4 instructions per asm block (one per chain)
DEF_KERNEL(k_fma, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5")
DEF_KERNEL(k_add, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4")
DEF_KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3")
DEF_KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3")
DEF_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4")
DEF_KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4")
DEF_KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5")
DEF_KERNEL(k_med3, "v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5")
DEF_KERNEL(k_subclamp, "v_sub_f32_e64 %0, 1.0, |%0| clamp\n v_sub_f32_e64 %1, 1.0, |%1| clamp\n v_sub_f32_e64 %2, 1.0, |%2| clamp\n v_sub_f32_e64 %3, 1.0, |%3| clamp")
DEF_KERNEL(k_cvt, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3")
DEF_KERNEL(k_floor, "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3")
DEF_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 4, %4\n v_lshl_add_u32 %1, %1, 4, %4\n v_lshl_add_u32 %2, %2, 4, %4\n v_lshl_add_u32 %3, %3, 4, %4")
DEF_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc")
DEF_KERNEL(k_mov, "v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7")


DEF_KERNEL(k_mul, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4")
DEF_KERNEL(k_fmac, "v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5")
DEF_KERNEL(k_max, "v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4")
DEF_KERNEL(k_and, "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4")
DEF_KERNEL(k_add_u32, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4")
DEF_KERNEL(k_min_i32, "v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_min_i32 %3, %3, %4")
DEF_KERNEL(k_add3, "v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5")
DEF_KERNEL(k_add_lshl, "v_add_lshl_u32 %0, %0, %4, 4\n v_add_lshl_u32 %1, %1, %4, 4\n v_add_lshl_u32 %2, %2, %4, 4\n v_add_lshl_u32 %3, %3, %4, 4")
DEF_KERNEL(k_rndne, "v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3")
DEF_KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3")
DEF_KERNEL(k_fract, "v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3")
DEF_KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4")
DEF_KERNEL(k_cmp_e64, "v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cmp_lt_f32_e64 s[22:23], %1, %4\n v_cmp_lt_f32_e64 s[24:25], %2, %4\n v_cmp_lt_f32_e64 s[26:27], %3, %4")
DEF_KERNEL(k_cndmask_s, "v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]")
DEF_KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %5, vcc")
DEF_KERNEL(k_fma_sgpr, "v_fma_f32 %0, s20, %0, %5\n v_fma_f32 %1, s21, %1, %5\n v_fma_f32 %2, s22, %2, %5\n v_fma_f32 %3, s23, %3, %5")
//DEF_KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 4, %0\n v_lshl_add_u64 %1, %1, 4, %1\n v_lshl_add_u64 %2, %2, 4, %2\n v_lshl_add_u64 %3, %3, 4, %3")

// packed: 2 instructions per asm block on register pairs
#define DEF_KERNEL_PK(NAME, ASM)                                                                    \
    __global__ __launch_bounds__(1024) void NAME(float *out, int iters) {                           \
        typedef float f2 __attribute__((ext_vector_type(2)));                                        \
        f2 a = {threadIdx.x * 1e-3f + 1.0f, 2.0f}, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;          \
        f2 e = {1.0001f, 0.9999f}, f = {0.5f, 0.25f};                                                \
        for (int i = 0; i < iters; ++i) { REP8(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));) } \
        if (a.x + b.x + c.x + d.x + a.y + b.y + c.y + d.y == 12345.678f) out[0] = a.x;                \
    }
DEF_KERNEL_PK(k_pk_fma, "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5")
DEF_KERNEL_PK(k_pk_add, "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4")
DEF_KERNEL_PK(k_pk_mul, "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4")

// LDS reads: 4 per block into throw-away registers, address per lane = consecutive float4 / float2
#define DEF_KERNEL_DS(NAME, ASM, T)                                                                 \
    __global__ __launch_bounds__(1024) void NAME(float *out, int iters) {                           \
        __shared__ T buf[2048];                                                                      \
        buf[threadIdx.x] = T{}; buf[threadIdx.x + 1024] = T{};                                        \
        __syncthreads();                                                                             \
        unsigned addr = (unsigned)(size_t)(&buf[threadIdx.x]) ;                                       \
        T r0{}, r1{}, r2{}, r3{};                                                                    \
        for (int i = 0; i < iters; ++i) { REP8(asm volatile(ASM "\n s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr));) } \
        if (r0.x + r1.x + r2.x + r3.x == 12345.678f) out[0] = r0.x;                                   \
    }
DEF_KERNEL_DS(k_ds_b128, "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:1024\n ds_read_b128 %3, %4 offset:1040", float4)
DEF_KERNEL_DS(k_ds_b64, "ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:512\n ds_read_b64 %3, %4 offset:520", float2)

template <typename K>
void run(const char *name, K kern, int per_block, float *out) {
    const int iters = 2000, blocks = 512;   // 2 workgroups per CU on 256 CUs
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: 8 waves x iters x 8 x per_block
    const double winstr = 8.0 * iters * 8.0 * per_block;
    printf("%-12s %8.3f ms   %6.2f ns per wave-instr per SIMD  (= %5.2f clk @2.4GHz)\n", name, ms, ms * 1e6 / winstr, ms * 1e6 / winstr * 2.4);
}

int main() {
    float *out;
    hipMalloc(&out, 64);
    run("v_fma_f32", k_fma, 4, out); run("v_add_f32", k_add, 4, out); run("v_exp_f32", k_exp, 4, out); run("v_rcp_f32", k_rcp, 4, out);
    run("v_mul_lo_u32", k_mul_lo, 4, out); run("v_mul_u32_u24", k_mul_u24, 4, out); run("v_mad_u32_u24", k_mad_u24, 4, out);
    run("v_med3_f32", k_med3, 4, out); run("v_sub|clamp", k_subclamp, 4, out); run("v_cvt_i32", k_cvt, 4, out); run("v_floor", k_floor, 4, out);
    run("v_lshl_add", k_lshl_add, 4, out); run("v_cndmask", k_cndmask, 4, out); run("v_mov", k_mov, 4, out);
    run("v_mul_f32", k_mul, 4, out); run("v_fmac_f32", k_fmac, 4, out); run("v_max_f32", k_max, 4, out); run("v_and_b32", k_and, 4, out);
    run("v_add_u32", k_add_u32, 4, out); run("v_min_i32", k_min_i32, 4, out); run("v_add3_u32", k_add3, 4, out); run("v_add_lshl", k_add_lshl, 4, out);
    run("v_rndne", k_rndne, 4, out); run("v_cvt_f32_i32", k_cvt_f32_i32, 4, out); run("v_fract", k_fract, 4, out);
    run("v_cmp vcc", k_cmp, 4, out); run("v_cmp sgpr", k_cmp_e64, 4, out); run("v_cndmask s", k_cndmask_s, 4, out); run("cmp+cnd x2", k_cmp_cnd, 4, out);
    run("v_fma sgpr", k_fma_sgpr, 4, out);
    run("v_pk_fma_f32", k_pk_fma, 4, out); run("v_pk_add_f32", k_pk_add, 4, out); run("v_pk_mul_f32", k_pk_mul, 4, out);
    run("ds_read_b128", k_ds_b128, 4, out); run("ds_read_b64", k_ds_b64, 4, out);
    return 0;
}
