// Microbenchmarks that decided the reduction design of k_seg_bwd: cycles per instruction for DPP adds,
// v_readlane, ds_bpermute, v_permlane32_swap and plain VALU, one wave per SIMD and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2000
__global__ void k_dpp(float *out, unsigned long long *cyc) {
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, v8 = v0 + 8, v9 = v0 + 9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; i++) {
        asm volatile(
            "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
            "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
            "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
            "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
            "v_add_f32_dpp %8, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %9, %9, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(v9));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + v8 + v9;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_bcast(float *out, unsigned long long *cyc) {
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, v8 = v0 + 8, v9 = v0 + 9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; i++) {
        asm volatile(
            "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
            "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
            "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
            "v_add_f32_dpp %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
            "v_add_f32_dpp %8, %8, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %9, %9, %9 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(v9));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + v8 + v9;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma(float *out, unsigned long long *cyc) {
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, v8 = v0 + 8, v9 = v0 + 9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; i++) {
        asm volatile(
            "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n"
            "v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n"
            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(v9));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + v8 + v9;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_chain(float *out, unsigned long long *cyc) {   // fully dependent fma chain
    float v0 = threadIdx.x * 1e-9f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; i++) {
        asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n"
                     "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n" : "+v"(v0));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_readlane(float *out, unsigned long long *cyc, int lane) {
    float v0 = threadIdx.x, acc = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; i++) {
        int l = (lane + i) & 63;
        l = __builtin_amdgcn_readfirstlane(l);
#pragma unroll
        for (int u = 0; u < 10; u++) acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v0 + u), l));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_swap(float *out, unsigned long long *cyc) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, a8 = a0 + 8, a9 = a0 + 9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; i++) {
        asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %8, %9\n"
                     "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc, h; hipEvent_t e0, e1; float ms; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMalloc(&out, 1 << 26); hipMalloc(&cyc, 8);
    struct { const char *name; int threads, blocks; } cfgs[] = {{"1 wave/SIMD (256 thr), 1 block", 256, 1}, {"4 waves/SIMD (1024 thr), 1 block", 1024, 1},
                                                                {"4 waves/SIMD, 256 blocks (whole chip)", 1024, 256}, {"8 waves/SIMD, 512 blocks (whole chip)", 1024, 512}};
    for (auto &c : cfgs) {
        printf("== %s: cycles per instruction (per wave, block 0) ==\n", c.name);
#define RUN(K, ...) for (int rep = 0; rep < 2; rep++) { hipEventRecord(e0, 0); hipLaunchKernelGGL(K, dim3(c.blocks), dim3(c.threads), 0, 0, out, cyc, ##__VA_ARGS__); hipEventRecord(e1, 0); hipDeviceSynchronize(); } \
        hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
        printf("  %-14s %.2f cycles/instr (wave 0 counter)   kernel %.1f us -> %.2f ns per wave-instr per SIMD\n", #K, (double)h / (ITERS * 10.0), ms * 1e3, \
               ms * 1e6 / (ITERS * 10.0 * ((double)c.blocks * c.threads / 64 / (c.blocks < 256 ? 4.0 * c.blocks : 1024.0))));
        RUN(k_fma) RUN(k_chain) RUN(k_dpp) RUN(k_bcast) RUN(k_readlane, 3) RUN(k_swap)
    }
    return 0;
}
