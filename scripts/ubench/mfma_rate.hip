// What the matrix cores of an MI355X sustain on v_mfma_f32_16x16x32_bf16 with the LDS traffic of an implicit-GEMM tile around it:
// whole-chip kernels (2 workgroups of 8 waves per CU, like k_conv3x3_bf16_v2), 8 independent accumulators per wave, R ds_read_b128
// fragment reads per 8 MFMAs (conflict-free addresses), optionally one s_barrier per 24 MFMAs; random bf16 operands (zero operands clock higher: DVFS).  Prints TFLOP/s and LDS GB/s.
// hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int R, bool BAR, int M>   // M MFMAs per group (8 or 16)
__global__ void __launch_bounds__(512, 2) k(float *out, int iters, int zero) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) { unsigned h = (i + blockIdx.x * 7919u) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; reinterpret_cast<unsigned *>(lds)[i] = zero ? 0u : ((h & 0x807f807fu) | 0x3f003f00u); }   // random bf16 pairs in +-[0.5, 1)
    __syncthreads();
    f32x4 acc[M];
#pragma unroll
    for (int j = 0; j < M; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 f[2][8];
#pragma unroll
    for (int j = 0; j < 8; j++) f[0][j] = f[1][j] = *reinterpret_cast<const bf16x8 *>(lds + lane * 16 + j * 1024);
    const unsigned char *base = lds + lane * 16 + wave * 4096;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 3; g++) {
#pragma unroll
            for (int j = 0; j < R; j++) f[(g + 1) & 1][j] = *reinterpret_cast<const bf16x8 *>(base + ((it * 3 + g) & 7) * 256 + j * 1024 + (j & 1) * 32768);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < M; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[g & 1][j & 3], f[g & 1][4 + (j >> 2) % 4], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < M; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int R, bool BAR, int M>
void run(const char *name, float *out, int zero = 0) {
    const int iters = 2000, grid = 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<R, BAR, M>), dim3(grid), dim3(512), 0, 0, out, 50, zero);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k<R, BAR, M>), dim3(grid), dim3(512), 0, 0, out, iters, zero);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    const double flop = 2.0 * 16 * 16 * 32 * M * 3.0 * iters * 8 * grid;
    const double ldsb = 1024.0 * R * 3.0 * iters * 8 * grid;
    printf("%-44s %7.3f ms  %7.0f TFLOP/s  LDS %6.1f TB/s\n", name, best, flop / best / 1e9, ldsb / best / 1e9);
}
int main() {
    float *out; hipMalloc(&out, 512 * 512 * 4);
    run<0, false, 8>("8 MFMA, no LDS reads, ZERO data", out, 1);
    run<0, false, 8>("8 MFMA, no LDS reads", out);
    run<4, false, 8>("8 MFMA + 4 ds_read_b128 (0.5 / MFMA)", out);
    run<6, false, 8>("8 MFMA + 6 ds_read_b128 (0.75 / MFMA)", out);
    run<6, true, 8>("8 MFMA + 6 reads + barrier per 24 MFMA", out);
    run<8, false, 16>("16 MFMA + 8 ds_read_b128 (0.5 / MFMA)", out);
    run<8, true, 16>("16 MFMA + 8 reads + barrier per 48 MFMA", out);
    run<0, false, 16>("16 MFMA, no LDS reads", out);
    run<6, false, 16>("16 MFMA + 6 ds_read_b128 (0.375 / MFMA)", out);
    return 0;
}
