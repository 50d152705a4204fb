// Does a packed-fp32 FMA give a wrong result when waves of ANOTHER kernel issue matrix-core instructions on the same SIMD?  (LABBOOK R6.8: the weight-gradient
// kernel of the shadow MLP, k_mlp3_wgrad_partial, returned wrong sums -- rows o = 13 mod 16, even columns: lanes 48..63 of the accumulators written by
// `v_pk_fma_f32 ... op_sel:[0,1,0]` -- whenever the bf16x3 matrix-core layers of mlp_mc.hip ran beside it, from another process or another stream.)
// VICTIM: 256-thread workgroups, 16 KB of LDS, exact small-integer packed FMAs in three operand-select forms, results checked on the device against the closed form.
// AGGRESSORS on a second stream: 256-thread workgroups with 35 KB of LDS (so that both kernels fit a CU side by side), a loop of one instruction kind.
// hipcc --offload-arch=gfx950 -O3 -o pkfma_beside_mfma pkfma_beside_mfma.hip && ./pkfma_beside_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4s;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// hist[form][half][lane >> 4]
template <bool LDS>
__global__ void __launch_bounds__(256) k_victim(int iters, unsigned *hist, float *sample) {
    __shared__ float s_pad[LDS ? 4096 : 1];
    if (LDS) { s_pad[threadIdx.x] = (float)threadIdx.x; __syncthreads(); }
    const int lane = threadIdx.x & 63;
    const f32x2 x = {(float)(1 + lane % 3), (float)(2 + lane % 5)}, y = {3.f, (float)(5 + (lane & 1))};
    f32x2 a0[4], a1[4], a2[4];
#pragma unroll
    for (int j = 0; j < 4; j++) a0[j] = a1[j] = a2[j] = f32x2{0.f, 0.f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a0[j]) : "v"(x), "v"(y));          // lo += x.lo * y.HI, hi += x.hi * y.hi
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a1[j]) : "v"(x), "v"(y));       // lo += x.lo * y.lo, hi += x.hi * y.LO
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[j]) : "v"(x), "v"(y));                         // lo += x.lo * y.lo, hi += x.hi * y.hi
        }
    }
    const float n = (float)iters;
    const float e[3][2] = {{n * x[0] * y[1], n * x[1] * y[1]}, {n * x[0] * y[0], n * x[1] * y[0]}, {n * x[0] * y[0], n * x[1] * y[1]}};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const f32x2 g[3] = {a0[j], a1[j], a2[j]};
#pragma unroll
        for (int f = 0; f < 3; f++)
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (g[f][h] != e[f][h]) {
                    if (atomicAdd(&hist[(f * 2 + h) * 4 + (lane >> 4)], 1u) == 0u) { sample[(f * 2 + h) * 2] = g[f][h]; sample[(f * 2 + h) * 2 + 1] = e[f][h]; }
                }
    }
    if (LDS && s_pad[(threadIdx.x * 7) & 4095] < -1.f) hist[0] = 0;
}

template <int KIND>
__global__ void __launch_bounds__(256, 2) k_aggr(int iters, float *out, const bf16x8 *src) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[35840];
    for (int i = threadIdx.x; i < 35840 / 4; i += 256) { unsigned h = (i + blockIdx.x * 7919u) * 2654435761u; h ^= h >> 15; reinterpret_cast<unsigned *>(lds)[i] = (h & 0x807f807fu) | 0x3f003f00u; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 f[4];
#pragma unroll
    for (int j = 0; j < 4; j++) f[j] = *reinterpret_cast<const bf16x8 *>(lds + lane * 16 + j * 1024);
    float v = (float)lane;
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {          // v_mfma_f32_16x16x32_bf16, register operands
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[j & 3], f[(j >> 1) & 3], acc[j], 0, 0, 0);
        } else if (KIND == 1) {   // v_mfma_f32_16x16x16_bf16 (the K = 16 form of CDNA3)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const s16x4 a = __builtin_bit_cast(s16x4, __builtin_shufflevector(f[j & 3], f[j & 3], 0, 1, 2, 3)), b = __builtin_bit_cast(s16x4, __builtin_shufflevector(f[(j >> 1) & 3], f[(j >> 1) & 3], 4, 5, 6, 7));
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[j], 0, 0, 0);
            }
        } else if (KIND == 2) {   // fp32 VALU only
#pragma unroll
            for (int j = 0; j < 8; j++) { acc[j][0] = __builtin_fmaf(acc[j][0], 0.999f, v); acc[j][1] = __builtin_fmaf(acc[j][1], 0.998f, v); acc[j][2] = __builtin_fmaf(acc[j][2], 0.997f, v); acc[j][3] = __builtin_fmaf(acc[j][3], 0.996f, v); }
        } else {   // KIND >= 3, like mlp_mc.hip: A fragments from global memory, B fragments from LDS, three MFMAs per pair, a barrier per k-step group;
                   // bits of KIND - 3 take one ingredient out: 1 = no MFMA (the VALU consumes the fragments), 2 = no global loads, 4 = no LDS reads, 8 = no barrier
            constexpr int OFF = KIND - 3;
            bf16x8 a[2], b[4];
#pragma unroll
            for (int j = 0; j < 2; j++) a[j] = (OFF & 2) ? f[j] : src[(size_t)((it * 2 + j) & 255) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = (OFF & 4) ? f[j] : *reinterpret_cast<const bf16x8 *>(lds + ((it & 3) * 4352) + (j * 16 + (lane & 15)) * 272 + (lane >> 4) * 16);
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    if (OFF & 1) {
                        const f32x4 ua = __builtin_bit_cast(f32x4, a[n]), ub = __builtin_bit_cast(f32x4, b[m]);
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[m * 2 + n][r] = __builtin_fmaf(ua[r], 1e-30f, __builtin_fmaf(ub[r], 1e-30f, acc[m * 2 + n][r]));
                    } else {
                        acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[n], b[m], acc[m * 2 + n], 0, 0, 0);
                        acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[n], f[m], acc[m * 2 + n], 0, 0, 0);
                        acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[n], b[m], acc[m * 2 + n], 0, 0, 0);
                    }
                }
            if (!(OFF & 8) && (it & 3) == 3) __syncthreads();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The resource shapes of the product's own matrix-core kernels (vgg_bf16.hip): 8-wave workgroups, `extra` bytes of dynamic LDS on top of 16 KB of fragments
// (k_conv3x3_bf16_v2: two workgroups of ~76 KB per CU, <= 128 VGPRs -> the register file of every SIMD is full; k_conv3x3_x3s: one workgroup of 157 KB per CU).
template <int MINB, bool PIPE = false, bool BIGV = false>
__global__ void __launch_bounds__(512, MINB) k_aggr_shape(int iters, float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dl[];
    for (int i = threadIdx.x; i < 16384 / 4; i += 512) { unsigned h = (i + blockIdx.x * 7919u) * 2654435761u; h ^= h >> 15; reinterpret_cast<unsigned *>(dl)[i] = (h & 0x807f807fu) | 0x3f003f00u; }
    __syncthreads();
    if (BIGV) asm volatile("v_mov_b32 v200, 0" ::: "v200");   // a register allocation as large as k_conv3x3_x3s's (152) and beyond: the wave sits elsewhere in the register file
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load = [&](int it, bf16x8 (&a)[2], bf16x8 (&b)[4]) {
#pragma unroll
        for (int j = 0; j < 2; j++) a[j] = *reinterpret_cast<const bf16x8 *>(dl + ((it + j) & 3) * 4096 + lane * 16);
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = *reinterpret_cast<const bf16x8 *>(dl + ((it & 3) * 4096) + ((j * 16 + (lane & 15)) * 64 + (lane >> 4) * 16) % 4096);
    };
    bf16x8 a0[2], b0[4], a1[2], b1[4];
    auto mm = [&](bf16x8 (&a)[2], bf16x8 (&b)[4]) {
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[n], b[m], acc[m * 2 + n], 0, 0, 0);
    };
    if (PIPE) {      // register double buffer, unrolled by two: a trip's fragments are requested BEFORE the previous trip's MFMAs issue (the structure of the product's trunk kernels)
        load(0, a0, b0);
        for (int it = 0; it < iters; it += 2) {
            load(it + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mm(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            load(it + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mm(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        for (int it = 0; it < iters; it++) { load(it, a0, b0); mm(a0, b0); }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// PK_FORMS=1: which OTHER operand-select forms / packed instructions are vulnerable?  Same exact-integer chains, six more forms; hist2[form][half][lane >> 4].
__global__ void __launch_bounds__(256) k_victim_forms(int iters, unsigned *hist2) {
    const int lane = threadIdx.x & 63;
    const f32x2 x = {(float)(1 + lane % 3), (float)(2 + lane % 5)}, y = {3.f, (float)(5 + (lane & 1))};
    f32x2 a[6];
#pragma unroll
    for (int j = 0; j < 6; j++) a[j] = f32x2{0.f, 0.f};
    for (int it = 0; it < iters; it++) {
        f32x2 t;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(a[0]) : "v"(x), "v"(y));           // lo += x.HI * y.lo, hi += x.hi * y.hi   (src0 crosses)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[1]) : "v"(x), "v"(y));        // lo += x.lo * y.lo, hi += x.LO * y.hi   (src0 crosses, high half)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0]" : "+v"(a[2]) : "v"(x), "v"(y));           // lo += x.HI * y.HI, hi += x.hi * y.hi   (both cross)
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(x), "v"(y));                     // t.lo = x.lo * y.HI, t.hi = x.hi * y.hi
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[3]) : "v"(t));
        asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(a[4]) : "v"(y));                          // lo += y.HI, hi += y.hi
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(a[5]) : "v"(x), "v"(y));   // lo = x.lo * y.lo + acc.HI, hi = x.hi * y.hi + acc.hi   (src2 crosses)
    }
    const float n = (float)iters;
    // a[5]: hi = n x1 y1;  lo_k = x0 y0 + hi_{k-1}  ->  lo_n = x0 y0 + (n - 1) x1 y1
    const float e[6][2] = {{n * x[1] * y[0], n * x[1] * y[1]}, {n * x[0] * y[0], n * x[0] * y[1]}, {n * x[1] * y[1], n * x[1] * y[1]}, {n * x[0] * y[1], n * x[1] * y[1]},
                           {n * y[1], n * y[1]}, {x[0] * y[0] + (n - 1.f) * x[1] * y[1], n * x[1] * y[1]}};
#pragma unroll
    for (int f = 0; f < 6; f++)
#pragma unroll
        for (int h = 0; h < 2; h++)
            if (a[f][h] != e[f][h]) atomicAdd(&hist2[(f * 2 + h) * 4 + (lane >> 4)], 1u);
}

int main() {
    unsigned *hist; float *sample, *out; bf16x8 *src;
    CK(hipMalloc(&hist, 24 * 4)); CK(hipMalloc(&sample, 12 * 4)); CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&src, 256 * 64 * 16));
    CK(hipMemset(src, 0x3f, 256 * 64 * 16));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const char *names[] = {"no aggressor", "v_mfma_f32_16x16x32_bf16, register operands", "v_mfma_f32_16x16x16_bf16", "fp32 VALU only", "16x16x32 with A from global, B from LDS, barriers (the shape of mlp_mc.hip)",
                           "... without the MFMAs (VALU consumes the fragments)", "... without the global loads", "... without the LDS reads", "... without the barriers", "... LDS reads + barriers + VALU only", "... global loads + VALU only",
                           "... MFMA + LDS reads only",
                           "MFMA + LDS reads in the shape of k_conv3x3_bf16_v2 (8 waves, 76 KB of LDS, two workgroups per CU)", "MFMA + LDS reads in the shape of k_conv3x3_x3s (8 waves, 157 KB of LDS, one workgroup per CU)",
                           "... the same (157 KB), victim WITHOUT LDS", "... 76 KB twice per CU, victim WITHOUT LDS", "the shape of mlp_mc.hip, MFMA + LDS reads only, victim WITHOUT LDS",
                           "157 KB shape, LDS reads one trip AHEAD of their MFMAs (software pipeline), victim WITHOUT LDS", "157 KB shape, > 200 VGPRs allocated, victim WITHOUT LDS",
                           "157 KB shape, pipelined AND > 200 VGPRs, victim WITHOUT LDS"};
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_aggr_shape<1, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_aggr_shape<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_aggr_shape<1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_aggr_shape<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_aggr_shape<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const char *forms[3] = {"op_sel:[0,1,0]   ", "op_sel_hi:[1,0,1]", "plain            "};
    for (int k = -1; k < (getenv("PK_MORE") ? 19 : 16); k++) {   // PK_MORE=1: three more shapes of the 157 KB aggressor (reads a trip ahead in a register double buffer, a clobbered v200)
        CK(hipMemset(hist, 0, 24 * 4)); CK(hipMemset(sample, 0, 12 * 4)); CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, sa));
        for (int rep = 0; rep < 24; rep++) {
            if (k == 0) hipLaunchKernelGGL(k_aggr<0>, dim3(2048), dim3(256), 0, sa, 6000, out, src);
            if (k == 1) hipLaunchKernelGGL(k_aggr<1>, dim3(2048), dim3(256), 0, sa, 6000, out, src);
            if (k == 2) hipLaunchKernelGGL(k_aggr<2>, dim3(2048), dim3(256), 0, sa, 6000, out, src);
            if (k == 3) hipLaunchKernelGGL(k_aggr<3>, dim3(2048), dim3(256), 0, sa, 2000, out, src);
            if (k == 4) hipLaunchKernelGGL(k_aggr<3 + 1>, dim3(2048), dim3(256), 0, sa, 2000, out, src);
            if (k == 5) hipLaunchKernelGGL(k_aggr<3 + 2>, dim3(2048), dim3(256), 0, sa, 2000, out, src);
            if (k == 6) hipLaunchKernelGGL(k_aggr<3 + 4>, dim3(2048), dim3(256), 0, sa, 2000, out, src);
            if (k == 7) hipLaunchKernelGGL(k_aggr<3 + 8>, dim3(2048), dim3(256), 0, sa, 2000, out, src);
            if (k == 8) hipLaunchKernelGGL(k_aggr<3 + 1 + 2>, dim3(2048), dim3(256), 0, sa, 2000, out, src);
            if (k == 9) hipLaunchKernelGGL(k_aggr<3 + 1 + 4 + 8>, dim3(2048), dim3(256), 0, sa, 2000, out, src);
            if (k == 10 || k == 15) hipLaunchKernelGGL(k_aggr<3 + 2 + 8>, dim3(2048), dim3(256), 0, sa, 2000, out, src);
            if (k == 11 || k == 14) hipLaunchKernelGGL(k_aggr_shape<2>, dim3(1024), dim3(512), 76 * 1024, sa, 6000, out);
            if (k == 12 || k == 13) hipLaunchKernelGGL(k_aggr_shape<1>, dim3(1024), dim3(512), 157 * 1024, sa, 6000, out);
            if (k == 16) hipLaunchKernelGGL((k_aggr_shape<1, true, false>), dim3(1024), dim3(512), 157 * 1024, sa, 6000, out);
            if (k == 17) hipLaunchKernelGGL((k_aggr_shape<1, false, true>), dim3(1024), dim3(512), 157 * 1024, sa, 6000, out);
            if (k == 18) hipLaunchKernelGGL((k_aggr_shape<1, true, true>), dim3(1024), dim3(512), 157 * 1024, sa, 6000, out);
        }
        CK(hipEventRecord(e1, sa));
        for (int rep = 0; rep < 1200; rep++) {
            if (k >= 13) hipLaunchKernelGGL(k_victim<false>, dim3(1024), dim3(256), 0, sb, 4096, hist, sample);
            else hipLaunchKernelGGL(k_victim<true>, dim3(1024), dim3(256), 0, sb, 4096, hist, sample);
        }
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h[24]; float s[12];
        CK(hipMemcpy(h, hist, sizeof h, hipMemcpyDeviceToHost)); CK(hipMemcpy(s, sample, sizeof s, hipMemcpyDeviceToHost));
        printf("== %s (aggressor stream busy %.1f ms)\n", names[k + 1], ms);
        unsigned total = 0;
        for (int q = 0; q < 24; q++) total += h[q];
        if (!total) printf("   every result exact (all three forms, both halves, every lane)\n");
        for (int f = 0; f < 3; f++)
            for (int hf = 0; hf < 2; hf++) {
                const unsigned *q = h + (f * 2 + hf) * 4;
                if (!(q[0] + q[1] + q[2] + q[3])) continue;
                printf("   %s %s half: wrong results in lanes 0-15 / 16-31 / 32-47 / 48-63: %u / %u / %u / %u   (e.g. got %.1f, expected %.1f)\n", forms[f], hf ? "high" : "low ", q[0], q[1], q[2], q[3],
                       s[(f * 2 + hf) * 2], s[(f * 2 + hf) * 2 + 1]);
            }
    }
    if (getenv("PK_FORMS")) {
        unsigned *hist2; CK(hipMalloc(&hist2, 48 * 4));
        const char *fn[6] = {"v_pk_fma_f32 op_sel:[1,0,0] (src0 crosses into the low half)", "v_pk_fma_f32 op_sel_hi:[0,1,1] (src0 crosses into the high half)", "v_pk_fma_f32 op_sel:[1,1,0] (both sources cross)",
                             "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[0,1]", "v_pk_fma_f32 op_sel:[0,0,1] (the ADDEND crosses)"};
        for (int with = 0; with < 2; with++) {
            CK(hipMemset(hist2, 0, 48 * 4)); CK(hipDeviceSynchronize());
            if (with) for (int rep = 0; rep < 24; rep++) hipLaunchKernelGGL((k_aggr<3 + 2 + 8>), dim3(2048), dim3(256), 0, sa, 2000, out, src);
            for (int rep = 0; rep < 1200; rep++) hipLaunchKernelGGL(k_victim_forms, dim3(1024), dim3(256), 0, sb, 4096, hist2);
            CK(hipDeviceSynchronize());
            unsigned h[48]; CK(hipMemcpy(h, hist2, sizeof h, hipMemcpyDeviceToHost));
            printf("== other forms, %s\n", with ? "beside MFMA + LDS reads" : "no aggressor");
            for (int f = 0; f < 6; f++)
                for (int hf = 0; hf < 2; hf++) {
                    const unsigned *q = h + (f * 2 + hf) * 4;
                    printf("   %-66s %s half: wrong in lanes 0-15 / 16-31 / 32-47 / 48-63: %u / %u / %u / %u\n", fn[f], hf ? "high" : "low ", q[0], q[1], q[2], q[3]);
                }
        }
    }
    return 0;
}
