// The victim of pkfma_beside_mfma.hip as a small shared library, so that it can run beside the PRODUCT's own matrix-core kernels (scripts/coresidency_victim.py):
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libpkvictim.so pkfma_victim.hip
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool LDS>
__global__ void __launch_bounds__(256) k_victim(int iters, unsigned *hist) {   // hist[form * 2 + half][lane >> 4]
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
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a0[j]) : "v"(x), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a1[j]) : "v"(x), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[j]) : "v"(x), "v"(y));
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
                if (g[f][h] != e[f][h]) atomicAdd(&hist[(f * 2 + h) * 4 + (lane >> 4)], 1u);
    }
    if (LDS && s_pad[(threadIdx.x * 7) & 4095] < -1.f) hist[0] = 0;
}
extern "C" int pk_victim_launch(int with_lds, int blocks, int iters, unsigned *hist, void *stream) {
    if (with_lds) hipLaunchKernelGGL(k_victim<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, hist);
    else hipLaunchKernelGGL(k_victim<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, hist);
    return (int)hipGetLastError();
}
