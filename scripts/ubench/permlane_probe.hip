// Probe for the lane layout of the transposed 10-value wave reduction used by k_seg_bwd (v_permlane32_swap /
// v_permlane16_swap halve the number of live registers before the in-row DPP steps).  Prints, for every value j,
// the lane and register where its 64-lane total lands.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float sw32(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sw16(float a, float b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__global__ void k(float *out) {
    const int lane = threadIdx.x;
    float w[10];
    for (int j = 0; j < 10; j++) w[j] = (float)((j + 1) * 1000 + lane);
    float p0 = sw32(w[0], w[1]), p1 = sw32(w[2], w[3]), p2 = sw32(w[4], w[5]), p3 = sw32(w[6], w[7]), p4 = sw32(w[8], w[9]);
    float s0 = sw16(p0, p1), s1 = sw16(p2, p3);
#define ST(CTRL) "v_add_f32_dpp %0, %0, %0 " CTRL "\n v_add_f32_dpp %1, %1, %1 " CTRL "\n v_add_f32_dpp %2, %2, %2 " CTRL "\n"
    asm volatile("s_nop 1\n" ST("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") ST("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 ST("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0") ST("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 "s_nop 1\n v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                 : "+v"(s0), "+v"(s1), "+v"(p4));
    out[lane] = s0; out[64 + lane] = s1; out[128 + lane] = p4;
}
int main() {
    float *d, h[192];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 3; r++)
        for (int l = 15; l < 64; l += 16) {
            const float v = h[r * 64 + l];
            const int j = (int)((v - 2016.f) / 64000.f + 0.5f) - 1;
            printf("reg s%d lane %d: %.0f -> value %d %s\n", r, l, v, j, (v == 64000.f * (j + 1) + 2016.f) ? "(exact total)" : "(not a total)");
        }
    return 0;
}
