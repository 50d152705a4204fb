// Probe for gomavatar_amd/csrc/row_reduce.hpp: the transposed in-row sum of 10 registers over the four 16-lane rows of a wave
// (DPP bank_mask semantics, where the totals land) and the order of same-address ds_add_f32 lanes (two runs must agree bitwise).
//   hipcc --offload-arch=gfx950 -O3 -I gomavatar_amd/csrc scripts/ubench/row_reduce_probe.hip -o /tmp/row_reduce_probe && /tmp/row_reduce_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "row_reduce.hpp"
__global__ void k(float *out, float *acc_out, int seed) {
    __shared__ float s_acc[4][12];
    const int lane = threadIdx.x;
    float w[10];
    for (int j = 0; j < 10; j++) w[j] = (float)((j + 1) * 1000 + lane);      // row r, value j: total = 16 (j+1) 1000 + 16 (16 r) + 120
    float t0, t1, t2;
    row_sum10_t(w, t0, t1, t2);
    out[lane] = t0; out[64 + lane] = t1; out[128 + lane] = t2;
    // same-address LDS float adds: all four rows add into ONE 12-float record, values that do not sum exactly
    if (lane < 48) s_acc[lane / 12][lane % 12] = 0.f;
    __syncthreads();
    const int slot = row_sum10_slot(lane);
    const int j = lane & 3;
    float v = j == 0 ? t0 : (j == 1 ? t1 : (j == 2 ? t2 : 0.f));
    v = v * (1.0f + 1e-7f * (float)((lane * 2654435761u + seed) % 97));      // rounding-sensitive
    for (int rep = 0; rep < 4; rep++) lds_add_f32(&s_acc[rep][slot], v * (float)(rep + 1));
    __syncthreads();
    if (lane < 48) acc_out[lane] = s_acc[lane / 12][lane % 12];
}
int main() {
    float *d, *a, h[192], acc[2][48];
    hipMalloc(&d, sizeof(h)); hipMalloc(&a, sizeof(acc[0]));
    int bad = 0;
    for (int run = 0; run < 2; run++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, 7);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        hipMemcpy(acc[run], a, sizeof(acc[0]), hipMemcpyDeviceToHost);
    }
    const int perm[4] = {0, 2, 1, 3};
    for (int l = 0; l < 64; l++) {
        const int r = l >> 4, bank = (l & 15) >> 2;
        const float base = 16.f * 16.f * r + 120.f;
        const float e0 = 16000.f * (perm[bank] + 1) + base, e1 = 16000.f * (4 + perm[bank] + 1) + base, e2 = 16000.f * (8 + (bank >> 1) + 1) + base;
        if (h[l] != e0 || h[64 + l] != e1 || h[128 + l] != e2) { bad++; printf("lane %d: t0 %.0f (want %.0f) t1 %.0f (%.0f) t2 %.0f (%.0f)\n", l, h[l], e0, h[64 + l], e1, h[128 + l], e2); }
    }
    printf("row_sum10_t: %s\n", bad ? "MISMATCH" : "all 64 lanes hold the expected row totals");
    printf("ds_add_f32 same-address order: two runs %s\n", memcmp(acc[0], acc[1], sizeof(acc[0])) == 0 ? "bitwise equal" : "DIFFER");
    for (int s = 0; s < 12; s++) printf("  slot %2d: %.3f\n", s, acc[0][s]);
    return bad != 0;
}
