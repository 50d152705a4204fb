#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
__global__ void k(const uint4 *src, uint4 *dst, int mask_odd) {
    __shared__ __attribute__((aligned(16))) uint4 s[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s[i] = make_uint4(0xdead, 0, 0, 0);
    __syncthreads();
    // each wave loads 64 units: lane i reads src[(wave*64 + (63 - i))] (reversed per-lane source), lands at s[wave*64 + i]
    const uint4 *g = src + wave * 64 + (63 - lane);
    uint4 *l = s + __builtin_amdgcn_readfirstlane(wave) * 64;
    if (!mask_odd || (lane & 1) == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) dst[i] = s[i];
}
int main() {
    uint4 h[256], o[256]; for (int i = 0; i < 256; i++) h[i] = make_uint4(i, i * 2, i * 3, i * 4);
    uint4 *d, *e; hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int m = 0; m < 2; m++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, e, m);
        hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
        int ok = 1;
        for (int i = 0; i < 256; i++) {
            int w = i / 64, l = i % 64; u32 exp = (m && (l & 1)) ? 0xdead : (u32)(w * 64 + 63 - l);
            if (o[i].x != exp || (exp != 0xdead && o[i].w != exp * 4)) { if (ok) printf("mismatch at %d: got %u expected %u\n", i, o[i].x, exp); ok = 0; }
        }
        printf("mask_odd=%d %s\n", m, ok ? "OK: lane i lands at base + 16 i, masked lanes leave LDS untouched" : "FAIL");
    }
    return 0;
}
