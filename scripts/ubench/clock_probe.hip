// What does s_memtime count on gfx950, and how fast does the shader clock run under load?  A full-chip grid of dependent FMAs;
// every workgroup records s_memtime (clock64) and s_memrealtime (wall_clock64, 100 MHz) deltas; the host times the launch with HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(float *out, unsigned long long *t, int iters, int heavy) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
    float a2 = a + 1.f, a3 = a + 2.f, a4 = a + 3.f;
    for (int i = 0; i < iters; i++) {
        a = __builtin_fmaf(a, b, c);
        if (heavy) { a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c); a4 = __builtin_fmaf(a4, b, c); }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + a2 + a3 + a4;
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    const int blocks = 256 * 8, threads = 256, iters = 200000;
    float *out; unsigned long long *t;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&t, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int heavy = 0; heavy < 2; heavy++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, t, iters, heavy);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(2 * blocks);
            hipMemcpy(h.data(), t, blocks * 16, hipMemcpyDeviceToHost);
            double sc = 0, sw = 0;
            for (int i = 0; i < blocks; i++) { sc += h[2 * i]; sw += h[2 * i + 1]; }
            sc /= blocks; sw /= blocks;
            printf("heavy=%d kernel %.3f ms  clock64 delta %.0f  wall(100MHz) delta %.0f -> %.3f ms  => clock64 runs at %.1f MHz; dependent FMA issue: %.2f clock64 ticks per iteration\n", heavy, ms, sc, sw,
                   sw / 1e5, sc / (sw / 100.0), sc / iters);
        }
    }
    return 0;
}
