// How long does the GPU take to dispatch workgroups that exit at once?  (A batched launch has 8 192 tiles, ~1 300 with entries:
// the per-tile kernels used to launch one workgroup per tile.)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_exit(const int *flag, int *out) { if (flag[blockIdx.x & 1023] == 12345) out[threadIdx.x] = 1; }
int main() {
    int *flag, *out; hipMalloc(&flag, 4096); hipMemset(flag, 0, 4096); hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {256, 1024, 2048, 8192, 32768}, threads[] = {64, 256, 1024};
    for (int t : threads)
        for (int g : grids) {
            for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_exit, dim3(g), dim3(t), 0, 0, flag, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_exit, dim3(g), dim3(t), 0, 0, flag, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%5d workgroups x %4d threads: %.1f us per launch\n", g, t, ms * 1000 / 50);
        }
    return 0;
}
