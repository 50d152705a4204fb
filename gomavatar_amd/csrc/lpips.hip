// LPIPS head (reference utils/lpips/lpips.py:81-123 with utils/lpips/__init__.py:40-42), one layer per launch:
//   value_b = mean_{hw} sum_c w_c * (f0_c / m0 - f1_c / m1)^2,   m = sqrt(sum_c f_c^2 + 1e-10) + 1e-10
// i.e. normalize_tensor of both feature maps, squared difference, the 1x1 "lin" convolution (non-negative weights,
// no bias, Dropout inert in eval mode) and the spatial average -- the reference runs ~12 element-wise / reduction
// launches with (B,C,H,W) temporaries per layer; here the features are read straight from the trunk's output.
// One thread per pixel, channels strided by HW (coalesced across the wave).  HBM/L2-bound: the forward reads each
// feature map twice (norms, then the weighted difference; the second read hits L2), the backward three times and
// writes the gradient once.  No atomics: per-block partial sums, summed by the caller in a fixed order.
#include "gom_internal.h"

namespace {

constexpr float kEps = 1e-10f;

__device__ __forceinline__ float block_sum_256(float v, float *s_red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ void __launch_bounds__(256) k_lpips_layer_fwd(int C, int HW, const float *__restrict__ f0, const float *__restrict__ f1,
                                                         const float *__restrict__ w, float *__restrict__ partials) {
    __shared__ float s_red[4];
    const size_t b = blockIdx.y;
    f0 += b * (size_t)C * HW;
    f1 += b * (size_t)C * HW;
    float acc = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        float s0 = 0.f, s1 = 0.f;
        for (int c = 0; c < C; c++) {
            const float a = f0[(size_t)c * HW + p], bb = f1[(size_t)c * HW + p];
            s0 += a * a;
            s1 += bb * bb;
        }
        const float i0 = 1.f / (sqrtf(s0 + kEps) + kEps), i1 = 1.f / (sqrtf(s1 + kEps) + kEps);
        float v = 0.f;
        for (int c = 0; c < C; c++) {
            const float d = f0[(size_t)c * HW + p] * i0 - f1[(size_t)c * HW + p] * i1;
            v += w[c] * (d * d);
        }
        acc += v;
    }
    const float tot = block_sum_256(acc, s_red);
    if (threadIdx.x == 0) partials[b * gridDim.x + blockIdx.x] = tot / (float)HW;
}

// d value_b / d f0, times grad_out[b]:  G_k = (2 / HW) w_k (f0_k/m0 - f1_k/m1),
// d/df0_c = G_c / m0 - f0_c / (m0^2 n0) * sum_k G_k f0_k,   n0 = sqrt(sum f0^2 + eps), m0 = n0 + eps
__global__ void __launch_bounds__(256) k_lpips_layer_bwd(int C, int HW, const float *__restrict__ f0, const float *__restrict__ f1,
                                                         const float *__restrict__ w, const float *__restrict__ grad_out,
                                                         float *__restrict__ d_f0) {
    const size_t b = blockIdx.y;
    f0 += b * (size_t)C * HW;
    f1 += b * (size_t)C * HW;
    d_f0 += b * (size_t)C * HW;
    const float go = grad_out[b] * (2.0f / (float)HW);
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        float s0 = 0.f, s1 = 0.f;
        for (int c = 0; c < C; c++) {
            const float a = f0[(size_t)c * HW + p], bb = f1[(size_t)c * HW + p];
            s0 += a * a;
            s1 += bb * bb;
        }
        const float n0 = sqrtf(s0 + kEps);
        const float i0 = 1.f / (n0 + kEps), i1 = 1.f / (sqrtf(s1 + kEps) + kEps);
        float dot = 0.f;
        for (int c = 0; c < C; c++) {
            const float a = f0[(size_t)c * HW + p];
            dot += w[c] * (a * i0 - f1[(size_t)c * HW + p] * i1) * a;
        }
        const float k = dot * i0 * i0 / n0;
        for (int c = 0; c < C; c++) {
            const float a = f0[(size_t)c * HW + p];
            const float g = w[c] * (a * i0 - f1[(size_t)c * HW + p] * i1);
            d_f0[(size_t)c * HW + p] = go * (g * i0 - a * k);
        }
    }
}

}  // namespace

extern "C" int gom_lpips_layer_forward(int B, int C, int HW, const float *f0, const float *f1, const float *w, float *partials,
                                       void *stream) {
    if (B <= 0 || C <= 0 || HW <= 0) { gom_set_error("gom_lpips_layer_forward: bad sizes"); return -1; }
    if (!f0 || !f1 || !w || !partials) { gom_set_error("gom_lpips_layer_forward: null pointer"); return -1; }
    hipLaunchKernelGGL(k_lpips_layer_fwd, dim3(GOM_LOSS_BLOCKS, B), dim3(256), 0, (hipStream_t)stream, C, HW, f0, f1, w, partials);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_lpips_layer_backward(int B, int C, int HW, const float *f0, const float *f1, const float *w, const float *grad_out,
                                        float *d_f0, void *stream) {
    if (B <= 0 || C <= 0 || HW <= 0) { gom_set_error("gom_lpips_layer_backward: bad sizes"); return -1; }
    if (!f0 || !f1 || !w || !grad_out || !d_f0) { gom_set_error("gom_lpips_layer_backward: null pointer"); return -1; }
    const int blocks = (HW + 255) / 256;
    hipLaunchKernelGGL(k_lpips_layer_bwd, dim3(blocks < 4096 ? blocks : 4096, B), dim3(256), 0, (hipStream_t)stream, C, HW, f0, f1, w,
                       grad_out, d_f0);
    GOM_LAUNCH_CHECK();
    return 0;
}
