// Fused photometric L1 losses with their backward, one pass over the image.
//
// Replaces unpack (train.py:53-55) + L1 rgb / L1 mask (train.py:101-111) and the
// autograd backward of those ~8 element-wise/reduction launches.  Reads the
// rasterizer's CHW output directly and writes dL/d(pred) in the same CHW layout
// the raster backward consumes.  HBM-bound: 4+1+3+1 floats read, 4(+1) written
// per pixel.  The loss VALUE is only needed for logging, so each block writes its
// partial sums and nobody waits on a same-address atomic (11-13 ns each on
// MI355X); the gradient does not depend on the value.
#include "gom_internal.h"

namespace {

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

__global__ void __launch_bounds__(256) k_l1_loss(int HW, const float *__restrict__ pred, const float *__restrict__ shade,
                                                 const float *__restrict__ gt_rgb, const float *__restrict__ gt_mask,
                                                 const float *__restrict__ bg, float k_rgb, float k_mask,
                                                 float *__restrict__ dpred, float *__restrict__ dshade, float *__restrict__ partials) {
    __shared__ float s_red[2][4];
    {  // blockIdx.y = frame of a batched launch: [B][4][HW] images, [B][HW][3] targets, [B][3] backgrounds
        const size_t fr = blockIdx.y;
        pred += fr * 4 * HW; gt_rgb += fr * 3 * HW; gt_mask += fr * HW; bg += fr * 3; dpred += fr * 4 * HW;
        partials += fr * 2 * gridDim.x;
        if (shade) shade += fr * HW;
        if (dshade) dshade += fr * HW;
    }
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    float sum_rgb = 0.f, sum_mask = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const float m = pred[3 * (size_t)HW + p];
        const float s = shade ? shade[p] : 1.f;
        const float a0 = pred[p], a1 = pred[(size_t)HW + p], a2 = pred[2 * (size_t)HW + p];
        const float r0 = a0 * s * m + b0 * (1.f - m) - gt_rgb[3 * (size_t)p];
        const float r1 = a1 * s * m + b1 * (1.f - m) - gt_rgb[3 * (size_t)p + 1];
        const float r2 = a2 * s * m + b2 * (1.f - m) - gt_rgb[3 * (size_t)p + 2];
        const float rm = m - gt_mask[p];
        sum_rgb += fabsf(r0) + fabsf(r1) + fabsf(r2);
        sum_mask += fabsf(rm);
        const float s0 = sgn(r0) * k_rgb, s1 = sgn(r1) * k_rgb, s2 = sgn(r2) * k_rgb;
        dpred[p] = s0 * s * m;
        dpred[(size_t)HW + p] = s1 * s * m;
        dpred[2 * (size_t)HW + p] = s2 * s * m;
        dpred[3 * (size_t)HW + p] = s0 * (a0 * s - b0) + s1 * (a1 * s - b1) + s2 * (a2 * s - b2) + sgn(rm) * k_mask;
        if (dshade) dshade[p] = (s0 * a0 + s1 * a1 + s2 * a2) * m;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sum_rgb += __shfl_xor(sum_rgb, d, 64);
        sum_mask += __shfl_xor(sum_mask, d, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_red[0][wave] = sum_rgb; s_red[1][wave] = sum_mask; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
        partials[2 * blockIdx.x + 1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
    }
}

}  // namespace

int gom_l1_loss_batch(int B, int H, int W, const float *pred, const float *shade, const float *gt_rgb, const float *gt_mask,
                      const float *bg, float c_rgb, float c_mask, float grad_scale, float *dL_dpred, float *dL_dshade,
                      float *loss_partials, void *stream) {
    if (H <= 0 || W <= 0) { gom_set_error("gom_l1_loss: bad image size"); return -1; }
    if (!pred || !gt_rgb || !gt_mask || !bg || !dL_dpred || !loss_partials) { gom_set_error("gom_l1_loss: null pointer"); return -1; }
    const int HW = H * W;
    const float k_rgb = grad_scale * c_rgb / (3.0f * (float)HW);
    const float k_mask = grad_scale * c_mask / (float)HW;
    hipLaunchKernelGGL(k_l1_loss, dim3(GOM_LOSS_BLOCKS, B), dim3(256), 0, (hipStream_t)stream, HW, pred, shade, gt_rgb, gt_mask, bg,
                       k_rgb, k_mask, dL_dpred, dL_dshade, loss_partials);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_l1_loss(int H, int W, const float *pred, const float *shade, const float *gt_rgb, const float *gt_mask,
                           const float *bg, float c_rgb, float c_mask, float grad_scale, float *dL_dpred, float *dL_dshade,
                           float *loss_partials, void *stream) {
    return gom_l1_loss_batch(1, H, W, pred, shade, gt_rgb, gt_mask, bg, c_rgb, c_mask, grad_scale, dL_dpred, dL_dshade, loss_partials,
                             stream);
}
