// Frame-parallel training step, the part behind the gradient: Adam on the FLAT parameter buffer (this file) and the direct all-reduce
// over peer pointers (below).  SURVEY.md 8(e): one process per GPU renders its own frame(s); the only exchange is the sum of the
// flat fp32 gradient buffer (951 023 floats at 55 104 Gaussians), after which every rank applies the same optimizer step.
//
// The reference's optimizer is torch.optim.Adam(param_groups, betas=(0.9, 0.999)) with one learning rate per group
// (train.py:263-267, models/model.py:305-327, update_lr train.py:166-175); through torch that is ~10 multi-tensor launches and
// ~100 us of host time per step -- half of a single-frame step (0.2 ms) of this path.  Here it is ONE launch over the flat buffer:
// 16 bytes read and 12 written per parameter, HBM-bound (951 023 parameters: 26.6 MB, ~6 us).
#include "gom_internal.h"
#include <math.h>

namespace {

struct AdamSegs {
    uint32_t begin[GOM_ADAM_MAX_SEGMENTS + 1];   // segment i = [begin[i], begin[i + 1]) of the flat buffer
    float step_size[GOM_ADAM_MAX_SEGMENTS];      // lr_i / (1 - beta1^t)
    int n;
};

// torch.optim.Adam (no weight decay, no amsgrad, maximize = False), the arithmetic of torch/optim/adam.py::_single_tensor_adam:
//   m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g g;  p -= step_size * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
__global__ void __launch_bounds__(256) k_adam_flat(uint32_t n, float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, AdamSegs segs, float beta1, float beta2, float eps, float inv_sqrt_bc2,
                                                   float grad_scale) {
    // 4 consecutive parameters per thread and trip (the buffers come from hipMalloc / torch: 16-byte aligned); the ragged end one by one
    const uint32_t n4 = n >> 2;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n4 + (n & 3u); i += gridDim.x * 256) {
        const bool vec = i < n4;
        const uint32_t e0 = vec ? 4u * i : 4u * n4 + (i - n4);
        const int cnt = vec ? 4 : 1;
        float pp[4], gg[4], mm[4], vv[4];
        if (vec) {
            const float4 a = reinterpret_cast<const float4 *>(p)[i], b = reinterpret_cast<const float4 *>(g)[i];
            const float4 c = reinterpret_cast<const float4 *>(m)[i], d = reinterpret_cast<const float4 *>(v)[i];
            pp[0] = a.x; pp[1] = a.y; pp[2] = a.z; pp[3] = a.w; gg[0] = b.x; gg[1] = b.y; gg[2] = b.z; gg[3] = b.w;
            mm[0] = c.x; mm[1] = c.y; mm[2] = c.z; mm[3] = c.w; vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
        } else {
            pp[0] = p[e0]; gg[0] = g[e0]; mm[0] = m[e0]; vv[0] = v[e0];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (u >= cnt) break;
            const uint32_t e = e0 + (uint32_t)u;
            float ss = 0.f;                                       // an element outside every segment (padding of the payload) is not a parameter: untouched
            bool in = false;
#pragma unroll
            for (int s = 0; s < GOM_ADAM_MAX_SEGMENTS; s++)
                if (s < segs.n && e >= segs.begin[s] && e < segs.begin[s + 1]) { ss = segs.step_size[s]; in = true; }
            if (!in) continue;
            const float gr = gg[u] * grad_scale;
            mm[u] = __fmaf_rn(beta1, mm[u], __fmul_rn(1.f - beta1, gr));              // exp_avg.lerp_(grad, 1 - beta1) up to rounding
            vv[u] = __fmaf_rn(beta2, vv[u], __fmul_rn(__fmul_rn(1.f - beta2, gr), gr));
            const float denom = __fadd_rn(__fmul_rn(__fsqrt_rn(vv[u]), inv_sqrt_bc2), eps);
            pp[u] = __fsub_rn(pp[u], __fmul_rn(ss, __fdiv_rn(mm[u], denom)));
        }
        if (vec) {
            reinterpret_cast<float4 *>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
            reinterpret_cast<float4 *>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
            reinterpret_cast<float4 *>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
            p[e0] = pp[0]; m[e0] = mm[0]; v[e0] = vv[0];
        }
    }
}

}  // namespace

extern "C" int gom_adam_flat(int64_t n, float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int32_t n_segments,
                             const int64_t *seg_begin, const float *seg_lr, int64_t step, float beta1, float beta2, float eps, float grad_scale,
                             void *stream) {
    if (n < 0 || n > 0xffffffffLL) { gom_set_error("gom_adam_flat: bad size"); return -1; }
    if (n_segments < 1 || n_segments > GOM_ADAM_MAX_SEGMENTS || !seg_begin || !seg_lr) { gom_set_error("gom_adam_flat: 1..%d segments", GOM_ADAM_MAX_SEGMENTS); return -1; }
    if (step < 1) { gom_set_error("gom_adam_flat: step counts from 1"); return -1; }
    if (n == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq) { gom_set_error("gom_adam_flat: null pointer"); return -1; }
    AdamSegs segs{};
    segs.n = n_segments;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    for (int i = 0; i <= n_segments; i++) {
        if (seg_begin[i] < 0 || seg_begin[i] > n || (i > 0 && seg_begin[i] < seg_begin[i - 1])) { gom_set_error("gom_adam_flat: segment bounds must ascend inside [0, n]"); return -1; }
        segs.begin[i] = (uint32_t)seg_begin[i];
    }
    for (int i = 0; i < n_segments; i++) segs.step_size[i] = (float)((double)seg_lr[i] / bc1);
    const uint32_t work = (uint32_t)(n >> 2) + (uint32_t)(n & 3);
    const unsigned blocks = (unsigned)((work + 255) / 256);
    hipLaunchKernelGGL(k_adam_flat, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, (hipStream_t)stream, (uint32_t)n, params, grads, exp_avg, exp_avg_sq, segs,
                       beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
    GOM_LAUNCH_CHECK();
    return 0;
}
