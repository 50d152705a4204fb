// Positional encoding of the shadow MLP's input (reference models/modules/shadow_module.py:96-97 through
// utils/network_util.py get_embedder: [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], input included,
// log-sampled frequencies) as one kernel forward and one backward instead of ~25 + ~35 elementwise launches per training
// iteration: the whole `Model` iteration is launch-bound, not bandwidth-bound.
#include "gom_internal.h"

namespace {

// x [n][3] -> out [n][3 + 6 L];  one thread per (row, frequency slot): slot 0 copies x, slot 1 + l writes sin / cos of 2^l x
__global__ void __launch_bounds__(256) k_posenc_fwd(size_t n, int L, const float *__restrict__ x, float *__restrict__ out) {
    const size_t total = n * (size_t)(L + 1);
    const int D = 3 + 6 * L;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / (L + 1);
        const int s = (int)(i % (L + 1));
        const float v[3] = {x[3 * r], x[3 * r + 1], x[3 * r + 2]};
        float *o = out + r * D;
        if (s == 0) {
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
        } else {
            const float f = (float)(1u << (s - 1));   // 2^l: the product is exact, like torch's x * freq
            float *q = o + 3 + 6 * (s - 1);
#pragma unroll
            for (int c = 0; c < 3; c++) { q[c] = sinf(v[c] * f); q[3 + c] = cosf(v[c] * f); }
        }
    }
}

// dx = g_x + sum_l 2^l (cos(2^l x) g_sin - sin(2^l x) g_cos);  one thread per row (the sum is short)
__global__ void __launch_bounds__(256) k_posenc_bwd(size_t n, int L, const float *__restrict__ x, const float *__restrict__ g, float *__restrict__ dx) {
    const int D = 3 + 6 * L;
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (size_t)gridDim.x * 256) {
        const float *gr = g + r * D;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = x[3 * r + c];
            float acc = gr[c];
            for (int l = 0; l < L; l++) {
                const float f = (float)(1u << l);
                acc += f * (cosf(v * f) * gr[3 + 6 * l + c] - sinf(v * f) * gr[3 + 6 * l + 3 + c]);
            }
            dx[3 * r + c] = acc;
        }
    }
}

}  // namespace

extern "C" int gom_posenc_forward(int64_t n, int L, const float *x, float *out, void *stream) {
    if (n < 0 || L < 0 || L > 16 || (n > 0 && (!x || !out))) { gom_set_error("gom_posenc_forward: bad arguments"); return -1; }
    if (n == 0) return 0;
    const size_t total = (size_t)n * (L + 1);
    hipLaunchKernelGGL(k_posenc_fwd, dim3((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0, (hipStream_t)stream, (size_t)n, L, x, out);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_posenc_backward(int64_t n, int L, const float *x, const float *g_out, float *dx, void *stream) {
    if (n < 0 || L < 0 || L > 16 || (n > 0 && (!x || !g_out || !dx))) { gom_set_error("gom_posenc_backward: bad arguments"); return -1; }
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_posenc_bwd, dim3((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), dim3(256), 0, (hipStream_t)stream, (size_t)n, L, x, g_out, dx);
    GOM_LAUNCH_CHECK();
    return 0;
}
