// Weight / bias gradient of a Linear layer whose batch dimension is the long one (the shadow MLP of model.py:279-287 sees one
// row per pixel under the mesh, ~24 000 at 512x512, and is 39 -> 128 -> 128 -> 128 -> 1 wide):
//     dW[o][i] = sum_r dY[r][o] X[r][i],   db[o] = sum_r dY[r][o]
// is a GEMM with M, N <= 128 and K = rows.  The BLAS library tiles M x N only (16 workgroups for 128 x 128) and walks K
// serially: 70-80 us per layer, 0.3 ms per training iteration.  Here the ROWS are split over the workgroups, every workgroup
// accumulates a 64 x 64 block of out x in in registers (thread = 4 x 4 sub-block, operands staged through LDS 32 rows at a time) and a
// second kernel adds the per-workgroup partials in a fixed order (no atomics: reproducible).
#include "gom_internal.h"

namespace {

constexpr int kRows = 32;   // rows staged per trip

// grid = (row slices, out blocks of 64, in blocks of 64); block = 256 threads = 16 x 16 sub-blocks of 4 x 4
__global__ void __launch_bounds__(256) k_linear_wgrad_partial(int64_t n, int in_dim, int out_dim, const float *__restrict__ X, const float *__restrict__ dY,
                                                              float *__restrict__ partial /* [slices][129][128]: row 128 = bias */) {
    __shared__ __attribute__((aligned(16))) float s_x[kRows][64], s_y[kRows][64];
    const int tid = threadIdx.x, to = tid >> 4, ti = tid & 15;
    const int o0 = blockIdx.y * 64, i0 = blockIdx.z * 64;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r_lo = (int64_t)blockIdx.x * per, r_hi = r_lo + per < n ? r_lo + per : n;
    float acc[4][4], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0.f;
    // software pipeline: the next trip's rows are already on their way (registers) while this trip's 32 rows are multiplied
    constexpr int NL = kRows * 64 / 256;
    float px[NL], py[NL];
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int idx = tid + 256 * k, rr = idx >> 6, c = idx & 63;
            const int64_t r = r0 + rr;
            px[k] = (r < r_hi && i0 + c < in_dim) ? X[r * in_dim + i0 + c] : 0.f;
            py[k] = (r < r_hi && o0 + c < out_dim) ? dY[r * out_dim + o0 + c] : 0.f;
        }
    };
    if (r_lo < r_hi) fetch(r_lo);
    for (int64_t r0 = r_lo; r0 < r_hi; r0 += kRows) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int idx = tid + 256 * k;
            s_x[idx >> 6][idx & 63] = px[k];
            s_y[idx >> 6][idx & 63] = py[k];
        }
        __syncthreads();
        if (r0 + kRows < r_hi) fetch(r0 + kRows);
#pragma unroll 8
        for (int rr = 0; rr < kRows; rr++) {
            const float4 xv = *reinterpret_cast<const float4 *>(&s_x[rr][4 * ti]);
            const float4 yv = *reinterpret_cast<const float4 *>(&s_y[rr][4 * to]);
            const float x4[4] = {xv.x, xv.y, xv.z, xv.w}, y4[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int a = 0; a < 4; a++) {
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] += y4[a] * x4[b];
                accb[a] += y4[a];
            }
        }
    }
    float *dst = partial + (size_t)blockIdx.x * 129 * 128;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        *reinterpret_cast<float4 *>(dst + (size_t)(o0 + 4 * to + a) * 128 + i0 + 4 * ti) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
        if (ti == 0 && blockIdx.z == 0) dst[128 * 128 + o0 + 4 * to + a] = accb[a];
    }
}

__global__ void __launch_bounds__(256) k_linear_wgrad_reduce(int slices, int in_dim, int out_dim, const float *__restrict__ partial, float *__restrict__ dW,
                                                             float *__restrict__ db) {
    const int idx = blockIdx.x * 256 + threadIdx.x;   // over 129 x 128
    if (idx >= 129 * 128) return;
    const int o = idx >> 7, i = idx & 127;
    const bool is_bias = o == 128;
    if (is_bias ? (i >= out_dim || !db) : (o >= out_dim || i >= in_dim)) return;
    float s = 0.f;
    for (int k0 = 0; k0 < slices; k0 += 16) {   // 16 loads in flight, added in slice order
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = k0 + u < slices ? partial[(size_t)(k0 + u) * 129 * 128 + idx] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; u++) s += v[u];
    }
    if (is_bias) db[i] = s;
    else dW[(size_t)o * in_dim + i] = s;
}

}  // namespace

extern "C" int gom_linear_wgrad_slices(void) { return 128; }

extern "C" int gom_linear_wgrad(int64_t n, int in_dim, int out_dim, const float *X, const float *dY, float *dW, float *db, float *workspace, void *stream) {
    if (n <= 0 || in_dim <= 0 || out_dim <= 0 || in_dim > 128 || out_dim > 128) { gom_set_error("gom_linear_wgrad: in_dim and out_dim must be in 1..128"); return -1; }
    if (!X || !dY || !dW || !workspace) { gom_set_error("gom_linear_wgrad: null pointer"); return -1; }
    const int slices = gom_linear_wgrad_slices();
    hipLaunchKernelGGL(k_linear_wgrad_partial, dim3(slices, (out_dim + 63) / 64, (in_dim + 63) / 64), dim3(256), 0, (hipStream_t)stream, n, in_dim, out_dim, X, dY,
                       workspace);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_linear_wgrad_reduce, dim3((129 * 128 + 255) / 256), dim3(256), 0, (hipStream_t)stream, slices, in_dim, out_dim, workspace, dW, db);
    GOM_LAUNCH_CHECK();
    return 0;
}
