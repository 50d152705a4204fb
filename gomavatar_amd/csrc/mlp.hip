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

// ---------------------------------------------------------------------------------------------------------------------
// The shadow MLP itself (models/modules/shadow_module.py:66-117 at its default shape: D0 -> H -> H -> H -> 1, ReLU, sigmoid; no
// skip connection inside depth 3), forward and input-gradient chain as ONE kernel each.  The layers are 0.9 GFLOP per frame: what
// they cost through the BLAS library is launches (4 GEMMs + 4 activations forward, 3 GEMMs + 4 activation derivatives backward,
// ~50 us of host time per GEMM call).  A workgroup takes 32 rows through all layers; activations live in LDS as [feature][row]
// (a thread = one output feature x 16 rows reads 4 rows per ds_read_b128), weights stream through LDS 32 input features at a time.
// Saved for the backward: the three hidden activations (post-ReLU) and the output; the backward writes dz of every layer for
// gom_linear_wgrad and the gradient w.r.t. the input.
namespace {

constexpr int kTR = 32;     // rows per workgroup
constexpr int kHW = 128;    // widest layer supported

// out[o][rows] = act( b[o] + sum_i W[o][i] src[i][rows] ),  thread = (o = tid & 127, rows 16 (tid >> 7) .. +15)
// W row-major [out_dim][in_dim] (nn.Linear).  TRANS = false: weights used as W[o][i] (forward);  TRANS = true: computes
// out[i][rows] = sum_o W[o][i] src[o][rows] (backward through the layer), thread = (i, row half).
template <bool TRANS>
__device__ __forceinline__ void mlp_layer(int n_red, int n_out, int ld, const float *__restrict__ W, float (*s_w)[kHW], const float (*s_src)[kTR],
                                          float (&acc)[16]) {
    const int tid = threadIdx.x, c = tid & 127, rh = tid >> 7;
    for (int rc = 0; rc < n_red; rc += 32) {
        __syncthreads();   // previous chunk's readers are done
        // stage a 32 x 128 block of weights as s_w[reduction index][output index]
        if (!TRANS) {      // s_w[ii][o] = W[o][rc + ii]: thread (o = c, 16 consecutive ii)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int ii = 16 * rh + k;
                s_w[ii][c] = (c < n_out && rc + ii < n_red) ? W[(size_t)c * ld + rc + ii] : 0.f;
            }
        } else {           // s_w[oo][i] = W[rc + oo][i]: rows of W are contiguous in i
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int oo = 16 * rh + k;
                s_w[oo][c] = (c < n_out && rc + oo < n_red) ? W[(size_t)(rc + oo) * ld + c] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int ii = 0; ii < 32; ii++) {
            const float w = s_w[ii][c];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 v = *reinterpret_cast<const float4 *>(&s_src[rc + ii][16 * rh + 4 * j]);
                acc[4 * j] += w * v.x; acc[4 * j + 1] += w * v.y; acc[4 * j + 2] += w * v.z; acc[4 * j + 3] += w * v.w;
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_mlp3_fwd(int64_t n, int D0, int H, const float *__restrict__ x, const float *__restrict__ W1,
                                                  const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ b2,
                                                  const float *__restrict__ W3, const float *__restrict__ b3, const float *__restrict__ w4,
                                                  const float *__restrict__ b4, float *__restrict__ h1, float *__restrict__ h2,
                                                  float *__restrict__ h3, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float s_a[kHW][kTR], s_b[kHW][kTR];
    __shared__ float s_w[32][kHW];
    const int tid = threadIdx.x, c = tid & 127, rh = tid >> 7;
    const int64_t r0 = (int64_t)blockIdx.x * kTR;
    for (int idx = tid; idx < kHW * kTR; idx += 256) {   // input rows -> s_a[feature][row] (zero padded)
        const int rr = idx / kHW, i = idx % kHW;          // consecutive threads read consecutive features of one row
        s_a[i][rr] = (r0 + rr < n && i < D0) ? x[(r0 + rr) * D0 + i] : 0.f;
    }
    float(*src)[kTR] = s_a;
    float(*dst)[kTR] = s_b;
    const float *Ws[3] = {W1, W2, W3}, *bs[3] = {b1, b2, b3};
    float *hs[3] = {h1, h2, h3};
#pragma unroll
    for (int l = 0; l < 3; l++) {
        const int in_dim = l == 0 ? D0 : H;
        float acc[16];
        const float bias = c < H ? bs[l][c] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) acc[k] = bias;
        mlp_layer<false>(in_dim, H, in_dim, Ws[l], s_w, src, acc);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            acc[k] = c < H ? fmaxf(acc[k], 0.f) : 0.f;
            const int64_t r = r0 + 16 * rh + k;
            if (r < n && c < H) hs[l][r * H + c] = acc[k];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<float4 *>(&dst[c][16 * rh + 4 * j]) = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
        float(*t)[kTR] = src; src = dst; dst = t;
    }
    __syncthreads();
    if (tid < kTR && r0 + tid < n) {   // output layer: one thread per row
        float s = b4[0];
        for (int o = 0; o < H; o++) s += src[o][tid] * w4[o];
        out[r0 + tid] = 1.f / (1.f + __expf(-s));
    }
}

// g [n] = dL/d out  ->  dz4 [n], dz3 / dz2 / dz1 [n][H], dx [n][D0]
__global__ void __launch_bounds__(256) k_mlp3_bwd(int64_t n, int D0, int H, const float *__restrict__ g, const float *__restrict__ out,
                                                  const float *__restrict__ h1, const float *__restrict__ h2, const float *__restrict__ h3,
                                                  const float *__restrict__ W1, const float *__restrict__ W2, const float *__restrict__ W3,
                                                  const float *__restrict__ w4, float *__restrict__ dz4, float *__restrict__ dz3,
                                                  float *__restrict__ dz2, float *__restrict__ dz1, float *__restrict__ dx) {
    __shared__ __attribute__((aligned(16))) float s_a[kHW][kTR], s_b[kHW][kTR];
    __shared__ float s_w[32][kHW];
    __shared__ float s_d4[kTR];
    const int tid = threadIdx.x, c = tid & 127, rh = tid >> 7;
    const int64_t r0 = (int64_t)blockIdx.x * kTR;
    if (tid < kTR) {
        const int64_t r = r0 + tid;
        float d = 0.f;
        if (r < n) { const float o = out[r]; d = g[r] * o * (1.f - o); dz4[r] = d; }
        s_d4[tid] = d;
    }
    __syncthreads();
    {   // dz3 = dz4 w4^T (.) [h3 > 0]
        const float w = c < H ? w4[c] : 0.f;
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int64_t r = r0 + 16 * rh + k;
            const bool on = r < n && c < H && h3[r * H + c] > 0.f;
            v[k] = on ? s_d4[16 * rh + k] * w : 0.f;
            if (r < n && c < H) dz3[r * H + c] = v[k];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<float4 *>(&s_a[c][16 * rh + 4 * j]) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    }
    float(*src)[kTR] = s_a;
    float(*dst)[kTR] = s_b;
    const float *Ws[3] = {W3, W2, W1};
    const float *hprev[3] = {h2, h1, nullptr};
    float *dzs[3] = {dz2, dz1, dx};
#pragma unroll
    for (int l = 0; l < 3; l++) {
        const int n_out = l == 2 ? D0 : H;   // width of the layer's INPUT side (what this step produces)
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; k++) acc[k] = 0.f;
        mlp_layer<true>(H, n_out, n_out, Ws[l], s_w, src, acc);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int64_t r = r0 + 16 * rh + k;
            const bool live = r < n && c < n_out;
            if (l < 2) acc[k] = (live && hprev[l][r * H + c] > 0.f) ? acc[k] : 0.f;
            if (live) dzs[l][r * n_out + c] = acc[k];
        }
        if (l < 2) {
#pragma unroll
            for (int j = 0; j < 4; j++) *reinterpret_cast<float4 *>(&dst[c][16 * rh + 4 * j]) = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
            float(*t)[kTR] = src; src = dst; dst = t;
        }
    }
}

}  // namespace

extern "C" int gom_mlp3_forward(int64_t n, int D0, int H, const float *x, const float *W1, const float *b1, const float *W2, const float *b2,
                                const float *W3, const float *b3, const float *w4, const float *b4, float *h1, float *h2, float *h3, float *out,
                                void *stream) {
    if (n < 0 || D0 < 1 || D0 > kHW || H < 1 || H > kHW) { gom_set_error("gom_mlp3_forward: widths must be in 1..128"); return -1; }
    if (n == 0) return 0;
    if (!x || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !w4 || !b4 || !h1 || !h2 || !h3 || !out) { gom_set_error("gom_mlp3_forward: null pointer"); return -1; }
    hipLaunchKernelGGL(k_mlp3_fwd, dim3((unsigned)((n + kTR - 1) / kTR)), dim3(256), 0, (hipStream_t)stream, n, D0, H, x, W1, b1, W2, b2, W3, b3, w4, b4, h1, h2,
                       h3, out);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_mlp3_backward(int64_t n, int D0, int H, const float *g, const float *out, const float *h1, const float *h2, const float *h3,
                                 const float *W1, const float *W2, const float *W3, const float *w4, float *dz4, float *dz3, float *dz2, float *dz1,
                                 float *dx, void *stream) {
    if (n < 0 || D0 < 1 || D0 > kHW || H < 1 || H > kHW) { gom_set_error("gom_mlp3_backward: widths must be in 1..128"); return -1; }
    if (n == 0) return 0;
    if (!g || !out || !h1 || !h2 || !h3 || !W1 || !W2 || !W3 || !w4 || !dz4 || !dz3 || !dz2 || !dz1 || !dx) { gom_set_error("gom_mlp3_backward: null pointer"); return -1; }
    hipLaunchKernelGGL(k_mlp3_bwd, dim3((unsigned)((n + kTR - 1) / kTR)), dim3(256), 0, (hipStream_t)stream, n, D0, H, g, out, h1, h2, h3, W1, W2, W3, w4, dz4,
                       dz3, dz2, dz1, dx);
    GOM_LAUNCH_CHECK();
    return 0;
}
