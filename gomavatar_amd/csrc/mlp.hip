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

// one workgroup: row slice blockIdx.x of gridDim.x, the 64 x 64 block (o0, i0) of out x in; 256 threads = 16 x 16 sub-blocks of 4 x 4
__device__ __forceinline__ void wgrad_partial_block(int64_t n, int in_dim, int out_dim, const float *__restrict__ X, const float *__restrict__ dY,
                                                    float *__restrict__ partial /* [slices][129][128]: row 128 = bias */, int o0, int i0) {
    __shared__ __attribute__((aligned(16))) float s_x[kRows][64], s_y[kRows][64];   // (n may come from device memory: gom_shade_*, below)
    const int tid = threadIdx.x, to = tid >> 4, ti = tid & 15;
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
        if (ti == 0 && i0 == 0) dst[128 * 128 + o0 + 4 * to + a] = accb[a];
    }
}

// grid = (row slices, out blocks of 64, in blocks of 64)
__global__ void __launch_bounds__(256) k_linear_wgrad_partial(int64_t n, int in_dim, int out_dim, const float *__restrict__ X, const float *__restrict__ dY,
                                                              float *__restrict__ partial) {
    wgrad_partial_block(n, in_dim, out_dim, X, dY, partial, blockIdx.y * 64, blockIdx.z * 64);
}

// The four layers of the shadow MLP in one launch: grid = (row slices, 2 out blocks, 4 layers x 2 in blocks); blocks beyond a
// layer's width return at once.  Four launches of 512 workgroups each left the chip half empty four times over.
struct WgradLayers {
    const float *X[4], *dY[4];
    float *dW[4], *db[4];
    int in_dim[4], out_dim[4];
};
__global__ void __launch_bounds__(256) k_mlp3_wgrad_partial(int64_t n, WgradLayers L, float *__restrict__ partial, int slices, const int32_t *__restrict__ n_dev) {
    if (n_dev) n = (int64_t)n_dev[0] + 1;   // rows under the mesh + the background row (gom_shade_*)
    const int layer = blockIdx.z >> 1, o0 = blockIdx.y * 64, i0 = (blockIdx.z & 1) * 64;
    if (o0 >= L.out_dim[layer] || i0 >= L.in_dim[layer]) return;
    wgrad_partial_block(n, L.in_dim[layer], L.out_dim[layer], L.X[layer], L.dY[layer], partial + (size_t)layer * slices * 129 * 128, o0, i0);
}

__device__ __forceinline__ void wgrad_reduce_block(int slices, int in_dim, int out_dim, const float *__restrict__ partial, float *__restrict__ dW,
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
__global__ void __launch_bounds__(256) k_linear_wgrad_reduce(int slices, int in_dim, int out_dim, const float *__restrict__ partial, float *__restrict__ dW,
                                                             float *__restrict__ db) {
    wgrad_reduce_block(slices, in_dim, out_dim, partial, dW, db);
}
__global__ void __launch_bounds__(256) k_mlp3_wgrad_reduce(int slices, WgradLayers L, const float *__restrict__ partial) {
    const int layer = blockIdx.y;
    wgrad_reduce_block(slices, L.in_dim[layer], L.out_dim[layer], partial + (size_t)layer * slices * 129 * 128, L.dW[layer], L.db[layer]);
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
// skip connection inside depth 3), forward and input-gradient chain as ONE kernel each.  The layers are 1.8 GFLOP per frame: what
// they cost through the BLAS library is launches (4 GEMMs + 4 activations forward, 3 GEMMs + 4 activation derivatives backward,
// ~50 us of host time per GEMM call).  A workgroup takes 64 rows through all layers; the activations live in LDS as
// [feature][row] and are overwritten in place layer by layer, the weights stream through LDS 32 reduction indices at a time.
// A thread owns a 4 feature x 8 row register tile: per reduction index one ds_read_b128 of weights and two (wave-broadcast) of
// rows feed 32 FMAs issued as 16 v_pk_fma_f32, which keeps the LDS return bus (8 clk per b128 per CU) under the VALU time.
// Saved for the backward: the three hidden activations (post-ReLU) and the output; the backward writes dz of every layer for
// gom_linear_wgrad and the gradient w.r.t. the input.
namespace {

constexpr int kTR = 64;          // rows per workgroup
constexpr int kTRP = kTR + 4;    // padded row stride (keeps 16-byte alignment, spreads the tile write-back over the banks)
constexpr int kHW = 128;         // widest layer supported
constexpr int kWP = kHW + 4;     // row stride of the staged weights: lets the forward's transposing write-in spread over the banks
typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc[j][k] += sum_red Wt[red][4 cg + j] * src[red][8 rg + k]   (cg = tid & 31, rg = tid >> 5)
// TRANS = false: Wt[red][o] = W[o][red] (forward, W row-major [n_out][ld]);  TRANS = true: Wt[red][i] = W[red][i] (backward).
template <bool TRANS>
__device__ __forceinline__ void mlp_layer(int n_red, int n_out, int ld, const float *__restrict__ W, float (*s_w)[kWP], const float (*s_src)[kTRP],
                                          f32x2 (&acc)[4][4]) {
    const int tid = threadIdx.x, c = tid & 127, half = tid >> 7, cg = tid & 31, rg = tid >> 5;
    float wn[16];   // the next 32 x 128 block of weights, in flight while the current one is being used
    // forward: W[o][rc + ii] is contiguous in ii, so lanes run along ii (o = o0 + 8 k);  backward: W[rc + ii][i] is contiguous in i
    const int f_ii = TRANS ? 16 * half : (tid & 31), f_c = TRANS ? c : (tid >> 5);
    auto fetch = [&](int rc) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int ii = TRANS ? f_ii + k : f_ii, cc = TRANS ? f_c : f_c + 8 * k;
            wn[k] = 0.f;
            if (cc < n_out && rc + ii < n_red) wn[k] = TRANS ? W[(size_t)(rc + ii) * ld + cc] : W[(size_t)cc * ld + rc + ii];
        }
    };
    fetch(0);
    for (int rc = 0; rc < n_red; rc += 32) {
        __syncthreads();   // previous chunk's readers are done (and the activations written before the call are visible)
#pragma unroll
        for (int k = 0; k < 16; k++) s_w[TRANS ? f_ii + k : f_ii][TRANS ? f_c : f_c + 8 * k] = wn[k];
        __syncthreads();
        if (rc + 32 < n_red) fetch(rc + 32);
        const int lim = min(32, n_red - rc);
#pragma unroll 4
        for (int ii = 0; ii < lim; ii++) {
            const float4 w = *reinterpret_cast<const float4 *>(&s_w[ii][4 * cg]);
            const float4 va = *reinterpret_cast<const float4 *>(&s_src[rc + ii][8 * rg]);
            const float4 vb = *reinterpret_cast<const float4 *>(&s_src[rc + ii][8 * rg + 4]);
            const f32x2 v[4] = {{va.x, va.y}, {va.z, va.w}, {vb.x, vb.y}, {vb.z, vb.w}};
            const float wj[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x2 ww = {wj[j], wj[j]};
#pragma unroll
                for (int q = 0; q < 4; q++) acc[j][q] = __builtin_elementwise_fma(ww, v[q], acc[j][q]);
            }
        }
    }
}

__device__ __forceinline__ void tile_to_lds(float (*s)[kTRP], const f32x2 (&acc)[4][4]) {
    const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        *reinterpret_cast<float4 *>(&s[4 * cg + j][8 * rg]) = make_float4(acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y);
        *reinterpret_cast<float4 *>(&s[4 * cg + j][8 * rg + 4]) = make_float4(acc[j][2].x, acc[j][2].y, acc[j][3].x, acc[j][3].y);
    }
}
__device__ __forceinline__ float tile_get(const f32x2 (&acc)[4][4], int j, int k) { return (k & 1) ? acc[j][k >> 1].y : acc[j][k >> 1].x; }
__device__ __forceinline__ void tile_set(f32x2 (&acc)[4][4], int j, int k, float v) { if (k & 1) acc[j][k >> 1].y = v; else acc[j][k >> 1].x = v; }

__global__ void __launch_bounds__(256) k_mlp3_fwd(int64_t n, int D0, int H, const float *__restrict__ x, const float *__restrict__ W1,
                                                  const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ b2,
                                                  const float *__restrict__ W3, const float *__restrict__ b3, const float *__restrict__ w4,
                                                  const float *__restrict__ b4, float *__restrict__ h1, float *__restrict__ h2,
                                                  float *__restrict__ h3, float *__restrict__ out, const int32_t *__restrict__ n_dev) {
    __shared__ __attribute__((aligned(16))) float s_a[kHW][kTRP];
    __shared__ __attribute__((aligned(16))) float s_w[32][kWP];
    const int tid = threadIdx.x, cg = tid & 31, rg = tid >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * kTR;
    if (n_dev) { n = (int64_t)n_dev[0] + 1; if (r0 >= n) return; }   // row count in device memory (gom_shade_*): the grid covers the capacity
    const int rows = (int)min<int64_t>(kTR, n - r0);
    for (int idx = tid; idx < kTR * D0; idx += 256) {   // the workgroup's rows are one contiguous piece of x
        const int rr = idx / D0, i = idx - rr * D0;
        s_a[i][rr] = rr < rows ? x[r0 * D0 + idx] : 0.f;
    }
    const float *Ws[3] = {W1, W2, W3}, *bs[3] = {b1, b2, b3};
    float *hs[3] = {h1, h2, h3};
    const bool cols = 4 * cg < H;    // H % 4 == 0
#pragma unroll
    for (int l = 0; l < 3; l++) {
        const int in_dim = l == 0 ? D0 : H;
        f32x2 acc[4][4];
        const float4 bias = cols ? *reinterpret_cast<const float4 *>(bs[l] + 4 * cg) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float bj[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc[j][q] = f32x2{bj[j], bj[j]};
        mlp_layer<false>(in_dim, H, in_dim, Ws[l], s_w, s_a, acc);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc[j][q] = __builtin_elementwise_max(acc[j][q], f32x2{0.f, 0.f});
        if (cols) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (8 * rg + k < rows)
                    *reinterpret_cast<float4 *>(hs[l] + (r0 + 8 * rg + k) * H + 4 * cg) =
                        make_float4(tile_get(acc, 0, k), tile_get(acc, 1, k), tile_get(acc, 2, k), tile_get(acc, 3, k));
        }
        __syncthreads();   // every wave is done reading the layer's input
        tile_to_lds(s_a, acc);
    }
    __syncthreads();
    {   // output layer: four threads per row, 32 features each
        __shared__ float s_part[4][kTR];
        const int row = tid & (kTR - 1), part = tid >> 6;
        float s = 0.f;
#pragma unroll 8
        for (int o = 32 * part; o < min(H, 32 * part + 32); o++) s += s_a[o][row] * w4[o];
        s_part[part][row] = s;
        __syncthreads();
        if (tid < rows) {
            s = b4[0] + ((s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]));
            out[r0 + tid] = 1.f / (1.f + __expf(-s));
        }
    }
}

// g [n] = dL/d out  ->  dz4 [n], dz3 / dz2 / dz1 [n][H], dx [n][D0]
__global__ void __launch_bounds__(256) k_mlp3_bwd(int64_t n, int D0, int H, const float *__restrict__ g, const float *__restrict__ out,
                                                  const float *__restrict__ h1, const float *__restrict__ h2, const float *__restrict__ h3,
                                                  const float *__restrict__ W1, const float *__restrict__ W2, const float *__restrict__ W3,
                                                  const float *__restrict__ w4, float *__restrict__ dz4, float *__restrict__ dz3,
                                                  float *__restrict__ dz2, float *__restrict__ dz1, float *__restrict__ dx, const int32_t *__restrict__ n_dev) {
    __shared__ __attribute__((aligned(16))) float s_a[kHW][kTRP];
    __shared__ __attribute__((aligned(16))) float s_w[32][kWP];
    __shared__ float s_d4[kTR];
    const int tid = threadIdx.x, cg = tid & 31, rg = tid >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * kTR;
    if (n_dev) { n = (int64_t)n_dev[0] + 1; if (r0 >= n) return; }
    const int rows = (int)min<int64_t>(kTR, n - r0);
    const bool cols = 4 * cg < H;
    if (tid < kTR) {
        float d = 0.f;
        if (tid < rows) { const float o = out[r0 + tid]; d = g[r0 + tid] * o * (1.f - o); dz4[r0 + tid] = d; }
        s_d4[tid] = d;
    }
    __syncthreads();
    {   // dz3 = dz4 w4^T (.) [h3 > 0]
        const float4 w = cols ? *reinterpret_cast<const float4 *>(w4 + 4 * cg) : make_float4(0.f, 0.f, 0.f, 0.f);
        f32x2 acc[4][4];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cols && 8 * rg + k < rows) {
                const float4 h = *reinterpret_cast<const float4 *>(h3 + (r0 + 8 * rg + k) * H + 4 * cg);
                const float d = s_d4[8 * rg + k];
                v = make_float4(h.x > 0.f ? d * w.x : 0.f, h.y > 0.f ? d * w.y : 0.f, h.z > 0.f ? d * w.z : 0.f, h.w > 0.f ? d * w.w : 0.f);
                *reinterpret_cast<float4 *>(dz3 + (r0 + 8 * rg + k) * H + 4 * cg) = v;
            }
            tile_set(acc, 0, k, v.x); tile_set(acc, 1, k, v.y); tile_set(acc, 2, k, v.z); tile_set(acc, 3, k, v.w);
        }
        tile_to_lds(s_a, acc);
    }
    const float *Ws[3] = {W3, W2, W1};
    const float *hprev[3] = {h2, h1, nullptr};
    float *dzs[3] = {dz2, dz1, dx};
#pragma unroll
    for (int l = 0; l < 3; l++) {
        const int n_out = l == 2 ? D0 : H;   // width of the layer's INPUT side (what this step produces)
        f32x2 acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc[j][q] = f32x2{0.f, 0.f};
        mlp_layer<true>(H, n_out, n_out, Ws[l], s_w, s_a, acc);
        if (l < 2) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cols && 8 * rg + k < rows) {
                    const float4 h = *reinterpret_cast<const float4 *>(hprev[l] + (r0 + 8 * rg + k) * H + 4 * cg);
                    v = make_float4(h.x > 0.f ? tile_get(acc, 0, k) : 0.f, h.y > 0.f ? tile_get(acc, 1, k) : 0.f, h.z > 0.f ? tile_get(acc, 2, k) : 0.f,
                                    h.w > 0.f ? tile_get(acc, 3, k) : 0.f);
                    *reinterpret_cast<float4 *>(dzs[l] + (r0 + 8 * rg + k) * H + 4 * cg) = v;
                }
                tile_set(acc, 0, k, v.x); tile_set(acc, 1, k, v.y); tile_set(acc, 2, k, v.z); tile_set(acc, 3, k, v.w);
            }
            __syncthreads();
            tile_to_lds(s_a, acc);
        } else {   // the input gradient: D0 need not be a multiple of 4
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (8 * rg + k < rows && 4 * cg + j < D0) dx[(r0 + 8 * rg + k) * D0 + 4 * cg + j] = tile_get(acc, j, k);
        }
    }
}

}  // namespace

static int mlp3_forward_impl(int64_t n, int D0, int H, const float *x, const float *W1, const float *b1, const float *W2, const float *b2,
                             const float *W3, const float *b3, const float *w4, const float *b4, float *h1, float *h2, float *h3, float *out,
                             const int32_t *n_dev, void *stream) {
    if (n < 0 || D0 < 1 || D0 > kHW || H < 1 || H > kHW) { gom_set_error("gom_mlp3_forward: widths must be in 1..128"); return -1; }
    if (H % 4) { gom_set_error("gom_mlp3_forward: the hidden width must be a multiple of 4"); return -1; }
    if (n == 0) return 0;
    if (!x || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !w4 || !b4 || !h1 || !h2 || !h3 || !out) { gom_set_error("gom_mlp3_forward: null pointer"); return -1; }
    hipLaunchKernelGGL(k_mlp3_fwd, dim3((unsigned)((n + kTR - 1) / kTR)), dim3(256), 0, (hipStream_t)stream, n, D0, H, x, W1, b1, W2, b2, W3, b3, w4, b4, h1, h2,
                       h3, out, n_dev);
    GOM_LAUNCH_CHECK();
    return 0;
}
extern "C" int gom_mlp3_forward(int64_t n, int D0, int H, const float *x, const float *W1, const float *b1, const float *W2, const float *b2,
                                const float *W3, const float *b3, const float *w4, const float *b4, float *h1, float *h2, float *h3, float *out,
                                void *stream) {
    return mlp3_forward_impl(n, D0, H, x, W1, b1, W2, b2, W3, b3, w4, b4, h1, h2, h3, out, nullptr, stream);
}

static int mlp3_backward_impl(int64_t n, int D0, int H, const float *g, const float *out, const float *h1, const float *h2, const float *h3,
                              const float *W1, const float *W2, const float *W3, const float *w4, float *dz4, float *dz3, float *dz2, float *dz1,
                              float *dx, const int32_t *n_dev, void *stream) {
    if (n < 0 || D0 < 1 || D0 > kHW || H < 1 || H > kHW) { gom_set_error("gom_mlp3_backward: widths must be in 1..128"); return -1; }
    if (H % 4) { gom_set_error("gom_mlp3_backward: the hidden width must be a multiple of 4"); return -1; }
    if (n == 0) return 0;
    if (!g || !out || !h1 || !h2 || !h3 || !W1 || !W2 || !W3 || !w4 || !dz4 || !dz3 || !dz2 || !dz1 || !dx) { gom_set_error("gom_mlp3_backward: null pointer"); return -1; }
    hipLaunchKernelGGL(k_mlp3_bwd, dim3((unsigned)((n + kTR - 1) / kTR)), dim3(256), 0, (hipStream_t)stream, n, D0, H, g, out, h1, h2, h3, W1, W2, W3, w4, dz4,
                       dz3, dz2, dz1, dx, n_dev);
    GOM_LAUNCH_CHECK();
    return 0;
}
extern "C" int gom_mlp3_backward(int64_t n, int D0, int H, const float *g, const float *out, const float *h1, const float *h2, const float *h3,
                                 const float *W1, const float *W2, const float *W3, const float *w4, float *dz4, float *dz3, float *dz2, float *dz1,
                                 float *dx, void *stream) {
    return mlp3_backward_impl(n, D0, H, g, out, h1, h2, h3, W1, W2, W3, w4, dz4, dz3, dz2, dz1, dx, nullptr, stream);
}

static int mlp3_wgrad_impl(int64_t n, int D0, int H, const float *x, const float *h1, const float *h2, const float *h3, const float *dz1, const float *dz2,
                           const float *dz3, const float *dz4, float *dW1, float *db1, float *dW2, float *db2, float *dW3, float *db3, float *dW4,
                           float *db4, float *workspace, const int32_t *n_dev, void *stream) {
    if (n <= 0 || D0 < 1 || D0 > 128 || H < 1 || H > 128) { gom_set_error("gom_mlp3_wgrad: widths must be in 1..128"); return -1; }
    if (!x || !h1 || !h2 || !h3 || !dz1 || !dz2 || !dz3 || !dz4 || !dW1 || !db1 || !dW2 || !db2 || !dW3 || !db3 || !dW4 || !db4 || !workspace) {
        gom_set_error("gom_mlp3_wgrad: null pointer"); return -1;
    }
    WgradLayers L = {{x, h1, h2, h3}, {dz1, dz2, dz3, dz4}, {dW1, dW2, dW3, dW4}, {db1, db2, db3, db4}, {D0, H, H, H}, {H, H, H, 1}};
    const int slices = gom_linear_wgrad_slices();
    hipLaunchKernelGGL(k_mlp3_wgrad_partial, dim3(slices, 2, 8), dim3(256), 0, (hipStream_t)stream, n, L, workspace, slices, n_dev);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mlp3_wgrad_reduce, dim3((129 * 128 + 255) / 256, 4), dim3(256), 0, (hipStream_t)stream, slices, L, workspace);
    GOM_LAUNCH_CHECK();
    return 0;
}
extern "C" int gom_mlp3_wgrad(int64_t n, int D0, int H, const float *x, const float *h1, const float *h2, const float *h3, const float *dz1, const float *dz2,
                              const float *dz3, const float *dz4, float *dW1, float *db1, float *dW2, float *db2, float *dW3, float *db3, float *dW4,
                              float *db4, float *workspace, void *stream) {
    return mlp3_wgrad_impl(n, D0, H, x, h1, h2, h3, dz1, dz2, dz3, dz4, dW1, db1, dW2, db2, dW3, db3, dW4, db4, workspace, nullptr, stream);
}

// =====================================================================================================================================
// Shading of the pixels under the mesh, fused (round 4): models/model.py:279-283 evaluates shadow_module(normal) for every pixel; the
// normal map is exactly zero outside the mesh (~85 % of a frame), where the MLP's output is one constant.  The host layer used to select
// the pixels under the mesh with torch (ne / any / nonzero -- a host synchronisation --, index_select, cat, the embedding kernel,
// index_put, split, clone ... ~35 launches forward + backward of the drop-in Model's iteration).  Here:
//   gom_shade_select     ordered compaction of the pixels with a non-zero normal (two kernels: counts per 1 024 pixels, then positions),
//                        their positional encoding written straight into the MLP's input rows, row n = the background's (the zero normal);
//                        the row count stays in DEVICE memory -- no host synchronisation, static shapes (capacity = every pixel)
//   gom_mlp3_*_rows      the MLP kernels above with the row count read from that word (blocks beyond it return at once)
//   gom_shade_scatter    shading = scale * out[row of the pixel, or the background row]
//   gom_shade_backward_gather   d out rows from the image gradient (background row: the sum over the pixels outside the mesh, block
//                        partials added in block order by the last block to finish: one summation order)
//   gom_shade_backward_scatter  d normal = the embedding's backward of the pixel's row, zero outside the mesh
// Rows are in pixel order, as nonzero() returned them: the weight gradients sum the same rows in the same order as before.
namespace {
constexpr int kShadePx = 1024;   // pixels per workgroup of the selection kernels (4 per thread, consecutive)

__device__ __forceinline__ bool shade_under(const float *normal, size_t p) { return normal[3 * p] != 0.f || normal[3 * p + 1] != 0.f || normal[3 * p + 2] != 0.f; }

__global__ void __launch_bounds__(256) k_shade_count(size_t HW, const float *__restrict__ normal, int32_t *__restrict__ blk_cnt) {
    __shared__ int s_w[4];
    const size_t p0 = (size_t)blockIdx.x * kShadePx + 4 * threadIdx.x;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) c += (p0 + k < HW && shade_under(normal, p0 + k)) ? 1 : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__device__ __forceinline__ void posenc_row(const float *v, int L, float *o) {   // k_posenc_fwd's arithmetic (posenc.hip), one row
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    for (int l = 0; l < L; l++) {
        const float f = (float)(1u << l);
        float *q = o + 3 + 6 * l;
#pragma unroll
        for (int c = 0; c < 3; c++) { q[c] = sinf(v[c] * f); q[3 + c] = cosf(v[c] * f); }
    }
}
__global__ void __launch_bounds__(256) k_shade_place(size_t HW, int L, const float *__restrict__ normal, const int32_t *__restrict__ blk_cnt, int nblk,
                                                     int32_t *__restrict__ pos, float *__restrict__ pe, int32_t *__restrict__ n_dev) {
    __shared__ int s_w[4], s_base;
    // rows in front of this workgroup's pixels: the counts of the workgroups before it (a few hundred words from L2)
    int b = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) b += blk_cnt[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) b += __shfl_xor(b, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    const size_t p0 = (size_t)blockIdx.x * kShadePx + 4 * threadIdx.x;
    bool u[4];
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { u[k] = p0 + k < HW && shade_under(normal, p0 + k); c += u[k] ? 1 : 0; }
    // exclusive scan of the per-thread counts over the workgroup
    int incl = c;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); incl += lane >= d ? o : 0; }
    __syncthreads();
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    int off = s_base + incl - c;
    for (int w = 0; w < wv; w++) off += s_w[w];
    const int D = 3 + 6 * L;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (p0 + k >= HW) break;
        pos[p0 + k] = u[k] ? off : -1;
        if (u[k]) { posenc_row(normal + 3 * (p0 + k), L, pe + (size_t)off * D); off++; }
    }
    if (blockIdx.x == (unsigned)nblk - 1 && threadIdx.x == 255) {   // the last thread of the last workgroup knows the total
        n_dev[0] = off;
        const float z[3] = {0.f, 0.f, 0.f};
        posenc_row(z, L, pe + (size_t)off * D);   // the background's row
    }
}
__global__ void __launch_bounds__(256) k_shade_scatter(size_t HW, const int32_t *__restrict__ pos, const float *__restrict__ out, const int32_t *__restrict__ n_dev,
                                                       float scale, float *__restrict__ shading) {
    const int n = n_dev[0];
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (size_t)gridDim.x * 256) {
        const int r = pos[p];
        shading[p] = scale * out[r >= 0 ? r : n];
    }
}
// g_rows[row] = scale * g[p] under the mesh; g_rows[n] = scale * sum of g over the pixels outside it
__global__ void __launch_bounds__(256) k_shade_bwd_gather(size_t HW, const int32_t *__restrict__ pos, const float *__restrict__ g, const int32_t *__restrict__ n_dev,
                                                          float scale, float *__restrict__ g_rows, float *__restrict__ partial, uint32_t *__restrict__ done) {
    __shared__ float s_w[4];
    __shared__ bool s_last;
    float acc = 0.f;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (size_t)gridDim.x * 256) {
        const int r = pos[p];
        const float v = g[p];
        if (r >= 0) g_rows[r] = scale * v; else acc += v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        __threadfence();
        s_last = atomicAdd(done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (s_last) {   // the block partials in ONE order whatever workgroup finishes last: thread t adds partials 4t .. 4t+3, then a fixed tree
        __threadfence();
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) { const unsigned i = 4u * threadIdx.x + (unsigned)k; t += i < gridDim.x ? partial[i] : 0.f; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        __syncthreads();   // (s_w is read by thread 0 above)
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            g_rows[n_dev[0]] = scale * ((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
            *done = 0;
        }
    }
}
// d normal [HW][3]: the embedding's backward (k_posenc_bwd's arithmetic) of the pixel's row of dpe, zero outside the mesh
__global__ void __launch_bounds__(256) k_shade_bwd_scatter(size_t HW, int L, const int32_t *__restrict__ pos, const float *__restrict__ normal,
                                                           const float *__restrict__ dpe, float *__restrict__ d_normal) {
    const int D = 3 + 6 * L;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (size_t)gridDim.x * 256) {
        const int r = pos[p];
        float o[3] = {0.f, 0.f, 0.f};
        if (r >= 0) {
            const float *gr = dpe + (size_t)r * D;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float v = normal[3 * p + c];
                float acc = gr[c];
                for (int l = 0; l < L; l++) {
                    const float f = (float)(1u << l);
                    acc += f * (cosf(v * f) * gr[3 + 6 * l + c] - sinf(v * f) * gr[3 + 6 * l + 3 + c]);
                }
                o[c] = acc;
            }
        }
        d_normal[3 * p] = o[0]; d_normal[3 * p + 1] = o[1]; d_normal[3 * p + 2] = o[2];
    }
}
}  // namespace

extern "C" int gom_shade_workspace_ints(int64_t HW) { return (int)((HW + kShadePx - 1) / kShadePx) + 1024 + 4; }   // block counts, 1 024 gather partials (as floats), the row count, a counter

extern "C" int gom_shade_select(int64_t HW, int L, const float *normal, int32_t *pos, float *pe, int32_t *workspace, void *stream) {
    if (HW <= 0 || L < 0 || L > 16 || !normal || !pos || !pe || !workspace) { gom_set_error("gom_shade_select: bad arguments"); return -1; }
    const int nblk = (int)((HW + kShadePx - 1) / kShadePx);
    int32_t *blk_cnt = workspace, *n_dev = workspace + nblk + 1024;
    hipLaunchKernelGGL(k_shade_count, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (size_t)HW, normal, blk_cnt);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_shade_place, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (size_t)HW, L, normal, blk_cnt, nblk, pos, pe, n_dev);
    GOM_LAUNCH_CHECK();
    return 0;
}
static const int32_t *shade_n_dev(int64_t HW, const int32_t *workspace) { return workspace + (HW + kShadePx - 1) / kShadePx + 1024; }

// csrc/mlp_mc.hip: the same layers on the bf16 matrix cores at fp32 precision (three passes over hi / lo planes)
size_t gom_mlp3_mc_pack_elems(void);
bool gom_mlp3_mc_supported(int D0, int H);
int gom_mlp3_mc_forward(int64_t n, int D0, const float *x, const float *W1, const float *b1, const float *W2, const float *b2, const float *W3, const float *b3,
                        const float *w4, const float *b4, float *h1, float *h2, float *h3, float *out, const int32_t *n_dev, uint16_t *pack, void *stream);
int gom_mlp3_mc_backward(int64_t n, int D0, const float *g, const float *out, const float *h1, const float *h2, const float *h3, const float *W1, const float *W2,
                         const float *W3, const float *w4, float *dz4, float *dz3, float *dz2, float *dz1, float *dx, const int32_t *n_dev, uint16_t *pack,
                         void *stream);
extern "C" int gom_mlp3_pack_elems(void) { return (int)gom_mlp3_mc_pack_elems(); }

extern "C" int gom_mlp3_forward_rows(int64_t HW, const int32_t *workspace, int D0, int H, const float *x, const float *W1, const float *b1, const float *W2,
                                     const float *b2, const float *W3, const float *b3, const float *w4, const float *b4, float *h1, float *h2, float *h3,
                                     float *out, uint16_t *pack, void *stream) {
    if (!workspace) { gom_set_error("gom_mlp3_forward_rows: null workspace"); return -1; }
    if (pack && gom_mlp3_mc_supported(D0, H)) {
        if (!x || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !w4 || !b4 || !h1 || !h2 || !h3 || !out) { gom_set_error("gom_mlp3_forward_rows: null pointer"); return -1; }
        return gom_mlp3_mc_forward(HW + 1, D0, x, W1, b1, W2, b2, W3, b3, w4, b4, h1, h2, h3, out, shade_n_dev(HW, workspace), pack, stream);
    }
    return mlp3_forward_impl(HW + 1, D0, H, x, W1, b1, W2, b2, W3, b3, w4, b4, h1, h2, h3, out, shade_n_dev(HW, workspace), stream);
}
extern "C" int gom_mlp3_backward_rows(int64_t HW, const int32_t *workspace, int D0, int H, const float *g, const float *out, const float *h1, const float *h2,
                                      const float *h3, const float *W1, const float *W2, const float *W3, const float *w4, float *dz4, float *dz3, float *dz2,
                                      float *dz1, float *dx, uint16_t *pack, void *stream) {
    if (!workspace) { gom_set_error("gom_mlp3_backward_rows: null workspace"); return -1; }
    if (pack && gom_mlp3_mc_supported(D0, H)) {
        if (!g || !out || !h1 || !h2 || !h3 || !W1 || !W2 || !W3 || !w4 || !dz4 || !dz3 || !dz2 || !dz1 || !dx) { gom_set_error("gom_mlp3_backward_rows: null pointer"); return -1; }
        return gom_mlp3_mc_backward(HW + 1, D0, g, out, h1, h2, h3, W1, W2, W3, w4, dz4, dz3, dz2, dz1, dx, shade_n_dev(HW, workspace), pack, stream);
    }
    return mlp3_backward_impl(HW + 1, D0, H, g, out, h1, h2, h3, W1, W2, W3, w4, dz4, dz3, dz2, dz1, dx, shade_n_dev(HW, workspace), stream);
}
extern "C" int gom_mlp3_wgrad_rows(int64_t HW, const int32_t *workspace, int D0, int H, const float *x, const float *h1, const float *h2, const float *h3,
                                   const float *dz1, const float *dz2, const float *dz3, const float *dz4, float *dW1, float *db1, float *dW2, float *db2,
                                   float *dW3, float *db3, float *dW4, float *db4, float *wgrad_workspace, void *stream) {
    if (!workspace) { gom_set_error("gom_mlp3_wgrad_rows: null workspace"); return -1; }
    return mlp3_wgrad_impl(HW + 1, D0, H, x, h1, h2, h3, dz1, dz2, dz3, dz4, dW1, db1, dW2, db2, dW3, db3, dW4, db4, wgrad_workspace, shade_n_dev(HW, workspace), stream);
}
extern "C" int gom_shade_scatter(int64_t HW, const int32_t *pos, const float *out, const int32_t *workspace, float scale, float *shading, void *stream) {
    if (HW <= 0 || !pos || !out || !workspace || !shading) { gom_set_error("gom_shade_scatter: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_shade_scatter, dim3((unsigned)((HW + 255) / 256 < 2048 ? (HW + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream, (size_t)HW, pos, out,
                       shade_n_dev(HW, workspace), scale, shading);
    GOM_LAUNCH_CHECK();
    return 0;
}
extern "C" int gom_shade_backward_gather(int64_t HW, const int32_t *pos, const float *g, int32_t *workspace, float scale, float *g_rows, void *stream) {
    if (HW <= 0 || !pos || !g || !workspace || !g_rows) { gom_set_error("gom_shade_backward_gather: bad arguments"); return -1; }
    const int nblk = (int)((HW + kShadePx - 1) / kShadePx);
    const unsigned grid = (unsigned)((HW + 255) / 256 < 256 ? (HW + 255) / 256 : 256);   // (every workgroup ends with one atomic on one word: few, long workgroups)
    hipLaunchKernelGGL(k_shade_bwd_gather, dim3(grid), dim3(256), 0, (hipStream_t)stream, (size_t)HW, pos, g, shade_n_dev(HW, workspace), scale, g_rows,
                       reinterpret_cast<float *>(workspace + nblk), reinterpret_cast<uint32_t *>(workspace + nblk + 1024 + 1));
    GOM_LAUNCH_CHECK();
    return 0;
}
extern "C" int gom_shade_backward_scatter(int64_t HW, int L, const int32_t *pos, const float *normal, const float *dpe, float *d_normal, void *stream) {
    if (HW <= 0 || L < 0 || L > 16 || !pos || !normal || !dpe || !d_normal) { gom_set_error("gom_shade_backward_scatter: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_shade_bwd_scatter, dim3((unsigned)((HW + 255) / 256 < 2048 ? (HW + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream, (size_t)HW, L, pos,
                       normal, dpe, d_normal);
    GOM_LAUNCH_CHECK();
    return 0;
}
